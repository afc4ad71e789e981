"""-m gpu: dyf_apply_boundary_conditions (one masked-write kernel, metadata in device memory) against the imported
reference's `PhysicalSystemsBenchmarkDataModule.boundary_conditions` outputs (tests/golden/boundary_*.npz) and the oracle.
Bit-exact: the op is masked writes plus one fp32 expression evaluated in the reference's order."""
import pytest
import torch

import dyffusion_amd as D
from oracle import boundary
from tests.gpu_common import DEV
from tests.helpers import boundary_case

pytestmark = pytest.mark.gpu
CASES = ["boundary_ns_b2", "boundary_ns_n3b2", "boundary_spring_b3", "boundary_spring_n2b3"]


@pytest.fixture(scope="module")
def engine():
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    return D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False)


@pytest.mark.parametrize("name", CASES)
def test_device_boundary_conditions_equal_reference(engine, name):
    system, preds, targets, meta, time, expected = boundary_case(name)
    bc = D.PhysicalSystemsBoundaryConditions(system, engine)
    x = preds.to(DEV)
    got = bc(preds=x, targets=targets, metadata=meta, time=time)
    assert got.data_ptr() == x.data_ptr()  # in place, like the reference
    diff = (got.cpu() != expected)
    assert not bool(diff.any()), (int(diff.sum()), float((got.cpu() - expected).abs().max()))


def test_seeded_cases_equal_oracle_and_errors_match(engine):
    g = torch.Generator().manual_seed(3)
    B, N = 4, 6
    meta = {"fixed_mask": torch.rand(B, 3, 221, 42, generator=g) < 0.1, "in_velocity": 1 + torch.rand(B, generator=g),
            "vertices": torch.rand(B, 2, 221, 42, generator=g) * 0.41}
    time = torch.tensor([0.1, 0.2, 0.7, 3.0])
    preds = torch.randn(N, B, 3, 221, 42, generator=g)
    want = boundary.boundary_conditions("navier-stokes", preds.clone(), torch.zeros(B, 3, 221, 42), meta, time=time)
    bc = D.PhysicalSystemsBoundaryConditions("navier-stokes", engine)
    got = bc(preds=preds.to(DEV), targets=torch.zeros(B, 3, 221, 42), metadata=meta, time=time)
    assert torch.equal(got.cpu(), want)
    with pytest.raises(IndexError):  # B > N: the reference indexes member b_i of an (N, B, ...) stack
        bc(preds=preds[:2].contiguous().to(DEV), targets=torch.zeros(B, 3, 221, 42), metadata=meta, time=time)
    with pytest.raises(NotImplementedError):
        D.PhysicalSystemsBoundaryConditions("oisst", engine)
