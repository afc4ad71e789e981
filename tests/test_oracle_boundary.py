"""CPU: oracle/boundary.py against the imported reference's `boundary_conditions` outputs (tests/golden/boundary_*.npz),
bit for bit (masked writes + one fp32 expression)."""
import pytest
import torch

from oracle import boundary
from tests.helpers import boundary_case

CASES = ["boundary_ns_b2", "boundary_ns_n3b2", "boundary_spring_b3", "boundary_spring_n2b3"]


@pytest.mark.parametrize("name", CASES)
def test_oracle_boundary_conditions_equal_reference(name):
    system, preds, targets, meta, time, expected = boundary_case(name)
    got = boundary.boundary_conditions(system, preds.clone(), targets, meta, time=time)
    assert torch.equal(got, expected)
    assert not torch.equal(expected, preds)  # the fixture really changes something


def test_reference_indexing_of_ensemble_stacks():
    """navier-stokes indexes the FIRST dimension with the batch index: ensemble members >= B stay untouched, B > N raises."""
    system, preds, targets, meta, time, expected = boundary_case("boundary_ns_n3b2")
    assert torch.equal(expected[2], preds[2])          # member 2 (>= B = 2): untouched
    assert not torch.equal(expected[0], preds[0])
    with pytest.raises(IndexError):
        boundary.boundary_conditions(system, preds[:1].clone(), targets, meta, time=time)
