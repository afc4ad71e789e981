"""CPU checks of the host side: state_dict compatibility, schedule logic vs the reference goldens, the C-ABI
library's exported symbols, and that the product path refuses to run without the HIP engine / a GPU."""
import ctypes
import os
import re

import pytest
import torch

import dyffusion_amd as D
from dyffusion_amd import _lib
from oracle import init as oinit
from tests.helpers import jload

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def test_mirror_state_dict_matches_reference_layout():
    for dim, cin, ccond, cout in [(64, 3, 2, 3), (64, 6, 2, 3), (8, 4, 1, 4)]:
        net = D.UNet(dim=dim, with_time_emb=True, upsample_dims=[64, 64], dropout=0.1, num_input_channels=cin,
                     num_output_channels=cout, num_conditional_channels=ccond)
        want = oinit.unet_simple_param_shapes(dim, cin + ccond, cout)  # verified == reference in make_golden.py
        got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        assert got == {k: tuple(v) for k, v in want.items()}


def test_simple_conv_net_mirror_state_dict_matches_reference_layout():
    ks = [9, 7, 5, 3]
    net = D.SimpleConvNet(dim=8, with_time_emb=True, kernel_sizes=ks, dropout=0.1, num_input_channels=8,
                          num_output_channels=4, num_conditional_channels=1)
    want = oinit.simple_conv_net_param_shapes(8, 9, 4, ks)  # == reference layout (tests/golden/net_simple_conv.npz)
    got = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    assert got == {k: tuple(v) for k, v in want.items()}
    cfg = net.engine_net_config()
    assert cfg.arch == 2 and cfg.n_mults == 4 and list(cfg.dim_mults[:4]) == ks


def _pair(h, **kw):
    F = D.UNet(dim=8, with_time_emb=True, upsample_dims=[64, 64], num_input_channels=4, num_output_channels=4,
               num_conditional_channels=1)
    I = D.UNet(dim=8, with_time_emb=True, upsample_dims=[64, 64], num_input_channels=8, num_output_channels=4,
               num_conditional_channels=1)
    return D.DYffusion(F, D.InterpolatorHandle(I, h), timesteps=h, forward_conditioning="none", **kw)


@pytest.mark.parametrize("case", jload("schedules.json"),
                         ids=lambda c: f"{c['schedule']}-h{c['h']}-k{c['k']}-f{c['fac']}-b{int(c['before_t1'])}")
def test_product_schedule_logic_matches_reference(case):
    m = _pair(case["h"], schedule=case["schedule"], additional_interpolation_steps=case["k"],
              additional_interpolation_steps_factor=case["fac"], interpolate_before_t1=case["before_t1"])
    assert m.num_timesteps == case["num_timesteps"]
    for d, i in case["d_to_i"].items():
        assert float(m.diffusion_step_to_interpolation_step(int(d))) == pytest.approx(i, abs=1e-12)
        ti = float(m.diffusion_step_to_interpolation_step(torch.tensor(float(d))))
        assert ti == pytest.approx(i, abs=4e-6)  # the reference's own float-vs-tensor self check (dyffusion.py:75-80)
    assert {str(k) for k in m.dynamical_steps} == set(case["dynamical_steps"])
    for name, want in case["schedules"].items():
        spec = m.full_sampling_schedule if name == "None" else name
        if not want["ok"]:
            with pytest.raises((AssertionError, ValueError, IndexError)):
                m.sampling_schedule = spec
            continue
        m.sampling_schedule = spec
        assert [float(s) for s in m.sampling_schedule] == pytest.approx(want["steps"], abs=1e-12), name


def test_plan_resolution_ns_config():
    F = D.UNet(dim=64, with_time_emb=True, num_input_channels=3, num_output_channels=3, num_conditional_channels=2)
    I = D.UNet(dim=64, with_time_emb=True, num_input_channels=6, num_output_channels=3, num_conditional_channels=2)
    m = D.DYffusion(F, D.InterpolatorHandle(I, 16), timesteps=16, forward_conditioning="none",
                    interpolate_before_t1=True, refine_intermediate_predictions=True)
    steps, refine, n_slots = m._build_plan()
    assert n_slots == 16 and len(steps) == 16 and len(refine) == 15
    n_i = sum(s["i_next"] is not None for s in steps) + sum(s["i_cur"] is not None and not s["is_last"] for s in steps)
    assert n_i + len(refine) == 44  # 16 forecaster + 44 interpolator forwards
    assert [s["out_slot"] for s in steps] == list(range(16))


def test_invalid_arguments_raise_like_reference():
    with pytest.raises(AssertionError):
        _pair(1, interpolate_before_t1=True)
    with pytest.raises(AssertionError):
        _pair(4, interpolate_before_t1=False)
    with pytest.raises(ValueError):
        _pair(4, schedule="cosine", interpolate_before_t1=True)
    with pytest.raises(ValueError):
        F = D.UNet(dim=8, with_time_emb=True, num_input_channels=4, num_output_channels=4)
        D.DYffusion(F, D.InterpolatorHandle(F, 7), timesteps=4, interpolate_before_t1=True)  # horizon mismatch


def test_library_exports_every_declared_symbol():
    hdr = open(os.path.join(ROOT, "include", "dyffusion_hip.h")).read() + \
        open(os.path.join(ROOT, "include", "dyffusion_hip_testing.h")).read()
    declared = set(re.findall(r"\b(dyf_[a-z0-9_]+)\s*\(", hdr))
    assert {"dyf_engine_create", "dyf_sample", "dyf_net_forward", "dyf_load_weights"} <= declared
    lib = ctypes.CDLL(_lib.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), f"{name} declared in include/dyffusion_hip.h but not exported"
    assert {n for n, _, _ in _lib.SYMBOLS} == declared, "ctypes binding table out of sync with the header"
    assert lib.dyf_abi_version() == _lib.DYF_ABI_VERSION


def test_kernel_form_switches_come_from_the_testing_call_only_never_from_the_environment(monkeypatch):
    """libdyffusion_hip.so read 79 DYF_* variables from the environment (kernel forms and, for the training
    operands, numerics, under a caller who never asked).  Now: (1) the only getenv()s left in csrc/ are DYF_VERBOSE and DYF_RCCL_LIB;
    (2) the switch table is written through dyf_debug_set_form alone -- both builds, set / overwrite / remove / clear, and a variable
    in the process environment does not appear in it."""
    csrc = os.path.join(ROOT, "dyffusion_amd", "csrc")
    seen = set()
    for fn in os.listdir(csrc):
        seen |= set(re.findall(r'getenv\("([A-Z0-9_]+)"\)', open(os.path.join(csrc, fn)).read()))
    assert seen == {"DYF_VERBOSE", "DYF_RCCL_LIB"}, seen
    monkeypatch.setenv("DYF_IGEMM2_MIN_TILES", "1")

    def table(lib):
        buf = ctypes.create_string_buffer(4096)
        n = lib.dyf_debug_forms(buf, 4096)
        assert n == len(buf.value)
        return dict(kv.split("=", 1) for kv in buf.value.decode().split(";") if kv)

    try:
        for dtype in ("bf16", "fp16"):
            lib = _lib.lib(dtype)
            assert table(lib) == {}  # nothing from the environment
        _lib.set_form("DYF_GN16", "0")
        _lib.set_form("DYF_ROWS_TR", "2")
        _lib.set_form("DYF_GN16", "1")
        for dtype in ("bf16", "fp16"):
            assert table(_lib.lib(dtype)) == {"DYF_GN16": "1", "DYF_ROWS_TR": "2"}
        _lib.set_form("DYF_ROWS_TR", None)
        assert table(_lib.lib("bf16")) == {"DYF_GN16": "1"} and _lib.forms() == {"DYF_GN16": "1"}
    finally:
        _lib.set_form(None)
    assert table(_lib.lib("bf16")) == {} and table(_lib.lib("fp16")) == {}


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU failure mode")
def test_product_path_fails_loudly_without_gpu():
    m = _pair(4, interpolate_before_t1=True)
    with pytest.raises(D.EngineError):
        m.sample(torch.zeros(1, 4, 10, 10), static_condition=torch.zeros(1, 1, 10, 10))
    with pytest.raises(D.EngineError):
        m.model(torch.zeros(1, 4, 10, 10), time=torch.zeros(1), condition=torch.zeros(1, 1, 10, 10))


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "dyffusion_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            src = open(os.path.join(pkg, fn)).read()
            assert "oracle" not in src.replace("the oracle", ""), f"{fn} references the oracle package"


def test_resnet_unet_mirror_state_dict_matches_reference_layout():
    import json
    from tests.helpers import load_npz, split_state
    for name in ("net_unet_resnet_a", "net_unet_resnet_b"):
        z = load_npz(name + ".npz")
        P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
        n_cond = z["c"].shape[1] if "c" in z else 0
        net = D.Unet(dim=cfg["dim"], dim_mults=cfg["dim_mults"], with_time_emb=True, num_input_channels=z["x"].shape[1],
                     num_output_channels=1, num_conditional_channels=n_cond)
        assert {k: tuple(v.shape) for k, v in net.state_dict().items()} == {k: tuple(v.shape) for k, v in P.items()}


def test_boundary_metadata_cache_is_keyed_by_the_object_not_its_address():
    """A freed metadata dict's address is reused by CPython (`id()` collides for consecutive per-step batch dicts): the device
    copies must be rebuilt for a new dict even at the same address, and reused for the same dict (h calls per batch)."""
    from dyffusion_amd.boundary import PhysicalSystemsBoundaryConditions

    bc = PhysicalSystemsBoundaryConditions("navier-stokes", engine=None)

    def mk(i):
        return {"fixed_mask": torch.full((2, 3, 4, 5), bool(i % 2)), "in_velocity": torch.full((2,), float(i)),
                "vertices": torch.full((2, 2, 4, 5), float(i))}

    seen_ids = set()
    for i in range(6):  # each dict dies before the next is made: addresses repeat
        meta = mk(i)
        seen_ids.add(id(meta))
        d = bc._prepare(meta, "cpu")
        assert float(d["in_velocity"][0]) == float(i) and bool(d["fixed_mask"].any()) == bool(i % 2), i
        assert bc._prepare(meta, "cpu") is d  # same object, unchanged tensors: cache hit
        meta["in_velocity"].add_(100.0)       # in-place edit of a source tensor: cache miss
        assert float(bc._prepare(meta, "cpu")["in_velocity"][0]) == float(i) + 100.0
        del meta, d
    assert len(seen_ids) < 6, "this interpreter did not reuse an address; the test did not exercise the collision"


def test_validation_split_runs_one_outer_iteration_like_the_reference():
    """forecasting_multi_horizon.py:134-141: split == 'val' with dataloader_idx in (0, None) -> ONE rollout and no length check;
    every other split runs num_autoregressive_steps + 1 rollouts and refuses batches shorter than the prediction horizon."""
    calls = []

    class FakeDiffusion(torch.nn.Module):
        hparams = D.dyffusion._AttrDict(timesteps=4)
        _engine = None

        def predict_forward(self, inputs, num_predictions=None, **kw):
            calls.append(inputs.clone())
            return {f"t{k}_preds": inputs[:, :3] + k for k in range(1, 5)}

    exp = D.MultiHorizonForecastingDYffusion(FakeDiffusion(), num_predictions=2, autoregressive_steps=2)
    assert exp.prediction_horizon == 12
    dyn = torch.randn(3, 1 + 4, 3, 6, 5)  # window + ONE horizon: a normal validation batch
    out = exp.validation_step({"dynamics": dyn.clone()})
    assert len(calls) == 1 and sorted(out) == sorted([f"t{k}_{s}" for k in range(1, 5) for s in ("preds", "targets")])
    assert torch.equal(out["t4_targets"], dyn[:, 4])  # batch["dynamics"] not multiplied by 1e6
    with pytest.raises(ValueError):
        exp.evaluation_step({"dynamics": dyn.clone()}, split="test")
    calls.clear()
    long = torch.randn(3, 1 + 12, 3, 6, 5)
    out = exp.validation_step({"dynamics": long.clone()}, dataloader_idx=1)  # second val loader: the full autoregressive rollout
    assert len(calls) == 3 and "t12_preds" in out


def test_boundary_metadata_key_accepts_inference_tensors():
    """Lightning's evaluation loops move the batch to the device under torch.inference_mode(); such tensors have no
    version counter (`_version` raises), and the per-batch metadata cache key must not read it."""
    import torch
    from dyffusion_amd.boundary import PhysicalSystemsBoundaryConditions

    bc = PhysicalSystemsBoundaryConditions("navier-stokes", engine=None)
    with torch.inference_mode():
        meta = {"fixed_mask": torch.zeros(2, 3, 5, 4, dtype=torch.bool), "in_velocity": torch.ones(2, 1),
                "vertices": torch.zeros(2, 2, 5, 4)}
    assert meta["fixed_mask"].is_inference()
    key = bc._source_key(meta)
    assert key == bc._source_key(meta) and key[0][1] is None
    plain = {k: v.clone() for k, v in meta.items()}  # ordinary tensors keep their modification counter in the key
    k0 = bc._source_key(plain)
    plain["in_velocity"].mul_(2)
    assert bc._source_key(plain) != k0


def test_engine_loss_rejects_a_tape_overwritten_by_another_owner():
    """two owners can share one engine (a DYffusion and its attached forecaster's get_loss); a training forward by the
    OTHER owner overwrites the tape slots, so a loss of the first owner must refuse to run backward."""
    import pytest
    import torch
    from dyffusion_amd.engine import EngineLoss

    class Eng:
        train_step_id = 0

    class Owner:
        def __init__(self, eng):
            self.eng, self.ran = eng, 0

        def forward(self):
            self.eng.train_step_id += 1
            self._train_state = dict(eng=self.eng, step_id=self.eng.train_step_id)
            return EngineLoss.apply(torch.zeros((), requires_grad=True), self, 1.0)

        def _train_backward(self, upstream):
            self.ran += 1

    eng = Eng()
    a, b = Owner(eng), Owner(eng)
    la = a.forward()
    la.backward()
    assert a.ran == 1
    la = a.forward()
    b.forward()  # overwrites the shared tapes
    with pytest.raises(RuntimeError, match="earlier training forward"):
        la.backward()
    assert a.ran == 1


def test_unet_learned_variance_is_the_reference_noop_and_init_dim_is_refused():
    """unet.py:233-236: `learned_variance` only feeds default_out_dim, which out_dim = default(output_channels, ...) never uses
    (output_channels is always set) -- accepted as the no-op it is; init_dim != dim cannot run in the reference either
    (final_res_block expects 2 * dim channels) and stays refused, as does init_stride != 1."""
    import os
    import pytest
    import dyffusion_amd as D

    kw = dict(dim=8, dim_mults=(1, 2), with_time_emb=True, num_input_channels=2, num_output_channels=3, num_conditional_channels=1)
    a, b = D.Unet(**kw), D.Unet(learned_variance=True, **kw)
    assert {k: tuple(v.shape) for k, v in a.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
    assert b.final_conv.out_channels == 3 and b.hparams.learned_variance is True
    for bad in (dict(init_dim=16), dict(init_stride=2)):
        with pytest.raises(NotImplementedError):
            D.Unet(**bad, **kw)
    if os.path.isdir("/root/reference/src"):  # the reference agrees (build container only)
        from oracle import ref_import
        ref_import.activate()
        from src.models.unet import Unet as RefUnet
        r = RefUnet(dim=8, dim_mults=(1, 2), with_time_emb=True, learned_variance=True, num_input_channels=2, num_output_channels=3,
                    num_conditional_channels=1, spatial_shape=(8, 8), verbose=False)
        assert r.final_conv.out_channels == 3
        assert {k: tuple(v.shape) for k, v in r.state_dict().items()} == {k: tuple(v.shape) for k, v in b.state_dict().items()}
