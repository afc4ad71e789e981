"""-m gpu: the caller contract of the hot path on the engine, against outputs of the imported reference.

  * `MultiHorizonForecastingDYffusion.predict()` (`_base_experiment.py:315-379`): ensemble-tiled inputs -> `t{i}_preds`
    reshaped to (N, B, ...), ensemble-major -- golden `sample_ens3` (N=3, B=2).
  * `predict_step()` (`_base_experiment.py:700-703` -> `evaluation_step` -> `_evaluation_step`,
    forecasting_multi_horizon.py:114-229): numpy outputs appended to `_predict_step_outputs`, targets next to predictions,
    TWO autoregressive outer iterations re-feeding t4, the datamodule's spring-mesh boundary conditions applied to every
    field on the device, the batch's dynamics scaled by 1e6 afterwards -- golden `predict_step_spring_ar2`.
"""
import json

import numpy as np
import pytest
import torch

import dyffusion_amd as D
from tests.gpu_common import DEV, build_dyffusion
from tests.helpers import load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu
TOL = 2.5e-2


def test_predict_matches_reference_golden_ensemble_layout():
    z = load_npz("sample_ens3.npz")
    hp = json.loads(str(z["hp"]))
    N, B = hp["num_predictions"], hp["B"]
    m = build_dyffusion(split_state(z, "F"), split_state(z, "I"), hp["model"], 4, 1, hp, max_batch=N * B)
    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N)
    x0, c = torch.from_numpy(z["x0"]).to(DEV), torch.from_numpy(z["c"]).to(DEV)
    out = exp.predict(exp.get_ensemble_inputs(x0), condition=exp.get_ensemble_inputs(c))
    want = {k[len("out::"):]: v for k, v in z.items() if k.startswith("out::")}
    assert sorted(out) == sorted(want)
    for k, w in want.items():
        assert tuple(out[k].shape) == w.shape == (N, B, 4, 10, 10), k
        assert rel_rms(out[k].cpu(), w) <= TOL, k
    # dropout is off in this fixture: the N members of one batch item are identical, batch items differ (ensemble-major rows)
    assert torch.equal(out["t4_preds"][0], out["t4_preds"][2]) and not torch.equal(out["t4_preds"][0, 0], out["t4_preds"][0, 1])
    flat = exp.predict(exp.get_ensemble_inputs(x0), condition=exp.get_ensemble_inputs(c), reshape_ensemble_dim=False)
    assert tuple(flat["t4_preds"].shape) == (N * B, 4, 10, 10)
    assert torch.equal(flat["t4_preds"].reshape(N, B, 4, 10, 10), out["t4_preds"])


class _SpringDataModule:
    """The two datamodule methods `evaluation_step` touches (_base_experiment.py:486-488), on the device op."""

    def __init__(self, engine):
        self.bc = D.PhysicalSystemsBoundaryConditions("spring-mesh", engine)

    def boundary_conditions(self, preds, targets, metadata, time=None):
        return self.bc(preds=preds, targets=targets, metadata=metadata, time=time)

    def get_boundary_condition_kwargs(self, batch, batch_idx, split):
        return dict(t0=0.0, dt=1.0)


def test_predict_step_autoregressive_with_boundary_conditions_matches_reference_golden():
    z = load_npz("predict_step_spring_ar2.npz")
    hp = json.loads(str(z["hp"]))
    N, B, h = hp["num_predictions"], hp["B"], hp["timesteps"]
    m = build_dyffusion(split_state(z, "F"), split_state(z, "I"), hp["model"], 4, 1, hp, max_batch=N * B)
    eng = m._ensure_engine((10, 10), N * B)
    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N, autoregressive_steps=1, datamodule=_SpringDataModule(eng))
    assert exp.prediction_horizon == 2 * h == hp["prediction_horizon"] and exp.num_autoregressive_steps == 1
    feats = torch.zeros(B, 5, 4, 10, 10)
    feats[:, 0, 2:] = torch.from_numpy(z["base_q"])
    dyn = torch.from_numpy(z["dynamics"]).to(DEV)
    batch = {"dynamics": dyn.clone(), "condition": torch.from_numpy(z["condition"]).to(DEV),
             "metadata": {"fixed_mask": torch.from_numpy(z["fixed_mask"]), "features": feats}}
    assert exp.predict_step(batch, 0) is None  # like the reference: results are collected, not returned
    got = exp._predict_step_outputs[0]
    want = {k[len("out::"):]: v for k, v in z.items() if k.startswith("out::")}
    assert list(got) == list(want)  # same keys in the same order: t1_targets, t1_preds, ..., t8_targets, t8_preds
    worst = 0.0
    fixed = z["fixed_mask"]
    for k, w in want.items():
        assert isinstance(got[k], np.ndarray) and got[k].shape == w.shape, k
        if k.endswith("targets"):
            assert np.array_equal(got[k], w), k
            continue
        worst = max(worst, rel_rms(got[k], w))
        # boundary values are written, not computed: exact on every fixed node of every member (rows n*B + b -> metadata b)
        assert np.array_equal(got[k][:, fixed], w[:, fixed]), k
    print("predict_step (2 x h=4, spring BC) worst rel-rms", worst)
    assert worst <= 2.5e-2
    assert torch.allclose(batch["dynamics"], dyn * 1e6)  # forecasting_multi_horizon.py:221
    merged = exp.on_predict_epoch_end()
    assert merged["t8_preds"].shape == (N, B, 4, 10, 10) and exp._predict_step_outputs == []


@pytest.mark.parametrize("ph", [32, 64], ids=["2-outer-iterations", "configs3-h64-4-outer-iterations"])
def test_navier_stokes_long_rollout_with_device_boundary_conditions(ph):
    """BASELINE configs[3] at full size: Navier-Stokes 221x42, dim 64 @256^2, prediction horizon 64 (and 32) with horizon
    16 -> four (two) autoregressive outer iterations re-feeding t16 / t32 / t48 (forecasting_multi_horizon.py:149-221), N=2
    members x B=2, the reference's NS boundary conditions
    (fixed mask zeroed, parabolic inflow growing as 1 - exp(-5 t), per-batch-element times t0 + k dt) applied to every field
    on the device.  Checks: every field finite and carrying the boundary values EXACTLY where the reference writes them
    (ensemble members 0..B-1 of the (N, B, ...) stack -- its first-dimension indexing), untouched elsewhere relative to a run
    without boundary conditions for the first outer iteration, and the second iteration starting from the BC-applied t16."""
    from oracle import boundary as obc
    from tests.gpu_common import seeded_pair
    PF, PI = seeded_pair(64, 3, 2)
    mk = dict(dim=64, upsample_dims=[256, 256], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    N, B = 2, 2
    g = torch.Generator().manual_seed(8)
    dyn = torch.randn(B, ph + 1, 3, 221, 42, generator=g)
    cond = torch.rand(B, 2, 221, 42, generator=g)
    meta = {"fixed_mask": torch.rand(B, 3, 221, 42, generator=g) < 0.05, "in_velocity": 1.0 + torch.rand(B, generator=g),
            "vertices": torch.rand(B, 2, 221, 42, generator=g) * 0.41}
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=N * B)
    bc = D.PhysicalSystemsBoundaryConditions("navier-stokes", m._ensure_engine((221, 42), N * B))

    class DM:
        def boundary_conditions(self, preds, targets, metadata, time=None):
            return bc(preds=preds, targets=targets, metadata=metadata, time=time)

        def get_boundary_condition_kwargs(self, batch, batch_idx, split):
            return dict(t0=torch.tensor([0.0, 0.5]), dt=torch.tensor([0.01, 0.02]))

    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N, prediction_horizon=ph, datamodule=DM())
    assert exp.num_autoregressive_steps == ph // 16 - 1
    batch = {"dynamics": dyn.clone().to(DEV), "condition": cond.to(DEV), "metadata": meta}
    out = exp.evaluation_step(batch)
    free = D.MultiHorizonForecastingDYffusion(m, num_predictions=N, prediction_horizon=16).evaluation_step(
        {"dynamics": dyn.clone().to(DEV), "condition": cond.to(DEV)})
    assert [k for k in out if k.endswith("preds")] == [f"t{k}_preds" for k in range(1, ph + 1)]
    for k in sorted({1, 16, 17, 32, ph - 31, ph - 16, ph - 15, ph}):
        p = out[f"t{k}_preds"].cpu()
        assert tuple(p.shape) == (N, B, 3, 221, 42) and bool(torch.isfinite(p).all())
        time = torch.tensor([0.0, 0.5])
        for _ in range(k):  # total_t += dt per step, in fp32, as _evaluation_step accumulates it
            time = time + torch.tensor([0.01, 0.02])
        want = obc.boundary_conditions("navier-stokes", p.clone(), dyn[:, k], meta, time=time)
        assert torch.equal(p, want), k  # idempotent: the field already satisfies the reference's boundary conditions
        if k <= 16:  # first outer iteration: identical to the unconstrained rollout away from the written elements
            q = obc.boundary_conditions("navier-stokes", free[f"t{k}_preds"].cpu().clone(), dyn[:, k], meta, time=time)
            assert torch.equal(p, q), k
    assert not torch.equal(out["t17_preds"].cpu(), out["t1_preds"].cpu())
    # every outer iteration starts from the BC-applied last field of the one before: re-running iteration j alone from that field
    # (one rollout, no boundary conditions) reproduces the raw forecasts the boundary conditions were then applied to
    for j in range(1, ph // 16):
        start = out[f"t{16 * j}_preds"].reshape(N * B, 3, 221, 42)
        again = m.sample(start, static_condition=cond.to(DEV).repeat(N, 1, 1, 1))
        time = torch.tensor([0.0, 0.5])
        for _ in range(16 * j + 16):
            time = time + torch.tensor([0.01, 0.02])
        q = obc.boundary_conditions("navier-stokes", again["t16_preds"].reshape(N, B, 3, 221, 42).cpu().clone(), dyn[:, 16 * j + 16], meta, time=time)
        assert torch.equal(out[f"t{16 * j + 16}_preds"].cpu(), q), j


def test_forecaster_experiment_get_loss_routes_the_batch_like_the_reference(monkeypatch):
    """`MultiHorizonForecastingDYffusion.get_loss(batch)` (forecasting_multi_horizon.py:412-420 -> BaseDiffusion.get_loss / forward,
    _base_diffusion.py:81-117): first frame(s) = inputs = `condition`, last frame = `xt_last`, batch["condition"] = static
    condition, one diffusion step per item from torch.randint -- equal to calling `p_losses` with those arguments directly."""
    import json
    from tests.gpu_common import DEV, build_dyffusion
    from tests.helpers import load_npz, split_state
    z = load_npz("plosses_a.npz")
    hp = json.loads(str(z["hp"]))
    PF, PI = split_state(z, "F"), split_state(z, "I")
    xt_last, cond, sc, t = (torch.from_numpy(z[k]).to(DEV) for k in ("xt_last", "cond", "sc", "t"))
    m = build_dyffusion(PF, PI, hp["model"], 4, 1, hp, max_batch=xt_last.shape[0])
    exp = D.MultiHorizonForecastingDYffusion(m, window=1)
    h = hp["timesteps"]
    dyn = torch.randn(xt_last.shape[0], 1 + h, *xt_last.shape[1:], device=DEV)
    dyn[:, 0], dyn[:, -1] = cond, xt_last
    monkeypatch.setattr(torch, "randint", lambda *a, **k: t.clone())
    got = exp.get_loss({"dynamics": dyn, "condition": sc})
    monkeypatch.undo()
    want = m.p_losses(xt_last, cond, t, static_condition=sc)
    assert set(got) == set(want)
    for k in want:
        assert float(got[k]) == float(want[k]), k
    ref = json.loads(str(z["losses"]))  # the imported reference's own p_losses on these tensors
    assert float(got["loss"]) == pytest.approx(ref["loss"], rel=2e-2)


def test_validation_epoch_ensemble_metrics_on_the_device():
    """validation_step x 2 + on_validation_epoch_end (_base_experiment.py:603-660): the per-horizon CRPS / SSR / MSE of the
    concatenated ensemble forecasts, computed on the GPU, against oracle/metrics.py (pinned to the reference's
    evaluate_ensemble_prediction by tests/test_oracle_metrics.py) on the same fields."""
    from oracle import metrics as om
    from tests.gpu_common import seeded_pair
    mk = dict(dim=8, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True, input_dropout=0.0, dropout=0.2)
    hp = dict(timesteps=3, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=True)
    C, Cs, N = 2, 1, 4
    PF, PI = seeded_pair(8, C, Cs)
    m = build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=N * 3)
    m.seed(7)
    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N, window=1)
    g = torch.Generator().manual_seed(2)
    outs = []
    for b in (3, 2):
        batch = {"dynamics": torch.randn(b, 1 + 3, C, 19, 13, generator=g).to(DEV), "condition": torch.rand(b, Cs, 19, 13, generator=g).to(DEV)}
        outs.append(exp.validation_step(batch, 0))
    got = exp.on_validation_epoch_end()
    assert exp.on_validation_epoch_end() == {}  # the outputs were consumed
    for k in (1, 2, 3):
        preds = torch.cat([o[f"t{k}_preds"] for o in outs], dim=1).cpu().numpy()     # (N, 5, C, H, W)
        targets = torch.cat([o[f"t{k}_targets"] for o in outs], dim=0).cpu().numpy()
        assert preds.shape[:2] == (N, 5)
        ref = om.evaluate_ensemble_prediction(preds, targets)
        for name in ("mse", "ssr", "crps"):
            assert got[f"val/{N}ens_mems/t{k}/{name}"] == pytest.approx(ref[name], rel=5e-5), (k, name)
    assert got[f"val/{N}ens_mems/avg/crps"] == pytest.approx(np.mean([got[f"val/{N}ens_mems/t{k}/crps"] for k in (1, 2, 3)]))
