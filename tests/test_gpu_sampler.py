"""-m gpu: dyf_sample (HIP, hipGraph) against the reference's golden rollouts and the oracle.

Tolerance (stated): rel-RMS <= 2.5e-2 per forecast field over a rollout for the bf16 engine (bf16 operands and block
outputs, fp32 accumulation, fp32 sampler state and cold-sampling update); measured values are printed (7.5e-3 - 2.1e-2).
Why not SURVEY 8c's 1e-2: that figure rested on "the reference's own bf16-autocast drift is 3.5e-3 - 4.5e-3"; measured on
the full-size fixture G6 (tests/measure_bf16_drift.py, profiles/r02_bf16_drift.json) the reference under torch.autocast(bf16)
drifts 3.1e-2 - 4.1e-2 from its fp32 self over the h=16 rollout, bf16-rounded weights ALONE give 1.1e-2 - 1.4e-2, and the
oracle evaluated with the engine's rounding points (`nets.unet_simple_forward_bf16_model`) 1.7e-2 - 2.1e-2 -- which is what
the engine measures, layer by layer (`test_fullsize_engine_error_is_the_bf16_storage_error`).  The fp16 build of the same
kernels (11-bit mantissa) is held to 1e-2 (tests/test_gpu_fp16.py).
"""
import json

import numpy as np
import pytest
import torch

from oracle import nets, sampler
from tests.gpu_common import DEV, build_dyffusion, nhwc_masks, oracle_rollout, seeded_pair
from tests.helpers import jload, load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu
TOL = 2.5e-2

NAMES = ["sample_cold_refine", "sample_cold_norefine", "sample_naive", "sample_k2_data", "sample_k2_coldlast",
         "sample_k2_onlydyn", "sample_k2_plus2", "sample_ens3", "sample_dropout", "sample_datanoise", "sample_linear", "sample_fractional_refine",
         "sample_log_cold", "sample_log_naive"]


@pytest.mark.parametrize("name", NAMES)
def test_rollout_matches_reference_golden(name):
    """Every G4 fixture: predict() outputs of the imported reference (tiny dim-4 nets -> direct-conv path)."""
    z = load_npz(name + ".npz")
    hp = json.loads(str(z["hp"]))
    PF, PI = split_state(z, "F"), split_state(z, "I")
    N, B = hp["num_predictions"], hp["B"]
    x0 = torch.from_numpy(z["x0"]).repeat(N, 1, 1, 1)
    c = torch.from_numpy(z["c"]).repeat(N, 1, 1, 1)
    m = build_dyffusion(PF, PI, hp["model"], 4, 1, hp, max_batch=N * B)
    masks = noise = None
    if hp.get("enable_interpolator_dropout"):
        src = nets.DropoutSeeded(hp["dropout_seed"], record=True)
        oracle_rollout(PF, PI, hp["model"], hp, x0, c, drop=src)
        masks = nhwc_masks(src.masks)
    if hp["forward_conditioning"] == "data+noise":
        gen = torch.Generator().manual_seed(hp["noise_seed"])
        draws = []

        def nf(t):
            draws.append(torch.randn(t.shape, generator=gen))
            return draws[-1]

        oracle_rollout(PF, PI, hp["model"], hp, x0, c, noise_fn=nf)
        noise = torch.stack(draws, 0).to(DEV)
    _, got, _ = m.sample_loop(x0.to(DEV), static_condition=c.to(DEV), _masks=masks, _noise=noise)
    want = {k[len("out::"):]: v for k, v in z.items() if k.startswith("out::")}
    assert sorted(got) == sorted(want)
    worst = 0.0
    for k, w in want.items():
        g = got[k].cpu().reshape(w.shape)  # (N*B, ...) -> (N, B, ...), ensemble-major
        worst = max(worst, rel_rms(g, w))
    print(name, "worst rel-rms", worst)
    assert worst <= TOL


def test_graph_replay_equals_eager_and_is_repeatable():
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False,
              num_input_channels=3)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(2)
    x0, c = torch.randn(3, 3, 23, 11, generator=g).to(DEV), torch.rand(3, 2, 23, 11, generator=g).to(DEV)
    a = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=3, use_graph=True)
    b = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=3, use_graph=False)
    ya1, ya2, yb = a.sample(x0, static_condition=c), a.sample(x0, static_condition=c), b.sample(x0, static_condition=c)
    for k in yb:
        assert torch.equal(ya1[k], yb[k]) and torch.equal(ya2[k], yb[k]), k
    want = oracle_rollout(PF, PI, mk, hp, x0.cpu(), c.cpu())
    worst = max(rel_rms(ya1[k].cpu(), want[k]) for k in want)
    print("dim64 h=4 rollout worst rel-rms", worst)
    assert worst <= TOL


def test_named_kernel_timer_counts_every_launch_of_the_hbm_bound_kernels():
    """bench.py `hbm_kernels` (north star: "HBM GB/s for the norm/activation kernels") rests on dyf_time_named_kernel_in_rollout: HIP
    events around every launch of ONE named kernel in an eager rollout.  On a small h = 4 rollout (4 forecaster + 8 interpolator
    forwards: cold sampling, refine on): the readout runs once per forward -- 12 launches -- with the algorithmic bytes of 12
    readouts; a kernel the rollout does not launch reports nothing; timing does not disturb the engine (the next sample is bitwise
    the previous one)."""
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False,
              num_input_channels=3)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(2)
    nb = 3
    x0, c = torch.randn(nb, 3, 23, 11, generator=g).to(DEV), torch.rand(nb, 2, 23, 11, generator=g).to(DEV)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=nb, use_graph=True)
    before = m.sample(x0, static_condition=c)
    eng = m._engine
    nf, ni = eng.forward_counts()
    assert (nf, ni) == (4, 8)
    seen = {}
    for name in ("readout_dma_kernel", "readout_mfma_kernel", "readout_regw_kernel", "stem16_rows_kernel", "stem16_kernel", "layernorm_c_vec_kernel"):
        ms, n, by = eng.time_named_kernel_in_rollout(name, nb)
        seen[name] = (ms, n, by)
    print({k: (round(v[0], 4), v[1], int(v[2])) for k, v in seen.items()})
    ms, n, by = seen["readout_dma_kernel"]
    # the three interpolator refine forwards of h = 4 run as ONE batched forward over 3 nb rows, the two interpolations of a cold step as one
    # paired forward: fewer launches than forwards, the same rows in total
    assert 1 <= n <= nf + ni and ms > 0
    # algorithmic bytes of one row: the (sparse-column) 64-channel decoder plane in, the native field out -- at most the dense plane
    per_row = by / ((nf + ni) * nb)
    assert 3 * 23 * 11 * 4.0 < per_row <= 64 * 64 * 64 * 2.0 + 3 * 23 * 11 * 4.0, per_row
    assert seen["layernorm_c_vec_kernel"][1] == 0 and seen["layernorm_c_vec_kernel"][2] == 0  # a ResNet-UNet kernel: not launched here
    after = m.sample(x0, static_condition=c)
    assert all(torch.equal(before[k], after[k]) for k in before)


def test_fullsize_ns_rollout_matches_reference_fields():
    """BASELINE config 2 (NS 221x42, h=16, cold, refine, dim 64 @256^2), NB=1, dropout off: t1/t8/t16 fields of the
    imported reference (fixture G6)."""
    meta, fields = jload("fullsize_checksums.json"), load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF, PI = seeded_pair(64, 3, 2, seeds=(meta["seeds"]["forecaster"], meta["seeds"]["interpolator"]))
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=2)
    out = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    assert sorted(out) == sorted(meta["rollout"])
    errs = {k: rel_rms(out[f"{k}_preds"].cpu(), fields[k]) for k in ("t1", "t8", "t16")}
    print("fullsize rollout rel-rms", errs)
    assert max(errs.values()) <= TOL
    for k, want in meta["rollout"].items():
        assert abs(float(out[k].mean()) - want["mean"]) <= 0.02 * max(1.0, want["std"]), k


def test_fullsize_engine_error_is_the_bf16_storage_error():
    """Per block of one full-size interpolator forward and over the h=16 rollout: the engine's distance from the fp32 oracle
    is the distance of the oracle's own bf16 arithmetic model (same rounding points, fp32 everywhere else) -- no layer of
    the engine adds error beyond bf16 storage.  Allowed: 1.25 x the model's error (rounding ties fall differently)."""
    meta = jload("fullsize_checksums.json")
    mk = meta["model"]
    PF, PI = seeded_pair(64, 3, 2, seeds=(meta["seeds"]["forecaster"], meta["seeds"]["interpolator"]))
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0, c = torch.randn(1, 3, 221, 42, generator=g), torch.rand(1, 2, 221, 42, generator=g)
    from tests.gpu_common import mirror_from_params
    net = mirror_from_params(PI, mk, 6, 2, 3)
    xin, t = torch.cat([x0, 0.5 * x0.flip(-1) + 0.1], 1), torch.tensor([5.0])
    y = net(xin.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    taps32, taps16 = {}, {}
    with torch.no_grad():
        y32 = nets.unet_simple_forward(PI, mk, xin, t, c, taps=taps32)
        y16 = nets.unet_simple_forward_bf16_model(PI, mk, xin, t, c, taps=taps16)
    for li, nm in enumerate([f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]):
        a = net._engine.read_block_output(0, li, 1).cpu()
        ok = torch.isfinite(a)  # the last decoder block computes only the columns the readout reads
        e_eng, e_model = rel_rms(a[ok], taps32[nm][ok]), rel_rms(taps16[nm][ok], taps32[nm][ok])
        print(f"{nm}: engine {e_eng:.2e}  bf16 model {e_model:.2e}  ({float(ok.float().mean()):.2f} of the block computed)")
        assert e_eng <= 1.25 * e_model + 2e-4, nm
    assert rel_rms(y, y32) <= 1.25 * rel_rms(y16, y32)
    hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=2)
    out = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    with torch.no_grad():
        o32 = oracle_rollout(PF, PI, mk, hp, x0, c)
        o16 = sampler.sample_loop(lambda x, tt, cond: nets.unet_simple_forward_bf16_model(PF, mk, x, tt, cond),
                                  lambda x, tt, cond: nets.unet_simple_forward_bf16_model(PI, mk, x, tt, cond), x0, c, hp)
    drift = jload("../../profiles/r02_bf16_drift.json")["results"]["autocast"]
    for k in ("t1", "t8", "t16"):
        e_eng, e_model = rel_rms(out[f"{k}_preds"].cpu(), o32[f"{k}_preds"]), rel_rms(o16[f"{k}_preds"], o32[f"{k}_preds"])
        print(f"rollout {k}: engine {e_eng:.2e}  bf16 model {e_model:.2e}  reference under bf16 autocast {drift[k]:.2e}")
        assert e_eng <= 1.25 * e_model and e_eng <= drift[k]


def test_ensemble_rows_and_batch_split_invariance():
    """Ensemble members are independent batch rows: sampling rows [0:2] and [2:4] separately gives the
    same fields as sampling all four at once (dropout off)."""
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.1)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(3)
    x0, c = torch.randn(4, 3, 23, 11, generator=g).to(DEV), torch.rand(4, 2, 23, 11, generator=g).to(DEV)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=4)
    full = {k: v.clone() for k, v in m.sample(x0, static_condition=c).items()}
    lo = {k: v.clone() for k, v in m.sample(x0[:2], static_condition=c[:2]).items()}
    hi = m.sample(x0[2:], static_condition=c[2:])
    for k in full:
        assert torch.equal(full[k][:2], lo[k]) and torch.equal(full[k][2:], hi[k]), k


def test_autoregressive_outer_loop_with_boundary_conditions():
    """forecasting_multi_horizon.py:114-229: prediction_horizon = 2 * horizon -> two engine rollouts, the last field fed
    back, a boundary-condition callable applied to every predicted field (as the NS datamodule does)."""
    import dyffusion_amd as D
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.1)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(5)
    B, N = 2, 2
    dyn = torch.randn(B, 9, 3, 23, 11, generator=g)
    cond = torch.rand(B, 2, 23, 11, generator=g)
    mask = torch.zeros(3, 23, 11, dtype=torch.bool)
    mask[:, :2, :] = True
    mask[0, :, 0] = True

    def bc(preds, targets, metadata, time):
        preds = preds.clone()
        preds[..., mask.to(preds.device)] = 0.25 * time
        return preds

    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=N * B)
    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N)
    got = exp.evaluation_step({"dynamics": dyn.to(DEV), "condition": cond.to(DEV)}, prediction_horizon=8,
                              boundary_conditions=bc)
    # oracle: same loop on the CPU
    x = dyn[:, 0].repeat(N, 1, 1, 1)
    c = cond.repeat(N, 1, 1, 1)
    t = 0.0
    worst = 0.0
    for ar in range(2):
        o = oracle_rollout(PF, PI, mk, hp, x, c)
        for k in range(1, 5):
            t += 1.0
            p = bc(o[f"t{k}_preds"].reshape(N, B, 3, 23, 11), None, None, t)
            key = f"t{ar * 4 + k}_preds"
            assert tuple(got[key].shape) == (N, B, 3, 23, 11)
            worst = max(worst, rel_rms(got[key].cpu(), p))
            last = p
        x = last.reshape(N * B, 3, 23, 11)
    assert torch.equal(got["t8_targets"].cpu(), dyn[:, 8])
    print("autoregressive (2 x h=4) worst rel-rms", worst)
    assert worst <= 2.5e-2


def test_autoregressive_outer_loop_with_window_2():
    """forecasting_multi_horizon.py:194-221 with window = 2: the last TWO predicted fields of an outer iteration, stacked on
    the channel axis, are the next iteration's initial condition; the interpolator sees (window + 1) * C channels."""
    import dyffusion_amd as D
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=False, num_input_channels=3)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.1)
    PF, PI = seeded_pair(64, 3, 2, window=2)
    g = torch.Generator().manual_seed(15)
    B, N, W = 2, 2, 2
    dyn = torch.randn(B, W + 8, 3, 23, 11, generator=g)
    cond = torch.rand(B, 2, 23, 11, generator=g)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, window=W, max_batch=N * B)
    exp = D.MultiHorizonForecastingDYffusion(m, num_predictions=N, window=W, prediction_horizon=8)
    got = exp.evaluation_step({"dynamics": dyn.to(DEV), "condition": cond.to(DEV)})
    x = dyn[:, :W].reshape(B, W * 3, 23, 11).repeat(N, 1, 1, 1)
    c = cond.repeat(N, 1, 1, 1)
    worst = 0.0
    for ar in range(2):
        o = oracle_rollout(PF, PI, mk, hp, x, c)
        for k in range(1, 5):
            worst = max(worst, rel_rms(got[f"t{ar * 4 + k}_preds"].cpu().reshape(N * B, 3, 23, 11), o[f"t{k}_preds"]))
        x = torch.cat([o["t3_preds"], o["t4_preds"]], dim=1)
    assert torch.equal(got["t8_targets"].cpu(), dyn[:, W + 7])
    print("autoregressive window=2 (2 x h=4) worst rel-rms", worst)
    assert worst <= 2.5e-2


@pytest.mark.parametrize("name", ["plosses_a", "plosses_b", "plosses_c"])
def test_forecaster_objective_matches_reference_golden(name):
    """`DYffusion.p_losses` in eval mode (dyffusion.py:496-567): per-row diffusion steps incl. t = 0 and T-1, both loss terms,
    against the imported reference's values (fixtures plosses_*.npz) and the oracle.  Tolerance: 2 % of the loss value (two
    to four chained bf16 network forwards; the reduction itself is fp32 / fp64)."""
    from oracle import losses
    z = load_npz(name + ".npz")
    hp = json.loads(str(z["hp"]))
    PF, PI = split_state(z, "F"), split_state(z, "I")
    m = build_dyffusion(PF, PI, hp["model"], 4, 1, hp, max_batch=hp["B"])
    xt_last, cond, sc = (torch.from_numpy(z[k]).to(DEV) for k in ("xt_last", "cond", "sc"))
    t = torch.from_numpy(z["t"]).to(DEV)
    got = m.p_losses(xt_last, cond, t, static_condition=sc)
    want = json.loads(str(z["losses"]))
    for k_got, k_want in (("loss", "loss"), ("val/loss_forward", "loss_forward"), ("val/loss_forward2", "loss_forward2")):
        assert abs(got[k_got] - want[k_want]) <= 2e-2 * max(abs(want[k_want]), 1e-3), (k_got, got[k_got], want[k_want])
    assert all(isinstance(v, float) for v in got.values())  # eval mode: plain floats ("val/" keys); training: tests/test_gpu_training.py


def test_criterion_reduction_matches_torch():
    import dyffusion_amd as D
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    eng = D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False)
    g = torch.Generator().manual_seed(5)
    for n in (1, 3, 1027, 80 * 3 * 221 * 42):
        a, b = torch.randn(n, generator=g), 1.5 * torch.randn(n, generator=g)
        for kind, ref in (("l1", torch.nn.functional.l1_loss), ("mse", torch.nn.functional.mse_loss),
                          ("smoothl1", torch.nn.functional.smooth_l1_loss)):
            got = eng.criterion(a.to(DEV), b.to(DEV), kind)
            want = float(ref(a.double(), b.double()))
            assert abs(got - want) <= 2e-6 * max(1.0, abs(want)), (n, kind, got, want)
    big = torch.randn(4099, generator=g).to(DEV)
    assert abs(eng.criterion(big[1:], big[:-1], "l1") - float((big[1:] - big[:-1]).abs().double().mean())) <= 1e-5  # misaligned views


def test_lightning_checkpoint_through_to_dyf_sample_matches_the_reference(tmp_path):
    """SURVEY 8f-4 end to end: a checkpoint in the layout Lightning writes for `MultiHorizonForecastingDYffusion`
    (`state_dict["model.model.*"]` = forecaster, `["model.interpolator.model.*"]` = the frozen interpolator copy; src/interface.py:115-172)
    on disk -> `load_networks_from_checkpoints` -> fresh engine networks -> `dyf_sample`, against the fields the REFERENCE produced
    with those weights (fixture sample_cold_refine, make_golden.py)."""
    from dyffusion_amd.checkpoint import load_networks_from_checkpoints
    from tests.gpu_common import mirror_from_params
    import dyffusion_amd as D

    z = load_npz("sample_cold_refine.npz")
    hp = json.loads(str(z["hp"]))
    PF, PI = split_state(z, "F"), split_state(z, "I")
    sd = {"model.model." + k: v.clone() for k, v in PF.items()}
    sd.update({"model.interpolator.model." + k: v.clone() for k, v in PI.items()})
    sd["model.interpolator.some_metric_buffer"] = torch.zeros(3)  # non-network state of the wrapped experiment: ignored
    path = tmp_path / "epoch=3-step=77.ckpt"
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 77, "pytorch-lightning_version": "2.0.0"}, path)
    mk = hp["model"]
    kw = dict(dim=mk["dim"], with_time_emb=True, upsample_dims=mk["upsample_dims"], outer_sample_mode=mk["outer_sample_mode"],
              dropout=mk["dropout"], num_output_channels=4)
    F = D.UNet(num_input_channels=4, num_conditional_channels=1, **kw)   # forward_conditioning = "none": static condition only
    I = D.UNet(num_input_channels=8, num_conditional_channels=1, **kw)
    for net in (F, I):  # start from weights that are NOT the checkpoint's
        for p_ in net.parameters():
            torch.nn.init.normal_(p_, std=0.5)
    meta = load_networks_from_checkpoints(F, I, forecaster_ckpt=str(path))
    assert meta == {"epoch": 3, "global_step": 77}
    keys = ["forward_conditioning", "schedule", "additional_interpolation_steps", "interpolate_before_t1", "sampling_type",
            "time_encoding", "refine_intermediate_predictions", "enable_interpolator_dropout"]
    m = D.DYffusion(F, D.InterpolatorHandle(I, hp["timesteps"]), timesteps=hp["timesteps"], max_batch=hp["B"],
                    **{k: hp[k] for k in keys})
    got = m.sample(torch.from_numpy(z["x0"]).to(DEV), static_condition=torch.from_numpy(z["c"]).to(DEV))
    want = {k[len("out::"):]: v for k, v in z.items() if k.startswith("out::")}
    assert sorted(got) == sorted(want)
    worst = max(rel_rms(got[k].cpu().reshape(w.shape), w) for k, w in want.items())
    print("checkpoint -> dyf_sample vs the reference's fields: worst rel-rms", worst)
    assert worst <= TOL
