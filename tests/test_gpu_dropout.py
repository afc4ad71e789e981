"""-m gpu: the engine's counter-based MC-dropout generator, in the mode bench.py runs (engine RNG, paired 2 NB-row
interpolator launches, hipGraph replay).

The keep bit of an element is a function of (seed, forward index, global batch row, layer, element) -- csrc/common.h,
restated in numpy in tests/rng_host.py (pinned to the C++ source by tests/test_rng_host.py).  The tests rebuild the masks
of a forward / of a whole rollout on the host and feed them to the oracle:

  (a) one forward, and a whole cold-sampling rollout with refinement: engine (RNG mode, pairing on, graph on) == oracle on
      the host-rebuilt masks, to the bf16 tolerance;
  (b) two replays of the captured graph draw different masks; re-seeding reproduces a replay bit for bit; pairing on/off,
      graph/eager, batch splits and row offsets all give the SAME bits (the stream is a function of the global row);
  (c) ensemble mean / variance over 256 members against the imported reference's own 256-member statistics
      (tests/golden/stats_ens256.npz), within sampling error;
  (d) BASELINE config 2 at full size with NB=80 rows, dropout on: per-horizon mean / std / ensemble spread against the
      reference's 8-member statistics (tests/golden/fullsize_dropout_stats.json).
"""
import json
import os

import numpy as np
import pytest
import torch

from oracle import init as oinit
from oracle import nets, sampler
from tests import rng_host as R
from tests.gpu_common import DEV, build_dyffusion, mirror_from_params, seeded_pair
from tests.helpers import jload, load_npz, rel_rms

pytestmark = pytest.mark.gpu

HP4 = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
           sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=True)
MK64 = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)


def test_rng_mode_forward_equals_oracle_with_host_reproduced_masks():
    p = 0.15
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=36)
    g = torch.Generator().manual_seed(10)
    x, c, t = torch.randn(2, 3, 23, 11, generator=g), torch.rand(2, 2, 23, 11, generator=g), torch.tensor([2.0, 3.0])
    net = mirror_from_params(P, MK64, 3, 2, 3)
    seed = 1234567890123
    with net.inference_dropout_scope(True):
        net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV))  # builds the engine
        net._engine.seed(seed)
        y0 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()  # forward index 0 after seeding
        y1 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()  # forward index 1
        net._engine.seed(seed)
        net._engine.set_row_offset(5)
        y0_off = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()  # same forward index, global rows 5 and 6
        net._engine.set_row_offset(0)
    assert not torch.equal(y0, y1) and not torch.equal(y0, y0_off)
    for fwd, off, y in ((0, 0, y0), (1, 0, y1), (0, 5, y0_off)):
        drop = R.EngineDropout(seed, 64, 64, 64, row_offset=off, first_forward=fwd)
        drop.begin_forward()
        with torch.no_grad():
            want = nets.unet_simple_forward(P, MK64, x, t, c, dropout=drop)
        err = rel_rms(y, want)
        print("rng-mode vs oracle(host masks): forward", fwd, "row offset", off, "rel-rms", err)
        assert err <= 1.5e-2


def test_input_dropout_forward_and_rollout_equal_the_oracle_on_the_engines_masks():
    """input_dropout > 0 (unet_simple.py:116,168: a Dropout on init_conv's output, the first site of a forward): the engine runs the
    separate stem kernel with the dropout in its epilogue (the stem -> enc0 composition needs nothing non-linear in between).  One
    forward with a row offset, and a whole rollout (paired interpolator launches, graph) against the oracle on host-rebuilt masks."""
    mk = dict(MK64, input_dropout=0.1)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=36)
    g = torch.Generator().manual_seed(10)
    x, c, t = torch.randn(2, 3, 23, 11, generator=g), torch.rand(2, 2, 23, 11, generator=g), torch.tensor([2.0, 3.0])
    net = mirror_from_params(P, mk, 3, 2, 3)
    seed = 424242
    y_eval = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    with torch.no_grad():
        assert rel_rms(y_eval, nets.unet_simple_forward(P, mk, x, t, c)) <= 1e-2  # dropout off: the un-fused stem path itself
    with net.inference_dropout_scope(True):
        net._engine.seed(seed)
        net._engine.set_row_offset(3)
        y = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
        net._engine.set_row_offset(0)
    drop = R.EngineDropout(seed, 64, 64, 64, row_offset=3, input_dropout=True)
    drop.begin_forward()
    with torch.no_grad():
        want = nets.unet_simple_forward(P, mk, x, t, c, dropout=drop)
    err = rel_rms(y, want)
    print("input_dropout forward vs oracle(host masks): rel-rms", err)
    assert err <= 1.5e-2 and rel_rms(y, y_eval) > 0.1
    # rollout
    PF, PI = seeded_pair(64, 3, 2)
    m = build_dyffusion(PF, PI, mk, 3, 2, HP4, max_batch=3)
    m.seed(99)
    x0, cc = torch.randn(3, 3, 23, 11, generator=g), torch.rand(3, 2, 23, 11, generator=g)
    got = m.sample(x0.to(DEV), static_condition=cc.to(DEV))
    uh, uw = mk["upsample_dims"]
    dr = R.EngineDropout(99, 64, uh, uw, input_dropout=True)

    def i_fn(xx, tt, cond):
        dr.begin_forward()
        return nets.unet_simple_forward(PI, mk, xx, tt, cond, dropout=dr)

    with torch.no_grad():
        want = sampler.sample_loop(lambda xx, tt, cond: nets.unet_simple_forward(PF, mk, xx, tt, cond), i_fn, x0, cc, HP4)
    worst = max(rel_rms(got[k].cpu(), want[k]) for k in want)
    print("input_dropout rollout vs oracle(host masks): worst rel-rms", worst)
    assert worst <= 2.5e-2


def _oracle_rollout_with_engine_masks(PF, PI, mk, hp, x0, c, seed, row_offset=0, first_forward=0):
    """oracle.sampler.sample_loop with the interpolator drawing the ENGINE's masks: the forward counter advances by one
    per interpolator forward, in the order sample_loop makes them (next-step, then current-step, then the refinement
    pass) -- which is the order of the engine's launches, a paired launch being two consecutive forwards."""
    uh, uw = mk["upsample_dims"]
    drop = R.EngineDropout(seed, mk["dim"], uh, uw, row_offset=row_offset, first_forward=first_forward)

    def i_fn(x, t, cond):
        drop.begin_forward()
        return nets.unet_simple_forward(PI, mk, x, t, cond, dropout=drop)

    with torch.no_grad():
        out = sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward(PF, mk, x, t, cond), i_fn, x0, c, hp)
    return out, drop.fwd + 1


@pytest.mark.parametrize("max_batch", [3, 24], ids=["paired-refine", "batched-refine"])
@pytest.mark.parametrize("hp_extra", [dict(), dict(additional_interpolation_steps=2, timesteps=5, use_cold_sampling_for_last_step=True)],
                         ids=["h4", "h5k2_coldlast"])
def test_rng_mode_rollout_equals_oracle_with_host_reproduced_masks(hp_extra, max_batch):
    """(a) The benchmarked mode: engine RNG + paired interpolator launches + hipGraph.  Two consecutive sample() calls:
    the second continues the forward counter where the first stopped (fresh masks on graph replay).
    max_batch = 24 (round 4): the workspace holds 48 rows, so the refinement pass of the 3-row call -- h - 1 interpolator forwards
    that share their inputs -- goes out as ONE forward over (h - 1) * 3 rows (engine.hip run_plan) instead of pairs; row r must
    draw the masks of forward counter + r / 3, exactly what the oracle's sequential calls draw."""
    hp = dict(HP4, **hp_extra)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(21)
    nb = 3
    x0, c = torch.randn(nb, 3, 23, 11, generator=g), torch.rand(nb, 2, 23, 11, generator=g)
    m = build_dyffusion(PF, PI, MK64, 3, 2, hp, max_batch=max_batch, use_graph=True)
    seed = 987654321
    m.seed(seed)
    got1 = {k: v.cpu() for k, v in m.sample(x0.to(DEV), static_condition=c.to(DEV)).items()}
    got2 = {k: v.cpu() for k, v in m.sample(x0.to(DEV), static_condition=c.to(DEV)).items()}  # graph replay
    want1, nfwd = _oracle_rollout_with_engine_masks(PF, PI, MK64, hp, x0, c, seed)
    want2, _ = _oracle_rollout_with_engine_masks(PF, PI, MK64, hp, x0, c, seed, first_forward=nfwd)
    n_f, n_i = m._engine.forward_counts()
    assert nfwd == n_i
    for tag, got, want in (("first call", got1, want1), ("graph replay", got2, want2)):
        assert sorted(got) == sorted(want)
        worst = max(rel_rms(got[k], want[k]) for k in want)
        print(f"rng-mode rollout ({tag}) worst rel-rms vs oracle on host-rebuilt masks: {worst:.3e}")
        assert worst <= 2e-2
    # the two calls really drew different masks (member spread is O(0.2), far above the numeric noise)
    assert min(rel_rms(got1[k], got2[k]) for k in got1) > 5e-2


def test_rng_stream_is_invariant_to_pairing_graph_batching_and_row_offset(form_switch):
    """(b) same seed -> same bits, however the rows are launched."""
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(22)
    nb = 4
    x0, c = torch.randn(nb, 3, 23, 11, generator=g).to(DEV), torch.rand(nb, 2, 23, 11, generator=g).to(DEV)

    def run(use_graph, rows=slice(0, nb), offset=0, calls=1):
        m = build_dyffusion(PF, PI, MK64, 3, 2, HP4, max_batch=nb, use_graph=use_graph)
        m.seed(42)
        m.set_row_offset(offset)
        outs = [{k: v.clone() for k, v in m.sample(x0[rows], static_condition=c[rows]).items()} for _ in range(calls)]
        return outs

    ref = run(True, calls=2)
    assert not torch.equal(ref[0]["t2_preds"], ref[1]["t2_preds"])  # a replay draws fresh masks
    again = run(True, calls=2)
    for a, b in zip(ref, again):  # re-seeding reproduces both calls bit for bit
        assert all(torch.equal(a[k], b[k]) for k in a)
    eager = run(False)[0]
    assert all(torch.equal(ref[0][k], eager[k]) for k in eager), "graph replay != eager launch"
    lo, hi = run(True, rows=slice(0, 2))[0], run(True, rows=slice(2, 4), offset=2)[0]
    for k in eager:  # two "ranks" of two rows each == the four-row batch (invariant to the number of GPUs)
        assert torch.equal(ref[0][k][:2], lo[k]) and torch.equal(ref[0][k][2:], hi[k]), k
    form_switch.setenv("DYF_PAIR_INTERP", "0")  # one launch per interpolator forward instead of paired 2 nb-row launches
    unpaired = run(True)[0]
    assert all(torch.equal(ref[0][k], unpaired[k]) for k in unpaired), "paired launch != two separate forwards"


def test_ensemble_statistics_match_the_reference():
    """(c) 256 members (MC dropout in the interpolator), dim-64 pair on 23x11: per-pixel ensemble mean and variance against
    the imported reference's own 256-member ensemble (torch Bernoulli stream).  With independent ensembles of N members,
    (mean_a - mean_b) / sqrt((var_a + var_b) / N) is ~N(0,1) per pixel: its RMS over all pixels must be ~1 (a biased
    engine gives >> 1); the ratio of the pixel-averaged variances must be 1 within a few per cent."""
    z = load_npz("stats_ens256.npz")
    hp_meta = json.loads(str(z["hp"]))
    N = int(z["n_members"])
    PF, PI = seeded_pair(64, 3, 2, seeds=(hp_meta["seeds"]["forecaster"], hp_meta["seeds"]["interpolator"]))
    x0, c = torch.from_numpy(z["x0"]), torch.from_numpy(z["c"])
    m = build_dyffusion(PF, PI, hp_meta["model"], 3, 2, HP4, max_batch=N)
    m.seed(2024)
    out = m.sample(x0.repeat(N, 1, 1, 1).to(DEV), static_condition=c.repeat(N, 1, 1, 1).to(DEV))
    for k in sorted(out):
        v = out[k].double().cpu()
        mean_e, var_e = v.mean(0), v.var(0, unbiased=True)
        mean_r, var_r = torch.from_numpy(z[f"mean::{k}"]).double(), torch.from_numpy(z[f"var::{k}"]).double()
        zscore = (mean_e - mean_r) / ((var_e + var_r) / N).sqrt()
        z_rms = float(zscore.pow(2).mean().sqrt())
        ratio = float(var_e.mean() / var_r.mean())
        print(f"{k}: z-score RMS of the ensemble means {z_rms:.3f} (1 = sampling error only), variance ratio {ratio:.4f}")
        assert 0.75 <= z_rms <= 1.6, (k, z_rms)
        assert 0.90 <= ratio <= 1.10, (k, ratio)
    # the 256-member stack scored on the device (dyf_ensemble_metrics beyond 64 members) against the numpy oracle, with the
    # reference ensemble's mean standing in for the truth
    from dyffusion_amd.metrics import evaluate_ensemble_prediction
    from oracle import metrics as om
    k = sorted(out)[-1]
    truth = torch.from_numpy(z[f"mean::{k}"]).float()
    stack = out[k].reshape(N, 1, *truth.shape)  # (members, samples, C, H, W)
    got = evaluate_ensemble_prediction(stack, truth[None].to(DEV), m._engine)
    ref = om.evaluate_ensemble_prediction(stack.cpu().numpy(), truth[None].numpy())
    for name in ("mse", "ssr", "crps"):
        assert got[name] == pytest.approx(ref[name], rel=5e-5), (name, got[name], ref[name])


def test_fullsize_rollout_with_dropout_matches_reference_statistics():
    """(d) BASELINE config 2 exactly as bench.py runs it (NB=80 rows, engine RNG, pairing, hipGraph) on the fixture-G6 inputs:
    per-horizon mean / std over members and pixels, ensemble-mean std and ensemble spread against the reference's 8-member
    statistics; every member finite and distinct."""
    meta = jload("fullsize_dropout_stats.json")
    PF, PI = seeded_pair(64, 3, 2, seeds=(meta["seeds"]["forecaster"], meta["seeds"]["interpolator"]))
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0, c = torch.randn(1, 3, 221, 42, generator=g), torch.rand(1, 2, 221, 42, generator=g)
    hp = dict(HP4, timesteps=16)
    nb = 80
    m = build_dyffusion(PF, PI, meta["model"], 3, 2, hp, max_batch=nb)
    m.seed(7)
    out = m.sample(x0.repeat(nb, 1, 1, 1).to(DEV), static_condition=c.repeat(nb, 1, 1, 1).to(DEV))
    assert sorted(out) == sorted(meta["rollout"])
    for k, want in meta["rollout"].items():
        v = out[k].double()
        assert bool(torch.isfinite(v).all())
        got = dict(mean=float(v.mean()), std=float(v.std()), spread=float(v.var(0, unbiased=True).mean().sqrt()),
                   ens_mean_std=float(v.mean(0).std()))
        print(k, {q: (round(got[q], 4), round(want[q], 4)) for q in got})
        assert abs(got["mean"] - want["mean"]) <= 0.03 * want["std"], k
        assert abs(got["std"] / want["std"] - 1) <= 0.05, k
        # the reference's spread comes from 8 members: its own sampling error is a few per cent after pixel averaging
        assert abs(got["spread"] / want["spread"] - 1) <= 0.10, k
    assert float((out["t16_preds"][0] - out["t16_preds"][1]).abs().max()) > 0  # members differ
