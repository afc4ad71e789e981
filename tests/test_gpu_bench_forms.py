"""-m gpu: the kernel FORMS bench.py runs, held to the oracle.

Kernel forms are chosen per launch from tile counts (csrc/conv.hip launch_conv): a full-size parity test at NB = 1..3 runs
the implicit-GEMM / split-K forms for enc1-enc3 and dec2, while bench.py at NB = 80 (paired interpolator launches: 160 rows)
runs conv_up_halo_kernel<3/4> (stride-2 halo), conv_halo_rows_kernel<0/1/2>, the persistent conv_enc0_stem_kernel and no
split-K.  These tests run BASELINE configs[1] exactly as bench.py builds it -- NB = 80 rows, hipGraph, paired launches -- on 80
DISTINCT inputs, assert from the engine's form log that those kernels were the ones launched, and compare rows of the result
with CPU oracle rollouts of the same rows (reference: src/diffusion/dyffusion.py:335-426, src/models/unet_simple.py:181-197):

  (a) MC dropout off: rows {0, 79} of t1 / t8 / t16, first call and graph replay, rel-RMS <= 2.5e-2 (the bf16
      bound of tests/test_gpu_sampler.py);
  (b) MC dropout on (engine generator): row 79 against the oracle drawing the engine's masks rebuilt on the host
      (tests/rng_host.EngineDropout with the row's global index);
  (c) one interpolator forward over 160 rows (the paired launch size): block outputs of rows {0, 79, 80, 159} against the
      oracle's taps, <= 1.25 x the oracle's own bf16 storage model per block.
"""
import pytest
import torch

from oracle import nets, sampler
from tests import rng_host as R
from tests.gpu_common import DEV, build_dyffusion, cached, mirror_from_params, oracle_rollout, seeded_pair
from tests.helpers import jload, rel_rms

pytestmark = pytest.mark.gpu
TOL = 2.5e-2
NB = 80
HP = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
          sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
# forms bench.py's engine must take at NB = 80: {form: batch rows of its launches}; 160 = paired interpolator launches
BENCH_FORMS = {
    "conv_enc0_stem_kernel": {80, 160},       # enc0 on the fused stem
    "conv_up_halo_kernel<4>": {80, 160},      # enc1 (4x4 s2, 128 channels)
    "conv_up_halo_kernel<3>": {80, 160},      # enc2, enc3 (4x4 s2, 256 / 512 channels)
    "conv_halo_rows_kernel<2>": {80, 160},    # dec2 (plain 3x3, 256-channel blocks)
    "conv_halo_rows_kernel<0>": {80, 160},    # dec3, dec4 (fused x2 upsample + 3x3)
    "conv_halo_rows_kernel<1>": {80, 160},    # dec5 (sparse output columns)
    "stem16_rows_kernel": {80, 160},
    "readout_dma_kernel": {80, 160},
    "up2x_quad_kernel": {80, 160},
}


def _setup():
    meta = jload("fullsize_checksums.json")
    mk = meta["model"]
    PF, PI = seeded_pair(64, 3, 2, seeds=(meta["seeds"]["forecaster"], meta["seeds"]["interpolator"]))
    g = torch.Generator().manual_seed(4242)
    x0, c = torch.randn(NB, 3, 221, 42, generator=g), torch.rand(NB, 2, 221, 42, generator=g)
    return mk, PF, PI, x0, c


def _assert_bench_forms(forms):
    print("kernel forms launched:", {k: sorted(v) for k, v in forms.items()})
    for form, rows in BENCH_FORMS.items():
        assert form in forms, f"{form} was not launched: {sorted(forms)}"
        assert rows <= set(forms[form]), (form, forms[form])
    # the implicit-GEMM kernels (with split-K where a launch has <= 128 tiles) serve only the small planes: enc4, enc5, dec0,
    # dec1 (16^2 .. 4^2) -- 4 launches per forward; every conv of the 256^2 .. 32^2 planes went to one of the forms above
    n_fwd = sum(forms["readout_dma_kernel"].values())
    # (conv_skinny_kernel, round 4: the launches of <= 64 tiles -- enc5 and dec0 of an unpaired 80-row forward -- instead of split-K)
    small = sum(sum(v.values()) for f, v in forms.items() if f.startswith("conv_igemm") or f == "conv_skinny_kernel")
    assert small == 4 * n_fwd, (small, n_fwd, forms)
    assert "conv_direct_kernel" not in forms


def oracle_nb80_rows(mk, PF, PI, x0, c, rows):
    """fp32 oracle rollouts of rows of the 80-row benchmark batch, MC dropout off (t1 / t8 / t16 are what the test compares)."""
    def run():
        out = oracle_rollout(PF, PI, mk, HP, x0[rows], c[rows])
        return {k: out[k] for k in ("t1_preds", "t8_preds", "t16_preds")}
    return cached("ns80_rows_" + "_".join(map(str, rows)), run, depends_on=[x0[rows], c[rows], PF, PI], config=dict(model=mk, hp=HP, rows=list(rows)))


def oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, seed, r):
    """fp32 oracle rollout of ONE row of the 80-row batch with the interpolator drawing the ENGINE's MC-dropout masks of global row r,
    rebuilt on the host (tests/rng_host.py)."""
    uh, uw = mk["upsample_dims"]

    def run():
        drop = R.EngineDropout(seed, mk["dim"], uh, uw, row_offset=r)

        def i_fn(x, t, cond):
            drop.begin_forward()
            return nets.unet_simple_forward(PI, mk, x, t, cond, dropout=drop)

        with torch.no_grad():
            out = sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward(PF, mk, x, t, cond), i_fn, x0[r:r + 1], c[r:r + 1], hp)
        return {k: out[k] for k in ("t1_preds", "t8_preds", "t16_preds")}
    return cached(f"ns80_dropout_seed{seed}_row{r}", run,
                  depends_on=[x0[r:r + 1], c[r:r + 1], PF, PI], config=dict(model=mk, hp=hp, seed=int(seed), row=int(r)))


def test_nb80_graph_paired_rollout_rows_match_the_oracle():
    """(a)"""
    mk, PF, PI, x0, c = _setup()
    m = build_dyffusion(PF, PI, mk, 3, 2, HP, max_batch=NB, use_graph=True)
    m._ensure_engine((221, 42), NB)
    eng = m._engine
    eng.form_log(True)
    first = {k: v.clone() for k, v in m.sample(x0.to(DEV), static_condition=c.to(DEV)).items()}  # captures the graph
    forms = eng.form_log_read()
    eng.form_log(False)
    _assert_bench_forms(forms)
    replay = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    for k in first:
        assert torch.equal(first[k], replay[k]), k
    rows = [0, 79]  # (an oracle rollout of 60 forwards at 256^2 is 20-30 s of CPU time per row: disk-cached, tests/gpu_common.cached)
    want = oracle_nb80_rows(mk, PF, PI, x0, c, rows)
    worst = 0.0
    for k in ("t1_preds", "t8_preds", "t16_preds"):
        for j, r in enumerate(rows):
            e = rel_rms(replay[k][r].cpu(), want[k][j])
            print(f"NB=80 graph+paired, dropout off: {k} row {r}: rel-RMS {e:.3e}")
            worst = max(worst, e)
    assert worst <= TOL


# kernel forms a FEW-ROW rollout must take on an engine sized for 80 rows (round 5, DESIGN.md 4.9: what one GPU runs when the 80-row
# evaluation batch / a 50-member ensemble is sharded 8 ways): {rows of the call: forms that must appear}
SMALL_FORMS = {
    1: ["up_border_split_kernel", "up2x_epilogue_kernel", "conv_halo_rows_kernel<0>+splitk", "conv_halo_rows_kernel<0>+tr2",
        "groupnorm_wave_kernel", "conv_skinny_kernel"],
    7: ["up_border_split_kernel", "up2x_epilogue_kernel", "conv_halo_rows_kernel<0>+tr2", "conv_halo_rows_kernel<2>+splitk", "groupnorm_wave_kernel"],
    10: ["up_border_split_kernel", "up2x_epilogue_kernel", "conv_halo_rows_kernel<0>+tr2", "conv_halo_rows_kernel<2>+splitk", "groupnorm_wave_kernel",
         "conv_skinny_kernel<8>"],
}


@pytest.mark.parametrize("nb", [1, 7, 10])
def test_few_row_rollouts_on_the_80_row_engine_match_the_oracle(nb):
    """the forms of the few-rows regime pinned to the oracle.  The call's rows are rows of the 80-row benchmark batch
    (rows are independent, so the cached oracle rollouts of rows 0 and 79 are the targets): (a) MC dropout off -- rows 0 and 79 first,
    filler rows behind them; (b) the benchmarked mode -- rows 80 - nb .. 79 at row offset 80 - nb with the seed of the 80-row test, so
    the LAST row draws exactly the masks of global row 79, rebuilt on the host for the oracle (tests/rng_host.py)."""
    mk, PF, PI, x0, c = _setup()
    m = build_dyffusion(PF, PI, mk, 3, 2, HP, max_batch=NB, use_graph=True)
    m._ensure_engine((221, 42), NB)
    eng = m._engine
    order = ([0, 79] + list(range(1, nb - 1)))[:nb] if nb > 1 else [79]
    xs, cs = x0[order].to(DEV), c[order].to(DEV)
    eng.form_log(True)
    first = {k: v.clone() for k, v in m.sample(xs, static_condition=cs).items()}
    forms = eng.form_log_read()
    eng.form_log(False)
    print(f"{nb} rows:", {k: sorted(v) for k, v in forms.items()})
    for f in SMALL_FORMS[nb]:
        assert f in forms, (f, sorted(forms))
    replay = m.sample(xs, static_condition=cs)
    for k in first:
        assert torch.equal(first[k], replay[k]), k
    want = oracle_nb80_rows(mk, PF, PI, x0, c, [0, 79])
    worst = 0.0
    for k in ("t1_preds", "t8_preds", "t16_preds"):
        for pos, r in enumerate(order[:2]):
            e = rel_rms(replay[k][pos].cpu(), want[k][0 if r == 0 else 1])
            print(f"{nb} rows, dropout off: {k} batch row {pos} (= row {r} of the 80): rel-RMS {e:.3e}")
            worst = max(worst, e)
    assert worst <= TOL
    eng.close()
    # (b) engine MC dropout
    hp = dict(HP, enable_interpolator_dropout=True)
    seed = 20260929
    m2 = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=NB, use_graph=True)
    m2.seed(seed)
    m2.set_row_offset(NB - nb)
    rows = list(range(NB - nb, NB))
    got = m2.sample(x0[rows].to(DEV), static_condition=c[rows].to(DEV))
    want = oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, seed, 79)
    for k in ("t1_preds", "t8_preds", "t16_preds"):
        e = rel_rms(got[k][nb - 1].cpu(), want[k][0])
        print(f"{nb} rows, engine MC dropout, last row = global row 79: {k} rel-RMS {e:.3e}")
        assert e <= TOL, (k, e)
    m2._engine.close()


def test_nb80_graph_paired_rollout_with_mc_dropout_rows_match_the_oracle_on_the_engines_masks():
    """(b) the benchmarked mode itself: MC dropout drawn by the engine's generator inside the captured, paired rollout."""
    mk, PF, PI, x0, c = _setup()
    hp = dict(HP, enable_interpolator_dropout=True)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=NB, use_graph=True)
    seed = 20260929
    m.seed(seed)
    m._ensure_engine((221, 42), NB)
    eng = m._engine
    eng.form_log(True)
    got = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    forms = eng.form_log_read()
    eng.form_log(False)
    _assert_bench_forms(forms)
    for r in (79,):
        want = oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, seed, r)
        for k in ("t1_preds", "t8_preds", "t16_preds"):
            e = rel_rms(got[k][r].cpu(), want[k][0])
            print(f"NB=80 graph+paired, MC dropout on: {k} row {r}: rel-RMS {e:.3e}")
            assert e <= TOL, (k, r, e)


def test_nb160_forward_block_outputs_match_the_oracle_taps():
    """(c) per block, at the row count of a paired interpolator launch."""
    mk, PF, PI, x0, c = _setup()
    g = torch.Generator().manual_seed(77)
    n = 2 * NB
    xin = torch.randn(n, 6, 221, 42, generator=g)
    cc = torch.rand(n, 2, 221, 42, generator=g)
    t = torch.arange(n, dtype=torch.float32) % 15 + 1
    net = mirror_from_params(PI, mk, 6, 2, 3)
    net._own_engine(n, (221, 42))
    eng = net._engine
    eng.form_log(True)
    y = net(xin.to(DEV), time=t.to(DEV), condition=cc.to(DEV)).cpu()
    forms = eng.form_log_read()
    eng.form_log(False)
    for form in BENCH_FORMS:
        assert form in forms and n in forms[form], (form, forms.get(form))
    rows = [0, 79, 80, 159]
    taps32, taps16 = {}, {}
    with torch.no_grad():
        y32 = nets.unet_simple_forward(PI, mk, xin[rows], t[rows], cc[rows], taps=taps32)
        y16 = nets.unet_simple_forward_bf16_model(PI, mk, xin[rows], t[rows], cc[rows], taps=taps16)
    for li, nm in enumerate([f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]):
        a = eng.read_block_output(0, li, n)[rows].cpu()
        ok = torch.isfinite(a)  # the last decoder block computes only the columns the readout reads
        assert bool(ok.any())
        for j, r in enumerate(rows):
            e_eng = rel_rms(a[j][ok[j]], taps32[nm][j][ok[j]])
            e_model = rel_rms(taps16[nm][j][ok[j]], taps32[nm][j][ok[j]])
            print(f"{nm} row {r}: engine {e_eng:.2e}  bf16 model {e_model:.2e}")
            assert e_eng <= 1.25 * e_model + 2e-4, (nm, r)
    for j, r in enumerate(rows):
        assert rel_rms(y[r], y32[j]) <= 1.25 * rel_rms(y16[j], y32[j]) + 2e-4, r


# ------------------------------------------------------------------------------------------------ BASELINE configs[2] (OISST)
# The same question for the ResNet-UNet path: at NB = 300 (bench.py's config2_oisst line) the level-0 / level-1 3x3 convs run on
# conv_up_halo_kernel<5, 2> and the 15 x 15 level on conv_igemm2_kernel<2, true>, both with the GroupNorm FUSED into their epilogue
# (round 4, csrc/gn_fused.h: in-launch statistics exchange between the workgroups of a sample, no GroupNorm kernel at all) -- none
# of which a 2-row parity test launches.  With INJECTED dropout masks the engine keeps to the un-fused chain (statistics from the
# conv epilogue + gn_apply_part_kernel, gn_stats + gn_apply on the 15 x 15 level).  Reference: src/models/unet.py:58-109, 266-315.
OISST_FORMS = ["conv_gn16_kernel+gn_fused"]  # (round 6: the 16 x 16-tile form took every fused conv -- 60^2 / 30^2 / 15^2, single and two-source -- from conv_up_halo_kernel<5, 2> / conv_igemm2_kernel<2, true>)
OISST_FORMS_UNFUSED = ["conv_up_halo_kernel<5>", "conv_igemm2_kernel<2>", "gn_apply_part_kernel", "gn_stats_kernel+gn_apply"]
GN_KERNELS = ["gn_apply_part_kernel", "gn_stats_kernel+gn_apply", "gn_finalize_part_kernel+gn_apply"]
OISST_TOL = {"fp16": (4e-3, 1e-2), "bf16": (2e-2, None)}  # (per forward, per field over the T=32 rollout: fp16 only)


def _oisst_setup(block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0):
    import json as _json

    from tests.helpers import load_npz
    from tests.test_gpu_unet_resnet import seeded_unet
    z = load_npz("fullsize_oisst_fields.npz")
    meta = _json.loads(str(z["meta"]))
    cfg = dict(meta["model"], resnet_block_groups=8, input_dropout=0.0, upsample_dims=None, block_dropout=block_dropout,
               block_dropout1=block_dropout1, attn_dropout=attn_dropout)
    PF = seeded_unet(64, (1, 2, 4), 2, 1, seed=meta["seeds"]["forecaster"])
    PI = seeded_unet(64, (1, 2, 4), 2, 1, seed=meta["seeds"]["interpolator"])
    return cfg, PF, PI


def oisst_fwd_case():
    """Inputs of the 300-row forward test; the oracle evaluates every 5th row and the last one (rows are independent; 300 rows of fp32
    CPU forward are a minute)."""
    cfg, _, PI = _oisst_setup(block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.2)
    nb = 300
    g = torch.Generator().manual_seed(31)
    x, t = torch.randn(nb, 2, 60, 60, generator=g), (torch.arange(nb) % 7 + 1).float() * 0.5
    return cfg, PI, x, t, sorted(set(range(0, nb, 5)) | {nb - 1})


def oracle_oisst_fwd_eval(cfg, PI, x, t, rows):
    def run():
        with torch.no_grad():
            return nets.resnet_unet_forward(PI, cfg, x[rows], t[rows], None)
    return cached("oisst300_fwd_eval", run, depends_on=[x[rows], t[rows], PI], config=dict(model=cfg, rows=list(rows)))


OISST_ROLLOUT_HP = dict(timesteps=7, schedule="before_t1_only", additional_interpolation_steps=25, interpolate_before_t1=True,
                        sampling_type="cold", refine_intermediate_predictions=False, forward_conditioning="data+noise",
                        time_encoding="dynamics", enable_interpolator_dropout=False)


def oisst_rollout_case():
    cfg, PF, PI = _oisst_setup()
    nb = 300
    g = torch.Generator().manual_seed(33)
    x0 = torch.randn(nb, 1, 60, 60, generator=g)
    noise = torch.randn(32, nb, 1, 60, 60, generator=g)
    return cfg, PF, PI, x0, noise, [150, 299]


def oracle_oisst_rollout_rows(cfg, PF, PI, x0, noise, rows):
    def run():
        it = iter(range(32))
        with torch.no_grad():
            return sampler.sample_loop(lambda x, t, cnd: nets.resnet_unet_forward(PF, cfg, x, t, cnd),
                                       lambda x, t, cnd: nets.resnet_unet_forward(PI, cfg, x, t, cnd), x0[rows], None, OISST_ROLLOUT_HP,
                                       noise_fn=lambda tensor: noise[next(it)][rows])
    return cached("oisst300_rollout_rows", run, depends_on=[x0[rows], noise[:, rows], PF, PI], config=dict(model=cfg, hp=OISST_ROLLOUT_HP, rows=list(rows)))


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_oisst_nb300_forward_with_injected_dropout_matches_the_oracle(dtype):
    """One interpolator forward over 300 DISTINCT rows, eval and with every dropout site active (masks recorded from the oracle's
    seeded draws and injected): every 5th row against the oracle."""
    from tests.test_gpu_unet_resnet import engine_masks, mirror
    cfg, PI, x, t, rows = oisst_fwd_case()
    nb = x.shape[0]
    net = mirror(PI, cfg, 2, 0, 1, dtype)
    net._own_engine(nb, (60, 60))
    eng = net._engine
    eng.form_log(True)
    got = net(x.to(DEV), time=t.to(DEV)).cpu()
    forms = eng.form_log_read()
    eng.form_log(False)
    print("kernel forms launched:", {k: sorted(v) for k, v in forms.items()})
    for f in OISST_FORMS:
        assert f in forms and nb in forms[f], (f, forms.get(f))
    assert not any(f in forms for f in GN_KERNELS), sorted(forms)  # every GroupNorm of the forward ran inside its conv
    want = oracle_oisst_fwd_eval(cfg, PI, x, t, rows)
    errs = torch.tensor([rel_rms(got[r], want[j]) for j, r in enumerate(rows)])
    print(f"OISST NB=300 forward ({dtype}), eval: rel-RMS per row max {float(errs.max()):.3e} mean {float(errs.mean()):.3e}")
    assert float(errs.max()) <= OISST_TOL[dtype][0]

    def with_masks():
        src = nets.DropoutSeeded(17, record=True)
        with torch.no_grad():
            return nets.resnet_unet_forward(PI, cfg, x[rows], t[rows], None, dropout=src), src.masks

    want, masks_sub = cached("oisst300_fwd_drop", with_masks)
    # masks of the whole 300-row launch: the oracle's recorded ones on its rows, random keep bits elsewhere
    gm = torch.Generator().manual_seed(18)
    masks = []
    for m_ in masks_sub:
        full = (torch.rand((nb,) + tuple(m_.shape[1:]), generator=gm) < 0.7).to(torch.uint8)
        full[rows] = m_
        masks.append(full)
    eng.form_log(True)
    got = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=2, masks=engine_masks(masks, 3)).cpu()
    forms = eng.form_log_read()
    eng.form_log(False)
    for f in OISST_FORMS_UNFUSED:
        assert f in forms and nb in forms[f], (f, forms.get(f))
    errs = torch.tensor([rel_rms(got[r], want[j]) for j, r in enumerate(rows)])
    print(f"OISST NB=300 forward ({dtype}), dropout injected: rel-RMS per row max {float(errs.max()):.3e} mean {float(errs.mean()):.3e}")
    assert float(errs.max()) <= OISST_TOL[dtype][0]


@pytest.mark.parametrize("dtype", ["fp16"])  # (a 93-forward ResNet-UNet plan is refused in bf16: test_gpu_unet_resnet.py)
def test_oisst_nb300_rollout_rows_match_the_oracle(dtype):
    """The T = 32 rollout (32 forecaster + 61 interpolator forwards, data+noise with injected normal draws, dropout off) over 300
    distinct rows: rows {150, 299} of all seven fields against oracle rollouts of those rows."""
    import dyffusion_amd as D
    from tests.test_gpu_unet_resnet import mirror
    cfg, PF, PI, x0, noise, rows = oisst_rollout_case()
    nb = x0.shape[0]
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=nb, dtype=dtype, **OISST_ROLLOUT_HP)
    m._ensure_engine((60, 60), nb)
    eng = m._engine
    eng.form_log(True)
    _, got, _ = m.sample_loop(x0.to(DEV), _noise=noise.to(DEV))
    forms = eng.form_log_read()
    eng.form_log(False)
    for f in OISST_FORMS:
        assert f in forms and nb in forms[f], (f, forms.get(f))
    assert not any(f in forms for f in GN_KERNELS), sorted(forms)
    want = oracle_oisst_rollout_rows(cfg, PF, PI, x0, noise, rows)
    assert sorted(got) == sorted(want)
    worst = 0.0
    for k in sorted(want):
        for j, r in enumerate(rows):
            worst = max(worst, rel_rms(got[k][r].cpu(), want[k][j]))
    print(f"OISST NB=300 rollout ({dtype}): worst rel-RMS over rows {rows} and 7 fields {worst:.3e}")
    assert worst <= OISST_TOL[dtype][1]
