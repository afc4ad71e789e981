"""-m gpu: row groups (dyf_set_row_groups, ABI 5) -- a sampling call split into G concurrent rollouts of NB / G rows,
each on its own stream with its own captured graph.  This is the form bench.py's `config2_oisst` line runs (ResNet-UNet, 300 rows,
3 groups by default), so it is pinned here: against the oracle (rows of all groups, rollout with forward_conditioning="data"),
and against the ungrouped rollout of the same engine build with MC dropout and the engine's own noise draws on -- the generator
streams (seed, forward counter, GLOBAL row) must be those of the ungrouped call.  Also: uneven splits, the sampler state read
back share by share, the counters a grouped call leaves behind, the unet_simple backbone with explicit groups, call order rules."""
import pytest
import torch

import dyffusion_amd as D
from dyffusion_amd.engine import HipEngine
from oracle import nets, sampler
from tests.gpu_common import cached, DEV, build_dyffusion, oracle_rollout, seeded_pair
from tests.helpers import rel_rms

pytestmark = pytest.mark.gpu

OISST_HP = dict(timesteps=7, schedule="before_t1_only", additional_interpolation_steps=25, interpolate_before_t1=True,
                sampling_type="cold", refine_intermediate_predictions=False, time_encoding="dynamics")


def _oisst_pair(**drop):
    from tests.test_gpu_bench_forms import _oisst_setup
    from tests.test_gpu_unet_resnet import mirror
    cfg, PF, PI = _oisst_setup(**drop)
    return cfg, PF, PI, mirror


def sharded_group_model(max_batch, total=240):
    """(model, x0): the OISST-shaped pair with every dropout site on, `total` input rows, an engine sized for `max_batch` rows
    (>= 120: three row groups).  Shared with tests/test_gpu_distributed.py (two ranks of 120 rows on one GPU)."""
    cfg, PF, PI, mirror = _oisst_pair(block_dropout=0.3, block_dropout1=0.2, attn_dropout=0.1)
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    hp = dict(OISST_HP, forward_conditioning="data+noise", enable_interpolator_dropout=True, additional_interpolation_steps=3)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=max_batch, **hp)
    m.seed(99)
    x0 = torch.randn(total, 1, 60, 60, generator=torch.Generator().manual_seed(5))
    return m, x0


def test_default_groups_follow_architecture_and_size():
    cfg, PF, PI, mirror = _oisst_pair()
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    for nb, want in [(300, 3), (150, 3), (80, 3), (64, 1), (40, 1)]:  # three groups from 72 rows on (engine.hip default_row_groups)
        m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=nb, forward_conditioning="data+noise", **OISST_HP)
        m._ensure_engine((60, 60), nb)
        assert m._engine.row_groups == want, (nb, m._engine.row_groups)
        m._engine.close()
    PF2, PI2 = seeded_pair(64, 3, 2)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only", sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=True)
    ns = build_dyffusion(PF2, PI2, mk, 3, 2, hp, max_batch=80)
    ns._ensure_engine((23, 11), 80)
    assert ns._engine.row_groups == 1  # unet_simple: no groups unless asked for


def grouped_case():
    cfg, PF, PI, mirror = _oisst_pair()
    hp = dict(OISST_HP, forward_conditioning="data", enable_interpolator_dropout=False)
    x0 = torch.randn(300, 1, 60, 60, generator=torch.Generator().manual_seed(33))
    return cfg, PF, PI, mirror, hp, x0, [0, 100, 299]  # a row of each of the three groups


def oracle_grouped_rows(cfg, PF, PI, hp, x0, rows):
    def run():
        with torch.no_grad():
            return sampler.sample_loop(lambda x, t, cnd: nets.resnet_unet_forward(PF, cfg, x, t, cnd),
                                       lambda x, t, cnd: nets.resnet_unet_forward(PI, cfg, x, t, cnd), x0[rows], None, hp)
    return cached("oisst300_grouped_rows", run, depends_on=[x0[rows], PF, PI], config=dict(model=cfg, hp=hp, rows=list(rows)))


@pytest.mark.parametrize("dtype", ["fp16"])  # (a 93-forward ResNet-UNet plan is refused in bf16: test_gpu_unet_resnet.py)
def test_oisst_nb300_grouped_rollout_rows_match_the_oracle(dtype):
    """300 rows on 3 groups of 100 (the benchmarked split): rows of every group, all 7 fields, against oracle rollouts of those rows."""
    from tests.test_gpu_bench_forms import OISST_TOL
    cfg, PF, PI, mirror, hp, x0, rows = grouped_case()
    nb = x0.shape[0]
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=nb, dtype=dtype, **hp)
    m._ensure_engine((60, 60), nb)
    eng = m._engine
    assert eng.row_groups == 3
    eng.form_log(True)
    got = m.sample(x0.to(DEV))
    forms = eng.form_log_read()
    eng.form_log(False)
    # the launches are those of 100-row shares, GroupNorm fused into the convs (csrc/gn_fused.h)
    for f in ("conv_gn16_kernel+gn_fused",):
        assert f in forms and 100 in forms[f], (f, forms.get(f))
    assert all(nb not in v for v in forms.values()), forms
    want = oracle_grouped_rows(cfg, PF, PI, hp, x0, rows)
    assert sorted(got) == sorted(want)
    worst = max(rel_rms(got[k][r].cpu(), want[k][j]) for k in want for j, r in enumerate(rows))
    print(f"OISST NB=300 on 3 row groups ({dtype}): worst rel-RMS over rows {rows} and 7 fields {worst:.3e}")
    assert worst <= OISST_TOL[dtype][1]


@pytest.mark.parametrize("nb,groups", [(96, 3), (50, 3), (70, 2)])
def test_grouped_rollout_draws_the_streams_of_the_ungrouped_call(nb, groups):
    """MC dropout (block, block1 and attention sites) and the engine's own normal draws on: a grouped call must reproduce the
    ungrouped one up to 16-bit rounding -- a row that drew another row's masks or noise would differ by O(1).  Two consecutive
    calls: the second must continue the streams (the counters a grouped call leaves behind are those of an ungrouped call).
    Uneven splits: 50 rows on 3 groups = 17 + 17 + 16, 70 on 2 = 35 + 35."""
    cfg, PF, PI, mirror = _oisst_pair(block_dropout=0.3, block_dropout1=0.2, attn_dropout=0.1)
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    hp = dict(OISST_HP, forward_conditioning="data+noise", enable_interpolator_dropout=True, additional_interpolation_steps=3)
    x0 = torch.randn(nb, 1, 60, 60, generator=torch.Generator().manual_seed(5)).to(DEV)
    outs = {}
    for g in (1, groups):
        m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=nb, row_groups=g, **hp)
        m.seed(99)
        m._ensure_engine((60, 60), nb)
        m.set_row_offset(1000)
        assert m._engine.row_groups == g
        a = m.sample_loop(x0)
        b = m.sample_loop(x0)
        outs[g] = (a, b)
        m._engine.close()
    for call in (0, 1):
        (xs1, f1, xn1), (xsg, fg, xng) = outs[1][call], outs[groups][call]
        worst = max(rel_rms(fg[k][r], f1[k][r]) for k in f1 for r in range(nb))
        print(f"NB={nb} on {groups} groups, call {call}: worst row rel-RMS vs the ungrouped call {worst:.3e}")
        assert worst <= 1e-2
        assert rel_rms(xsg, xs1) <= 1e-2 and rel_rms(xng, xn1) <= 1e-2  # sampler state gathered share by share
    # the second call drew fresh masks / noise
    assert rel_rms(outs[groups][1][1]["t7_preds"], outs[groups][0][1]["t7_preds"]) > 1e-2


def test_unet_simple_with_explicit_groups_matches_the_oracle_and_the_ungrouped_call():
    PF, PI = seeded_pair(64, 3, 2)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only", sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    nb = 40
    g = torch.Generator().manual_seed(77)
    x0, c = torch.randn(nb, 3, 23, 11, generator=g), torch.rand(nb, 2, 23, 11, generator=g)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=nb, row_groups=2)
    got = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    assert m._engine.row_groups == 2
    rows = [0, 19, 20, 39]
    want = oracle_rollout(PF, PI, mk, hp, x0[rows], c[rows])
    worst = max(rel_rms(got[k][r].cpu(), want[k][j]) for k in want for j, r in enumerate(rows))
    print(f"unet_simple NB=40 on 2 row groups: worst rel-RMS vs the oracle {worst:.3e}")
    assert worst <= 2.5e-2
    # MC dropout on: same streams as the ungrouped call
    hp2 = dict(hp, enable_interpolator_dropout=True)
    outs = []
    for groups in (1, 2):
        mm = build_dyffusion(PF, PI, mk, 3, 2, hp2, max_batch=nb, row_groups=groups)
        mm.seed(4242)
        outs.append(mm.sample(x0.to(DEV), static_condition=c.to(DEV)))
    worst = max(rel_rms(outs[1][k][r], outs[0][k][r]) for k in outs[0] for r in range(nb))
    print(f"unet_simple NB=40, MC dropout on: 2 groups vs ungrouped, worst row rel-RMS {worst:.3e}")
    assert worst <= 2.5e-2


def test_row_groups_must_be_set_before_the_weights_and_small_calls_run_ungrouped():
    from dyffusion_amd.engine import EngineError
    PF, PI = seeded_pair(64, 3, 2)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.0)
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only", sampling_type="cold",
              refine_intermediate_predictions=False, enable_interpolator_dropout=False)
    m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=64, row_groups=2)
    g = torch.Generator().manual_seed(1)
    x0, c = torch.randn(64, 3, 23, 11, generator=g).to(DEV), torch.rand(64, 2, 23, 11, generator=g).to(DEV)
    big = m.sample(x0, static_condition=c)
    eng = m._engine
    with pytest.raises(EngineError, match="before dyf_load_weights"):
        eng._check(eng._lib.dyf_set_row_groups(eng._h, 3))
    eng.form_log(True)
    small = m.sample(x0[:8], static_condition=c[:8])  # 8 rows < 2 x 16: runs on the engine itself
    forms = eng.form_log_read()
    eng.form_log(False)
    assert all(set(v) == {8} or set(v) <= {8, 16} for v in forms.values()), forms  # paired interpolator launches are 16 rows
    for k in big:
        assert rel_rms(small[k], big[k][:8]) <= 2.5e-2


def test_live_communicator_and_a_second_engine_keep_the_grouped_rollout_fast():
    """VERDICT r3 item 3 / DESIGN 4.5: three concurrent row groups + the caller's stream use all four hardware queues of the process;
    one more stream with work (RCCL's, torch.distributed's, a second engine's graph) makes two groups share a queue and the 300-row
    OISST rollout falls well below what TWO groups deliver.  An engine that owns a communicator therefore runs its calls on two
    groups (engine.hip sample_into_stack).  Here: the clean 3-group rate; then, with a live 1-rank RCCL communicator on the engine
    AND a second engine with a captured graph alive in the process, the same call must (a) launch 150-row shares and (b) stay
    within 15 % of the clean rate (measured: 3 700 vs 3 850 fields/s; the perturbed 3-group rate is ~3 100)."""
    import time

    from tests.gpu_common import build_dyffusion, seeded_pair
    cfg, PF, PI, mirror = _oisst_pair(block_dropout=0.3, attn_dropout=0.1)
    nb = 300
    F_, I_ = mirror(PF, cfg, 1, 1, 1), mirror(PI, cfg, 2, 0, 1)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=nb, forward_conditioning="data+noise", **OISST_HP)
    x0 = torch.randn(nb, 1, 60, 60, generator=torch.Generator().manual_seed(5)).to(DEV)

    def rate(fn, reps=3):
        fn()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        torch.cuda.synchronize()
        return nb * 7 * reps / (time.perf_counter() - t0)

    clean = rate(lambda: m.sample(x0))
    eng = m._engine
    assert eng.row_groups == 3
    # company: a second engine with a captured graph, kept alive, and a live communicator on the sampling engine
    PF2, PI2 = seeded_pair(64, 3, 2)
    mk = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only", sampling_type="cold",
              refine_intermediate_predictions=True, enable_interpolator_dropout=True)
    other = build_dyffusion(PF2, PI2, mk, 3, 2, hp, max_batch=8)
    xo, co = torch.randn(8, 3, 23, 11).to(DEV), torch.rand(8, 2, 23, 11).to(DEV)
    other.sample(xo, static_condition=co)
    other.sample(xo, static_condition=co)  # graph replay
    m.comm_init(HipEngine.comm_unique_id(eng.dtype), 0, 1, (60, 60), nb)
    assert eng.comm_count() == 1
    eng.form_log(True)
    m.sample(x0)
    forms = eng.form_log_read()
    eng.form_log(False)
    rows = set(forms["conv_gn16_kernel+gn_fused"])
    assert rows == {150}, rows  # two groups of 150 rows, not three of 100
    busy = rate(lambda: (m.sample(x0), other.sample(xo, static_condition=co))[0])
    gathered = rate(lambda: m.sample_gathered(x0, None, nb))
    print(f"OISST 300 rows: clean 3 groups {clean:.0f} fields/s; live communicator + second engine: {busy:.0f} (sample), "
          f"{gathered:.0f} (sample_gathered through the 1-rank all-gather)")
    assert busy >= 0.85 * clean and gathered >= 0.85 * clean
    other._engine.close()
    eng.close()
