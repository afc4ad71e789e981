"""The disk cache of full-size oracle outputs (tests/golden/oracle_cache, tests/gpu_common.cached) against the oracle itself.  The GPU
suite reads these instead of spending minutes of fp32 CPU rollouts per test, so a stale entry would silently relax a parity test; the
CPU suite owns the check:
  * every entry carries the fingerprint of its inputs / weights, the hash of its configuration and the hash of the oracle's sources
    (oracle/*.py, tests/rng_host.py) -- and all three match what the test modules compute TODAY (an oracle edit without a cache
    refresh fails here, not on the GPU box);
  * ONE ROW OF EVERY ENTRY is recomputed here from scratch and must equal the cached tensor;
  * a changed input, a changed hyper-parameter and a changed oracle source each invalidate an entry."""
import os

import numpy as np
import pytest
import torch

from tests import gpu_common

CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cache")
EXPECTED = ["ns80_dropout_seed20260929_row79", "ns80_rows_0_79", "oisst300_fwd_eval", "oisst300_grouped_rows", "oisst300_rollout_rows"]


@pytest.fixture(autouse=True)
def _cache_env(monkeypatch):
    monkeypatch.setenv("DYF_ORACLE_CACHE", "1")
    monkeypatch.delenv("DYF_WRITE_ORACLE_CACHE", raising=False)
    gpu_common._ORACLE_CACHE.clear()
    torch.set_num_threads(max(1, min(16, os.cpu_count() or 1)))
    yield
    gpu_common._ORACLE_CACHE.clear()


def _same(a, b):
    """cached vs fresh oracle output: the same fp32 computation run in another batch composition (a row alone instead of inside a
    2- or 61-row batch: ATen picks other blockings) -- equal to fp32 round-off accumulated over up to 93 forwards, nothing more."""
    from tests.helpers import rel_rms
    err, mx = rel_rms(a, b), float((a - b).abs().max())
    assert err <= 5e-5 and mx <= 1e-4 * (1.0 + float(b.abs().max())), (err, mx)


def _from_disk(key, call):
    """Run the test module's own oracle function with an oracle that must NOT be needed: the entry has to come from disk."""
    before = dict(gpu_common._ORACLE_CACHE)
    with np.load(os.path.join(CACHE, key + ".npz"), allow_pickle=False) as z:
        stored = {k: z[k].copy() for k in z.files}
    val = call()
    assert key in gpu_common._ORACLE_CACHE and key not in before
    if torch.is_tensor(val):
        assert np.array_equal(val.numpy(), stored["value"]), f"{key}: not served from disk (fingerprint / config / code hash mismatch)"
    else:
        for k, v in val.items():
            assert np.array_equal(v.numpy(), stored["v::" + k]), f"{key}/{k}: not served from disk (fingerprint / config / code hash mismatch)"
    return val


def test_cache_entries_exist_and_carry_all_three_identities():
    have = sorted(f[:-4] for f in os.listdir(CACHE) if f.endswith(".npz"))
    assert have == EXPECTED, have
    code = gpu_common.oracle_code_hash()
    for k in have:
        with np.load(os.path.join(CACHE, k + ".npz"), allow_pickle=False) as z:
            assert z["__fingerprint__"].ndim == 1 and z["__fingerprint__"].size % 3 == 0
            assert str(z["__kind__"]) in ("tensor", "dict")
            assert str(z["__code__"]) == code, f"{k}: oracle/*.py or tests/rng_host.py changed since the cache was written -- run tests/golden/make_oracle_cache.py"
            assert len(str(z["__config__"])) == 64
            for name in z.files:
                if name.startswith("v::") or name == "value":
                    assert np.isfinite(z[name]).all(), (k, name)


def test_ns80_rows_entry_equals_a_fresh_oracle_rollout_of_one_row():
    from tests import test_gpu_bench_forms as BF
    from tests.gpu_common import oracle_rollout
    mk, PF, PI, x0, c = BF._setup()
    cached = _from_disk("ns80_rows_0_79", lambda: BF.oracle_nb80_rows(mk, PF, PI, x0, c, [0, 79]))
    fresh = oracle_rollout(PF, PI, mk, BF.HP, x0[79:80], c[79:80])
    for k in ("t1_preds", "t8_preds", "t16_preds"):
        _same(cached[k][1:2], fresh[k])


def test_ns80_dropout_entry_equals_a_fresh_oracle_rollout_on_host_rebuilt_masks():
    from tests import test_gpu_bench_forms as BF
    mk, PF, PI, x0, c = BF._setup()
    hp = dict(BF.HP, enable_interpolator_dropout=True)
    cached = _from_disk("ns80_dropout_seed20260929_row79",
                        lambda: BF.oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, 20260929, 79))
    gpu_common._ORACLE_CACHE.clear()
    os.environ["DYF_ORACLE_CACHE"] = "0"  # (restored by the fixture's monkeypatch)
    fresh = BF.oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, 20260929, 79)
    for k in ("t1_preds", "t8_preds", "t16_preds"):
        _same(cached[k], fresh[k])


def test_oisst_forward_entry_equals_a_fresh_oracle_run_and_reacts_to_its_inputs():
    from oracle import nets
    from tests import test_gpu_bench_forms as BF
    cfg, PI, x, t, rows = BF.oisst_fwd_case()
    cached = _from_disk("oisst300_fwd_eval", lambda: BF.oracle_oisst_fwd_eval(cfg, PI, x, t, rows))
    sub = rows[:6]                                               # six of the 61 rows, recomputed now
    with torch.no_grad():
        fresh = nets.resnet_unet_forward(PI, cfg, x[sub], t[sub], None)
    assert cached.shape[0] == len(rows)
    _same(cached[:6], fresh)
    # a changed input invalidates the entry: the fingerprint no longer matches and the oracle runs again (a different result)
    gpu_common._ORACLE_CACHE.clear()
    x2 = x.clone()
    x2[rows[0]] += 1.0
    other = BF.oracle_oisst_fwd_eval(cfg, PI, x2, t, rows[:3])  # 3 rows: cheap
    assert other.shape[0] == 3 and not torch.allclose(other[0], cached[0])
    # ... and so does a changed weight that is NOT one of the two tensors the round-4 fingerprint looked at
    gpu_common._ORACLE_CACHE.clear()
    P2 = dict(PI)
    mid = sorted(k for k in P2 if k.startswith("mid_block1") and k.endswith("weight"))[0]
    P2[mid] = P2[mid] * 1.5
    other = BF.oracle_oisst_fwd_eval(cfg, P2, x, t, rows[:3])
    assert other.shape[0] == 3 and not torch.allclose(other[0], cached[0])
    # ... and a changed configuration
    gpu_common._ORACLE_CACHE.clear()
    other = BF.oracle_oisst_fwd_eval(dict(cfg, resnet_block_groups=4), PI, x, t, rows[:3])
    assert other.shape[0] == 3 and not torch.allclose(other[0], cached[0])


def test_oisst_rollout_entry_equals_a_fresh_oracle_rollout_of_one_row():
    from oracle import nets, sampler
    from tests import test_gpu_bench_forms as BF
    cfg, PF, PI, x0, noise, rows = BF.oisst_rollout_case()
    cached = _from_disk("oisst300_rollout_rows", lambda: BF.oracle_oisst_rollout_rows(cfg, PF, PI, x0, noise, rows))
    r = rows[-1:]
    it = iter(range(32))
    with torch.no_grad():
        fresh = sampler.sample_loop(lambda x, t, cnd: nets.resnet_unet_forward(PF, cfg, x, t, cnd),
                                    lambda x, t, cnd: nets.resnet_unet_forward(PI, cfg, x, t, cnd), x0[r], None, BF.OISST_ROLLOUT_HP,
                                    noise_fn=lambda tensor: noise[next(it)][r])
    for k in fresh:
        _same(cached[k][-1:], fresh[k])


def test_oisst_grouped_entry_equals_a_fresh_oracle_rollout_of_one_row():
    from oracle import nets, sampler
    from tests import test_gpu_row_groups as RG
    cfg, PF, PI, _, hp, x0, rows = RG.grouped_case()
    cached = _from_disk("oisst300_grouped_rows", lambda: RG.oracle_grouped_rows(cfg, PF, PI, hp, x0, rows))
    r = rows[1:2]
    with torch.no_grad():
        fresh = sampler.sample_loop(lambda x, t, cnd: nets.resnet_unet_forward(PF, cfg, x, t, cnd),
                                    lambda x, t, cnd: nets.resnet_unet_forward(PI, cfg, x, t, cnd), x0[r], None, hp)
    for k in fresh:
        _same(cached[k][1:2], fresh[k])


def test_an_oracle_source_edit_invalidates_every_entry(monkeypatch):
    from tests import test_gpu_bench_forms as BF
    real = gpu_common.oracle_code_hash()
    monkeypatch.setattr(gpu_common, "oracle_code_hash", lambda: "0" * 64)
    cfg, PI, x, t, rows = BF.oisst_fwd_case()
    calls = []

    def spy_forward(*a, **k):
        calls.append(1)
        return torch.zeros(len(rows), 1, 60, 60)

    monkeypatch.setattr(BF.nets, "resnet_unet_forward", spy_forward)
    out = BF.oracle_oisst_fwd_eval(cfg, PI, x, t, rows)
    assert calls and float(out.abs().sum()) == 0.0, "a cache entry written by other oracle sources must be ignored"
    assert real != "0" * 64
