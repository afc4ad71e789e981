"""The disk cache of full-size oracle outputs (tests/golden/oracle_cache, tests/gpu_common.cached) against the oracle itself: one entry
-- the 61 rows of the OISST 300-row forward -- is recomputed here on the CPU and must equal the cached tensor; every entry must
carry a fingerprint of the right length for its case.  (The GPU suite reads these instead of spending minutes of fp32 CPU
rollouts per test; a stale entry would silently relax a parity test, so the CPU suite owns this check.)"""
import os

import numpy as np
import pytest
import torch

CACHE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cache")
EXPECTED = ["ns80_dropout_seed20260929_row79", "ns80_rows_0_79", "oisst300_fwd_eval", "oisst300_grouped_rows", "oisst300_rollout_rows"]


def test_cache_entries_exist_and_carry_fingerprints():
    have = sorted(f[:-4] for f in os.listdir(CACHE) if f.endswith(".npz"))
    assert have == EXPECTED, have
    for k in have:
        with np.load(os.path.join(CACHE, k + ".npz"), allow_pickle=False) as z:
            assert z["__fingerprint__"].ndim == 1 and z["__fingerprint__"].size % 3 == 0
            assert str(z["__kind__"]) in ("tensor", "dict")
            for name in z.files:
                if name.startswith("v::") or name == "value":
                    assert np.isfinite(z[name]).all(), (k, name)


def test_cached_oisst_forward_equals_a_fresh_oracle_run(monkeypatch):
    from tests import gpu_common
    from tests import test_gpu_bench_forms as BF
    cfg, PI, x, t, rows = BF.oisst_fwd_case()
    monkeypatch.setenv("DYF_ORACLE_CACHE", "1")
    monkeypatch.delenv("DYF_WRITE_ORACLE_CACHE", raising=False)
    gpu_common._ORACLE_CACHE.pop("oisst300_fwd_eval", None)
    cached = BF.oracle_oisst_fwd_eval(cfg, PI, x, t, rows)       # from disk (the fingerprint matches, or this recomputes)
    sub = rows[:6]                                               # six of the 61 rows, recomputed now
    from oracle import nets
    with torch.no_grad():
        fresh = nets.resnet_unet_forward(PI, cfg, x[sub], t[sub], None)
    assert cached.shape[0] == len(rows)
    assert torch.allclose(cached[:6], fresh, rtol=1e-5, atol=1e-6), float((cached[:6] - fresh).abs().max())
    # a changed input invalidates the entry: the fingerprint no longer matches and the oracle runs again (a different result)
    gpu_common._ORACLE_CACHE.pop("oisst300_fwd_eval", None)
    x2 = x.clone()
    x2[rows[0]] += 1.0
    other = BF.oracle_oisst_fwd_eval(cfg, PI, x2[:, :, :, :], t, rows[:2] + rows[2:3])  # 3 rows: cheap
    gpu_common._ORACLE_CACHE.pop("oisst300_fwd_eval", None)
    assert other.shape[0] == 3 and not torch.allclose(other[0], cached[0])
