"""Write tests/golden/oracle_cache/*.npz: OUTPUTS of the CPU oracle (oracle/*.py) for the full-size cases of the GPU suite, so that the
GPU box does not spend minutes of fp32 CPU rollouts per test (tests/gpu_common.cached: a cache entry is used only while the
fingerprint of the inputs and weights it was computed from matches; DYF_ORACLE_CACHE=0 ignores the cache).  The cases and their
oracle calls are the test modules' own functions -- this script only runs them with writing enabled.

Run in the build container:  python tests/golden/make_oracle_cache.py        (about 4 minutes of CPU time)
These are oracle outputs, not reference outputs: the oracle is pinned to the reference by the CPU suite (tests/test_oracle_*.py),
and tests/test_oracle_cache.py recomputes a row of every entry there.  The script refuses to write while the CPU
oracle suite (tests/test_oracle_*.py, tests/test_rng_host.py) is red."""
import os
import sys
import time

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, ROOT)
os.environ["DYF_WRITE_ORACLE_CACHE"] = "1"
os.environ["DYF_ORACLE_CACHE"] = "0"  # recompute everything

import subprocess  # noqa: E402

if os.environ.get("DYF_CACHE_SKIP_SUITE") != "1":
    # an oracle that no longer reproduces the reference's golden vectors must not be frozen into fixtures: the CPU oracle suite
    # (oracle vs tests/golden/*.npz|json, generated from the imported reference) has to be green first
    suite = ["tests/test_oracle_schedule.py", "tests/test_oracle_nets.py", "tests/test_oracle_sampler.py", "tests/test_oracle_losses.py",
             "tests/test_oracle_metrics.py", "tests/test_oracle_boundary.py", "tests/test_rng_host.py"]
    rc = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-m", "not gpu"] + suite, cwd=ROOT).returncode
    if rc != 0:
        raise SystemExit("the CPU oracle suite is red: refusing to write tests/golden/oracle_cache (fix the oracle first)")

import torch  # noqa: E402

torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
from tests import test_gpu_bench_forms as BF  # noqa: E402
from tests import test_gpu_row_groups as RG  # noqa: E402


def timed(name, fn):
    t0 = time.perf_counter()
    out = fn()
    print(f"{name}: {time.perf_counter() - t0:.1f} s", flush=True)
    return out


mk, PF, PI, x0, c = BF._setup()
timed("ns80 rows (dropout off)", lambda: BF.oracle_nb80_rows(mk, PF, PI, x0, c, [0, 79]))
hp = dict(BF.HP, enable_interpolator_dropout=True)
timed("ns80 row 79 (engine masks)", lambda: BF.oracle_nb80_row_with_engine_masks(mk, PF, PI, x0, c, hp, 20260929, 79))
cfg, PIo, x, t, rows = BF.oisst_fwd_case()
timed("oisst 300 forward, eval", lambda: BF.oracle_oisst_fwd_eval(cfg, PIo, x, t, rows))
cfg, PFo, PIo, xo, noise, rows = BF.oisst_rollout_case()
timed("oisst 300 rollout rows", lambda: BF.oracle_oisst_rollout_rows(cfg, PFo, PIo, xo, noise, rows))
cfg, PFg, PIg, _, hpg, xg, rows = RG.grouped_case()
timed("oisst 300 grouped rows", lambda: RG.oracle_grouped_rows(cfg, PFg, PIg, hpg, xg, rows))
d = os.path.join(ROOT, "tests", "golden", "oracle_cache")
print({f: os.path.getsize(os.path.join(d, f)) for f in sorted(os.listdir(d))})
