"""-m gpu: bench.py's N > 1 branch, launched the way the driver launches it (python -m torch.distributed.run --nnodes=1
--nproc-per-node N ... bench.py --gpus N), rehearsed on ONE GPU: both ranks wrap onto device 0 (bench.py maps LOCAL_RANK modulo the
visible devices) and torch.distributed uses gloo, because RCCL refuses two ranks of one communicator on the same device ("Duplicate
GPU detected").  Through round 4 this whole branch -- `strong`, `config2_oisst`, `config4_synth512`, the exchange selection,
`nranks_seen` -- had never executed anywhere: the first SCALE run would have been its first run.

What is covered: every key of the N > 1 line, no `error` entries, the all-gathered forecast stacks complete and finite on rank 0
(bench.py asserts both), `value` consistent with a 1-rank line of the same total rows, and the engine-owned exchange REQUESTED
(DYF_BENCH_EXCHANGE=engine) on a box where the 2-rank RCCL communicator cannot exist: every rank must agree on the fallback to the
torch.distributed route instead of hanging.  What is NOT covered here: RCCL itself across devices (tests/test_gpu_distributed.py
runs dyf_sample_gather on a 1-rank communicator; the multi-device collective is the driver's SCALE run)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_bench(nproc, extra_args, extra_env, timeout=900, self_launch=False):
    env = dict(os.environ, DYF_DIST_BACKEND="gloo", MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0",
               DYF_BENCH_OISST_ROWS="160", DYF_BENCH_SYNTH_ROWS="2", **{"DYF_BENCH_STRONG_ROWS": "50,80", **extra_env})
    if self_launch:  # the driver's N = 1 command shape with --gpus N: no launcher, no WORLD_SIZE / RANK in the environment
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "DYF_DIST_BACKEND", "MASTER_ADDR"):
            env.pop(k, None)
        cmd = [sys.executable, "bench.py", "--gpus", str(nproc)]
    elif nproc > 1:
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
               "--master-port", str(_free_port()), "bench.py", "--gpus", str(nproc)]
    else:
        cmd = [sys.executable, "bench.py", "--gpus", "1"]
    r = subprocess.run(cmd + extra_args, cwd=ROOT, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, f"rc {r.returncode}\n--- stdout\n{r.stdout[-3000:]}\n--- stderr\n{r.stderr[-6000:]}"
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{") and '"metric"' in ln]
    assert len(lines) == 1, f"exactly one JSON line from rank 0, got {len(lines)}:\n{r.stdout[-2000:]}"
    # ... and NOTHING else on stdout (bench.claim_stdout: gloo's connection notes and RCCL's banner go to stderr)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[:2000]
    return json.loads(lines[0]), r.stderr


def _check_multirank_line(res, world, rows_per_gpu, strong_rows=(("ensemble_50", 50), ("ensemble_80", 80))):
    assert res["n_gpus"] == world and res["scaling"] == "weak" and res["metric"].startswith("sampled fields/sec")
    assert res["config"]["rows_per_gpu"] == rows_per_gpu and res["config"]["total_rows"] == world * rows_per_gpu
    assert res["value"] > 0 and res["ms_per_step"] > 0 and res["unit"] == "fields/s" and res["dtype"] == "bf16"
    assert "nranks_seen" in res
    for key in ("strong", "config2_oisst", "config4_synth512"):
        assert key in res, sorted(res)
        assert "error" not in res[key], (key, res[key])
    assert sum(res["rows_by_rank"]) == world * rows_per_gpu and len(res["rows_by_rank"]) == world
    assert res["ms_per_step_without_exchange"] > 0 and "exchange_ms_per_step" in res
    a = res["rank0_alone"]  # the N = 1 equivalent measured inside the run (weak scaling: one rank's rows)
    assert a["rows"] == rows_per_gpu and a["fields_per_s"] > 0 and a["ideal"] == world and a["speedup_of_this_line"] > 0
    for name, rows in strong_rows:
        s = res["strong"][name]
        assert "error" not in s, s
        assert s["total_rows"] == rows and s["rows_per_gpu"] == -(-rows // world) and s["scaling"] == "strong"
        assert s["fields_per_s"] > 0 and s["exchange"] in ("torch", "engine")
        assert sum(s["rows_by_rank"]) == rows and max(s["rows_by_rank"]) == s["rows_per_gpu"]
        assert s["rank0_alone_fields_per_s"] > 0 and s["speedup_vs_rank0_alone"] > 0 and s["ceiling"] == round(rows / s["rows_per_gpu"], 3)
        assert s["ms_per_step_without_exchange"] > 0
    o = res["config2_oisst"]
    assert o["rank0_alone_fields_per_s"] > 0 and sum(o["rows_by_rank"]) == 160
    assert o["total_rows"] == 160 and o["rows_per_gpu"] == 80 and o["fields_per_s"] > 0 and o["scaling"] == "strong"
    assert o["row_groups"] >= 2, o  # 80 rows of 60 x 60 per rank: the grouped rollout is what the SCALE run will launch at 150
    z = res["config4_synth512"]
    assert z["total_rows"] == 2 and z["rows_per_gpu"] == 1 and z["fields_per_s"] > 0
    # the N > 1 line carries no single-GPU-only sections
    for key in ("roofline", "cpu_baseline", "batch_curve"):
        assert key not in res


def test_bench_two_ranks_on_one_gpu_torch_exchange_and_consistency_with_one_rank():
    # (50 members; 200 rows = the reference's NS test ensemble, 4 x 50: 100 rows per rank here)
    two, _ = _run_bench(2, ["--steps", "2", "--warmup", "1", "--nb", "8"], {"DYF_BENCH_STRONG_ROWS": "50,200"})
    _check_multirank_line(two, 2, 8, strong_rows=(("ensemble_50", 50), ("ensemble_200", 200)))
    assert two["strong"]["ensemble_200"]["rows_by_rank"] == [100, 100]
    assert "all-gather" in two["config"]["parallelism"] and "torch-owned" in two["config"]["parallelism"]
    assert two["nranks_seen"] == 0  # the torch.distributed route ran (gloo): no engine communicator
    assert all(two["strong"][k]["exchange"] == "torch" for k in two["strong"])
    one, _ = _run_bench(1, ["--steps", "2", "--warmup", "1", "--nb", "16", "--no-extra-configs", "--no-cpu-baseline"], {})
    assert one["n_gpus"] == 1 and one["config"]["total_rows"] == 16
    ratio = two["value"] / one["value"]
    print(f"2 ranks x 8 rows on one GPU: {two['value']:.0f} fields/s; 1 rank x 16 rows: {one['value']:.0f} fields/s; ratio {ratio:.2f}")
    # same GPU, same total rows: two time-sliced 8-row rollouts + a host-staged gloo exchange against one 16-row rollout
    assert 0.25 <= ratio <= 1.3, ratio


def test_bench_self_launches_its_ranks_when_no_launcher_started_it():
    """`python bench.py --gpus 2 --steps 2 --warmup 1` -- the driver's N = 1 command with another N, nothing in the environment:
    bench.py re-executes itself under torch.distributed.run (it used to SystemExit), notices that one visible GPU
    cannot carry two RCCL ranks and points torch.distributed at gloo, and rank 0 prints the one line.  (Headline section only:
    --no-extra-configs; the `strong` / OISST / 512^2 sections of the N > 1 line are covered by the torchrun-launched tests.)"""
    res, err = _run_bench(2, ["--steps", "2", "--warmup", "1", "--nb", "8", "--no-extra-configs"], {}, self_launch=True, timeout=900)
    assert "self-launch:" in err and "torch.distributed.run" in err
    assert res["n_gpus"] == 2 and res["scaling"] == "weak" and res["config"]["total_rows"] == 16 and res["value"] > 0
    assert res["nranks_seen"] == 0 and res["devices_visible"] >= 1 and res["rows_by_rank"] == [8, 8]
    assert res["ms_per_step_without_exchange"] > 0 and "exchange_ms_per_step" in res
    assert res["rank0_alone"]["rows"] == 8 and res["rank0_alone"]["ideal"] == 2 and res["rank0_alone"]["fields_per_s"] > 0
    assert "strong" not in res and "roofline" not in res


def test_bench_two_ranks_engine_exchange_requested_falls_back_on_every_rank_or_runs():
    res, err = _run_bench(2, ["--steps", "1", "--warmup", "1", "--nb", "4"], {"DYF_BENCH_EXCHANGE": "engine"})
    _check_multirank_line(res, 2, 4)
    exch = {res["strong"][k]["exchange"] for k in res["strong"]} | {res["config2_oisst"]["exchange"], res["config4_synth512"]["exchange"]}
    print("exchange with DYF_BENCH_EXCHANGE=engine and two ranks on one device:", exch, "nranks_seen", res["nranks_seen"])
    if exch == {"torch"}:  # RCCL refused two ranks on one device: all ranks agreed on the fallback (no hang, no error key)
        assert res["nranks_seen"] == 0
        assert "engine-owned communicator unavailable" in err
    else:                  # RCCL accepted the duplicate device: then the engine-owned exchange must have seen both ranks
        assert exch == {"engine"} and res["nranks_seen"] == 2
