"""Headline benchmark: sampled fields/sec of the DYffusion h-step rollout (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nb NB]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `DYffusion.sample` call = one full h=16 rollout (16 forecaster + 44 interpolator forwards,
cold sampling, refine pass, MC dropout ON in the interpolator) over NB ensemble rows resident in HBM, executed by
libdyffusion_hip.so as a captured hipGraph.  Workload = BASELINE.json configs[1]: Navier-Stokes 221x42, C=3 (+2
static channels), unet_simple dim 64 @256^2, bf16 MFMA / fp32 accumulate.  With N>1 every rank owns NB rows (weak
scaling, rows are independent ensemble members) and the forecast stack is all-gathered over RCCL at the end of every
step.  value = N * NB * h * K / max-over-ranks wall time.
"""
import argparse
import json
import math
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

H, W, C, CS, HORIZON, DIM = 221, 42, 3, 2, 16, 64
MODEL_KW = dict(dim=DIM, with_time_emb=True, outer_sample_mode="bilinear", upsample_dims=[256, 256], dropout=0.15,
                input_dropout=0.0)
DIFFUSION_KW = dict(timesteps=HORIZON, forward_conditioning="none", interpolate_before_t1=True,
                    schedule="before_t1_only", additional_interpolation_steps=0, sampling_type="cold",
                    time_encoding="dynamics", refine_intermediate_predictions=True, enable_interpolator_dropout=True)
PEAK_BF16_TFLOPS = 2500.0  # dense MFMA bf16 peak, /opt/skills/guides/MI355X_MICROARCH.md


def random_state(net, seed):
    """Random-init weights of the named architecture with O(1) activations (no checkpoints offline)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, v in net.state_dict().items():
        shp = tuple(v.shape)
        if k.endswith("num_batches_tracked"):
            sd[k] = torch.zeros((), dtype=torch.int64)
        elif k.endswith("running_var"):
            sd[k] = 0.5 + torch.rand(shp, generator=g)
        elif k.endswith("running_mean"):
            sd[k] = 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            sd[k] = (1.0 if k.endswith("weight") else 0.0) + 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = shp[0] * 4 if k.startswith("readout") else math.prod(shp[1:])
            sd[k] = torch.randn(shp, generator=g) * ((0.7 if "time_mlp" in k else 1.4) / math.sqrt(fan_in))
    return sd


def pmc_traffic(nb):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 --pmc passes (profiles/r01_pmc_traffic.json:
    FETCH_SIZE x2 + WRITE_SIZE, separate passes), scaled by rows; None when no PMC run is on record."""
    try:
        with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
            return round(json.load(f)["hbm_bytes_per_row"] * nb)
    except Exception:
        return None


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_model(nb, use_graph=True):
    import dyffusion_amd as D

    F = D.UNet(num_input_channels=C, num_output_channels=C, num_conditional_channels=CS, spatial_shape=(H, W), **MODEL_KW)
    I = D.UNet(num_input_channels=2 * C, num_output_channels=C, num_conditional_channels=CS, spatial_shape=(H, W), **MODEL_KW)
    F.load_state_dict(random_state(F, 0))
    I.load_state_dict(random_state(I, 1))
    m = D.DYffusion(F, D.InterpolatorHandle(I, HORIZON), max_batch=nb, use_graph=use_graph, **DIFFUSION_KW)
    return m, F, I


def cpu_baseline(F, I):
    """Reference CPU path timed on this host: the pure-PyTorch fp32 restatement (oracle/, parity-locked to the imported
    reference through tests/golden) runs the SAME workload at NB=1, MC dropout on.  Bounded sample: one rollout."""
    from oracle import nets, sampler

    # small-tensor ATen ops stop scaling (and bernoulli_/mkldnn oversubscribe) far below 256 threads: cap, and report it
    torch.set_num_threads(min(os.cpu_count() or 1, int(os.environ.get("DYF_CPU_THREADS", "32"))))
    PF = {k: v.float() for k, v in F.state_dict().items()}
    PI = {k: v.float() for k, v in I.state_dict().items()}
    cfg = dict(DIFFUSION_KW, num_input_channels=C)
    drop = nets.DropoutFast()
    g = torch.Generator().manual_seed(1)
    x0, c = torch.randn(1, C, H, W, generator=g), torch.rand(1, CS, H, W, generator=g)

    def f_fn(x, t, cond):
        return nets.unet_simple_forward(PF, MODEL_KW, x, t, cond)

    def i_fn(x, t, cond):
        return nets.unet_simple_forward(PI, MODEL_KW, x, t, cond, dropout=drop)

    with torch.no_grad():
        tw = time.perf_counter()
        f_fn(x0, torch.ones(1), c)  # warm-up (thread pool, mkldnn primitives)
        log(f"cpu baseline warm-up forward {time.perf_counter() - tw:.2f} s on {torch.get_num_threads()} threads")
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or (time.perf_counter() - t0 < 8.0 and reps < 3):
            sampler.sample_loop(f_fn, i_fn, x0, c, cfg)
            reps += 1
        dt = time.perf_counter() - t0
    return {"value": round(reps * HORIZON / dt, 4), "unit": "fields/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} full h={HORIZON} rollout(s) at NB=1 (60 network forwards each), fp32, MC dropout on, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nb", type=int, default=int(os.environ.get("DYF_BENCH_NB", "80")),
                    help="rows per GPU; default 80 = the reference's NS evaluation batch: eval_batch_size 4 x num_predictions 20 "
                         "(experiment/navier_stokes.yaml:12-16)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run for N>1")
    ndev = torch.cuda.device_count()
    dev_index = local_rank % max(1, ndev)  # one rank per GPU under the driver; ranks wrap only in single-GPU smoke runs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = os.environ.get("DYF_DIST_BACKEND", "nccl")  # "nccl" IS RCCL on ROCm; gloo only for plumbing checks
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    nb = args.nb
    log(f"building model, nb={nb}")
    model, F, I = build_model(nb, use_graph=not args.no_graph)
    log("model built")
    g = torch.Generator().manual_seed(100 + rank)
    x0 = torch.randn(nb, C, H, W, generator=g).to(dev)
    static = torch.rand(nb, CS, H, W, generator=g).to(dev)
    from dyffusion_amd.distributed import all_gather_rows

    gather = os.environ.get("DYF_BENCH_GATHER", "0") == "1"

    def step():
        _, preds, _ = model.sample_loop(x0, static_condition=static)
        if world > 1:
            # Rows (ensemble members x batch) are independent: the rollout itself has NO exchange step, so the data path
            # runs without a collective.  What the reference's DDP evaluation does exchange is per-rank metric scalars
            # (torchmetrics sync, _base_experiment.py:560-650): one small all-reduce per predict call stands in for it.
            # DYF_BENCH_GATHER=1 additionally all-gathers the full forecast stack onto every rank (sample_sharded).
            partial = torch.stack([preds[f"t{i}_preds"].float().abs().mean() for i in range(1, HORIZON + 1)])
            dist.all_reduce(partial, op=dist.ReduceOp.SUM)
            if gather:
                stack = torch.stack([preds[f"t{i}_preds"] for i in range(1, HORIZON + 1)], 0)  # (h, nb, C, H, W)
                full = all_gather_rows(stack, world * nb, row_dim=1)
                assert full.shape[1] == world * nb
        return preds

    model._ensure_engine((H, W), nb).seed(2 + rank)
    log("engine created, weights uploaded")
    for _ in range(args.warmup):
        step()
        torch.cuda.synchronize()
        log("warm-up step done")

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        preds = step()
    fence()
    dt = time.perf_counter() - t0
    log(f"timed region done: {dt:.3f} s for {args.steps} steps")
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    assert all(torch.isfinite(v).all() for v in preds.values()), "non-finite forecast"

    fields = world * nb * HORIZON * args.steps
    eng = model._engine
    n_f, n_i = eng.forward_counts()
    flops_rollout_row = n_f * eng.net_flops(0) + n_i * eng.net_flops(1)
    result = {
        "metric": "sampled fields/sec (h-step rollout)", "value": round(fields / dt, 3), "unit": "fields/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Navier-Stokes 221x42, C=3+2 static ch, unet_simple dim 64 @256^2, "
                               "DYffusion h=16 cold sampling + refine, interpolator MC dropout p=0.15, hipGraph rollout",
                   "rows_per_gpu": nb, "net_forwards_per_rollout": n_f + n_i, "parallelism": f"ensemble-sharded dp{world}" + (" + all-gather of the forecast stack" if gather and world > 1 else ""),
                   "gflop_per_field": round(flops_rollout_row / HORIZON / 1e9, 2),
                   "whole_rollout_tflops": round(world * nb * flops_rollout_row * args.steps / dt / 1e12, 2)},
    }
    if rank == 0:
        # roofline of the dominant kernel: the last decoder block's 3x3 conv (256 -> 64 ch @256^2, 40 % of a forward),
        # conv_up_halo_kernel; HIP events on the launch stream, operands = live workspace activations
        # Dominant kernel: conv_up_halo_kernel<0> (dense form; dec3 + dec4 = 27 % of a forward's time), largest launch dec4.
        # achieved = algorithmic FLOPs of one launch / average duration of that launch INSIDE the rollout: HIP events around
        # every dec4 conv (60 launches: 16 forecaster + 44 interpolator forwards, MC dropout on) of one eagerly launched
        # rollout on the launch stream (dyf_time_layer_in_rollout); the isolated back-to-back figure is kept beside it.
        # The sparse-column instance of the same kernel (dec5, only the output columns the readout reads) is reported too.
        def layer_roofline(layer, name):
            _, fl, by = eng.time_conv_layer(1, layer, nb, iters=1)
            ms, launches = eng.time_layer_in_rollout(layer, nb)
            ms_iso, _, _ = eng.time_conv_layer(1, layer, nb, iters=10)
            log(f"{name}: {ms:.3f} ms per launch in the rollout ({launches} launches), {ms_iso:.3f} ms isolated back-to-back")
            return {"bound": "mfma", "kernel": name, "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4), "avg_ms": round(ms, 4),
                    "launches": launches, "avg_ms_isolated": round(ms_iso, 4), "flops_per_launch": fl,
                    "algorithmic_bytes_per_launch": by}

        result["roofline"] = layer_roofline(10, "conv_up_halo_kernel<0> (dec4: fused x2-upsample + 3x3 conv, 512->128 ch, 64^2->128^2)")
        result["roofline"]["traffic"] = pmc_traffic(nb)
        result["roofline_dec5_sparse"] = layer_roofline(
            11, "conv_up_halo_kernel<1> (dec5: fused x2-upsample + 3x3 conv, 256->64 ch, 128^2->256^2, 104 of 256 output columns)")
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(F, I)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
