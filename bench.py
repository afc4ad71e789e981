"""Headline benchmark: sampled fields/sec of the DYffusion h-step rollout (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nb NB] [--ensemble-total M]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `DYffusion.sample` call = one full h=16 rollout (16 forecaster + 44 interpolator forwards,
cold sampling, refine pass, MC dropout ON in the interpolator) over NB ensemble rows resident in HBM, executed by
libdyffusion_hip.so as a captured hipGraph.  Workload = BASELINE.json configs[1]: Navier-Stokes 221x42, C=3 (+2
static channels), unet_simple dim 64 @256^2, bf16 MFMA / fp32 accumulate.

N > 1 (one process per GPU, RCCL): rows are independent ensemble members, so the rollout itself has no exchange step;
every rank keeps the SAME seed and samples its block of global rows (the dropout streams are keyed by the global row, so
the fields do not depend on N), and at the end of EVERY step the forecast stack is all-gathered over RCCL in ONE collective
issued by the engine itself on the rollout's stream (`dyf_sample_gather`; `DYF_BENCH_EXCHANGE=torch`: one
all_gather_into_tensor through torch.distributed) -- inside the timed region.  Default = weak scaling: NB rows per rank.  `--ensemble-total M` = strong scaling: a FIXED M-row ensemble
(e.g. the reference's 50 members) split 7,7,6,... over the ranks.  value = total rows * h * K / max-over-ranks wall time.
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
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the number is
    REPLAYED from the newest committed profiles/*_pmc_traffic.json (rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE in separate
    passes, corrected as MI355X_MICROARCH.md prescribes; the file records the command that produced it), scaled by rows; None
    when none is on record.  Returns (bytes, description of the source)."""
    import glob

    try:
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(newest) as f:
            rec = json.load(f)
        cmd = rec.get("command", "command not recorded")
        return round(rec["hbm_bytes_per_row"] * nb), (f"replayed, not measured in this run: profiles/{os.path.basename(newest)} "
                                                      f"(separate rocprofv3 --pmc passes of `{cmd}`), scaled to {nb} rows")
    except Exception:
        return None, None


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


def _resnet_state(net, seed, conv_gain=1.0):
    sd = random_state(net, seed)
    for k in sd:
        if k.endswith(".norm.g"):
            sd[k] = torch.ones_like(sd[k])
        elif sd[k].dim() == 4 and conv_gain != 1.0:
            sd[k] = sd[k] * conv_gain
    return sd


def _resnet_roofline(eng, kind, nb, name, peak_tflops):
    # with row groups the rollout's launches cover one group's share of the rows: time those (an eager rollout of that many rows
    # on the engine itself; inside the concurrent run the same launch shares the chip with the other groups' kernels)
    groups = eng.row_groups
    rows = -(-nb // groups)
    ms, launches, fl, by = eng.time_kernel_in_rollout(kind, rows)
    r = {"kernel": name, "rows_per_launch": rows, "row_groups": groups, "avg_ms": round(ms, 4), "launches": launches,
         "algorithmic_bytes_per_launch": by,
         "hbm_gbps": round(by / ms / 1e6, 1), "hbm_frac": round(by / ms / 1e6 / 8000.0, 4)}
    if fl > 0:
        r.update({"bound": "mfma", "achieved": round(fl / ms / 1e9, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                  "frac": round(fl / ms / 1e9 / peak_tflops, 4), "flops_per_launch": fl})
    else:
        r.update({"bound": "hbm", "achieved": r["hbm_gbps"], "peak": 8000.0, "unit": "GB/s", "frac": r["hbm_frac"]})
    return r


def _time_rollouts(model, x0, reps):
    model.sample(x0)  # captures the graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model.sample(x0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert all(bool(torch.isfinite(v).all()) for v in out.values()), "non-finite forecast"
    return dt


def bench_oisst(dev, nb=300, reps=3):
    """BASELINE configs[2] shapes on ONE GPU: OISST 60x60x1, ResNet-UNet pair dim 64 mults (1,2,4), DYffusion h=7, k=25 (T=32: 32
    forecaster + 61 interpolator forwards), data+noise, MC dropout on, hipGraph; NB rows (50 members x 6 tiles)."""
    import dyffusion_amd as D

    kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
    F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
    I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2,
               attn_dropout=0.6, **kw)
    # conv gains halved (as for the 512^2 pair): the T=32 recursion of a random-init pair must stay inside fp16's range
    F.load_state_dict(_resnet_state(F, 0, 0.5))
    I.load_state_dict(_resnet_state(I, 1, 0.5))
    dtype = os.environ.get("DYF_BENCH_OISST_DTYPE", D.default_dtype_for(I))
    m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                    additional_interpolation_steps=25, refine_intermediate_predictions=False, max_batch=nb, dtype=dtype)
    m.seed(2)
    x0 = torch.randn(nb, 1, 60, 60, generator=torch.Generator().manual_seed(3)).to(dev)
    dt = _time_rollouts(m, x0, reps)
    eng = m._engine
    nf, ni = eng.forward_counts()
    fl = nf * eng.net_flops(0) + ni * eng.net_flops(1)
    res = {"workload": "BASELINE configs[2] shapes, 1 GPU: OISST 60x60x1, unet.Unet dim 64 mults (1,2,4), DYffusion h=7 k=25 "
                       "(T=32), data+noise, MC dropout on, hipGraph rollout", "dtype": dtype, "rows": nb, "row_groups": eng.row_groups,
           "net_forwards_per_rollout": nf + ni, "fields_per_s": round(nb * 7 / dt, 1), "ms_per_rollout": round(1e3 * dt, 2),
           "gflop_per_field": round(fl / 7 / 1e9, 2), "whole_rollout_tflops": round(nb * fl / dt / 1e12, 1),
           "roofline": _resnet_roofline(eng, 0, nb, "level-0 3x3 weight-standardised convs 64->64 @60x60", PEAK_BF16_TFLOPS),
           "roofline_groupnorm": _resnet_roofline(eng, 2, nb, "GroupNorm(8)+FiLM+SiLU+dropout(+residual) chain, 64 ch @60x60", PEAK_BF16_TFLOPS)}
    log(f"OISST NB={nb} ({dtype}): {res['ms_per_rollout']} ms per rollout -> {res['fields_per_s']} fields/s")
    del m
    return res


def bench_synth512(dev, nb=4, reps=1):
    """BASELINE configs[4] shapes on ONE GPU: synthetic 512x512x4ch, ResNet-UNet pair dim 64 mults (1,2,4) (bottleneck attention
    over 128^2 = 16 384 tokens), DYffusion h=32 (32 + 61 forwards), fp16, MC dropout on, hipGraph; NB rows."""
    import dyffusion_amd as D

    I = D.Unet(dim=64, dim_mults=(1, 2, 4), with_time_emb=True, num_input_channels=8, num_output_channels=4,
               block_dropout=0.1, attn_dropout=0.1)
    F = D.Unet(dim=64, dim_mults=(1, 2, 4), with_time_emb=True, num_input_channels=4, num_output_channels=4)
    # conv gains halved: the h=32 recursion of a random-init pair must stay inside fp16's range
    I.load_state_dict(_resnet_state(I, 1, 0.5))
    F.load_state_dict(_resnet_state(F, 0, 0.5))
    m = D.DYffusion(F, D.InterpolatorHandle(I, 32), timesteps=32, forward_conditioning="none", interpolate_before_t1=True,
                    refine_intermediate_predictions=False, enable_interpolator_dropout=True, max_batch=nb, dtype="fp16")
    m.seed(2)
    x0 = torch.randn(nb, 4, 512, 512, generator=torch.Generator().manual_seed(4)).to(dev)
    dt = _time_rollouts(m, x0, reps)
    eng = m._engine
    nf, ni = eng.forward_counts()
    fl = nf * eng.net_flops(0) + ni * eng.net_flops(1)
    res = {"workload": "BASELINE configs[4] shapes, 1 GPU: synthetic 512x512x4, unet.Unet dim 64 mults (1,2,4), DYffusion h=32, "
                       "fp16 MFMA conv/attention, MC dropout on, hipGraph rollout", "dtype": "fp16", "rows": nb,
           "net_forwards_per_rollout": nf + ni, "fields_per_s": round(nb * 32 / dt, 1), "ms_per_rollout": round(1e3 * dt, 2),
           "gflop_per_field": round(fl / 32 / 1e9, 2), "whole_rollout_tflops": round(nb * fl / dt / 1e12, 1),
           "roofline": _resnet_roofline(eng, 1, nb, "flash_attention2_kernel (16 384 tokens, 4 heads x 32; dropout on the "
                                                    "probabilities in the interpolator's launches)", PEAK_BF16_TFLOPS),
           "roofline_conv": _resnet_roofline(eng, 0, nb, "level-0 3x3 weight-standardised convs 64->64 @512x512", PEAK_BF16_TFLOPS)}
    log(f"512^2 NB={nb} (fp16): {res['ms_per_rollout']} ms per rollout -> {res['fields_per_s']} fields/s")
    del m
    return res


def cpu_model():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name"):
                    return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(F, I):
    """Reference CPU path timed on this host: the pure-PyTorch fp32 restatement (oracle/, parity-locked to the imported
    reference through tests/golden) runs the SAME workload, MC dropout on (RNG is a third of the reference's CPU time and is
    not skipped), at NB = 1 and NB = 4.  Bounded sample (~30 s): full rollouts, repeated while time allows.
    Threads: ATen's small-tensor ops and bernoulli_ stop scaling far below the core count of a 2-socket host, so one
    interpolator forward is timed at 32 / 64 (/ all, up to 96) threads and the fastest setting is used -- and reported."""
    from oracle import nets, sampler

    ncpu = os.cpu_count() or 1
    PF = {k: v.float() for k, v in F.state_dict().items()}
    PI = {k: v.float() for k, v in I.state_dict().items()}
    cfg = dict(DIFFUSION_KW, num_input_channels=C)
    drop = nets.DropoutFast()
    g = torch.Generator().manual_seed(1)
    x4, c4 = torch.randn(4, C, H, W, generator=g), torch.rand(4, CS, H, W, generator=g)

    def f_fn(x, t, cond):
        return nets.unet_simple_forward(PF, MODEL_KW, x, t, cond)

    def i_fn(x, t, cond):
        return nets.unet_simple_forward(PI, MODEL_KW, x, t, cond, dropout=drop)

    with torch.no_grad():
        best = None
        forced = os.environ.get("DYF_CPU_THREADS")
        # (all 256 hardware threads of the 2-socket host: 17 s for the forward that takes 0.13 s on 32 -- not tried beyond 96)
        for nt in ([int(forced)] if forced else sorted({min(32, ncpu), min(64, ncpu)} | ({ncpu} if ncpu <= 96 else set()))):
            torch.set_num_threads(nt)
            xi = torch.cat([x4[:1], x4[:1]], 1)
            i_fn(xi, torch.ones(1), c4[:1])  # warm-up (thread pool, mkldnn primitives)
            t0 = time.perf_counter()
            i_fn(xi, torch.ones(1), c4[:1])
            dt = time.perf_counter() - t0
            log(f"cpu baseline: one interpolator forward on {nt} threads {dt:.3f} s")
            if best is None or dt < best[1]:
                best = (nt, dt)
        torch.set_num_threads(best[0])
        res = {}
        for nb, budget, max_reps in ((1, 14.0, 3), (4, 10.0, 3)):
            reps, t0 = 0, time.perf_counter()
            while reps < 1 or (time.perf_counter() - t0 < budget and reps < max_reps):
                sampler.sample_loop(f_fn, i_fn, x4[:nb], c4[:nb], cfg)
                reps += 1
            dt = time.perf_counter() - t0
            res[nb] = (reps, dt, reps * nb * HORIZON / dt)
            log(f"cpu baseline NB={nb}: {reps} rollout(s) in {dt:.1f} s = {res[nb][2]:.3f} fields/s")
    top = max(res, key=lambda k: res[k][2])
    return {"value": round(res[top][2], 4), "unit": "fields/s", "cores": best[0], "kind": "port",
            "host_cores": ncpu, "cpu_model": cpu_model(),
            "fields_per_s_nb1": round(res[1][2], 4), "fields_per_s_nb4": round(res[4][2], 4),
            "sample": f"full h={HORIZON} rollouts (60 network forwards each), fp32, MC dropout on: NB=1 x{res[1][0]} in {res[1][1]:.1f} s, "
                      f"NB=4 x{res[4][0]} in {res[4][1]:.1f} s; {best[0]} of {ncpu} host threads (fastest of 32/64 on one forward)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--nb", type=int, default=int(os.environ.get("DYF_BENCH_NB", "80")),
                    help="rows per GPU; default 80 = the reference's NS evaluation batch: eval_batch_size 4 x num_predictions 20 "
                         "(experiment/navier_stokes.yaml:12-16)")
    ap.add_argument("--ensemble-total", type=int, default=int(os.environ.get("DYF_BENCH_ENSEMBLE", "0")),
                    help="strong scaling: a fixed ensemble of this many rows split over the ranks (0 = weak scaling, --nb rows per rank)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the OISST (configs[2]) and 512^2 (configs[4]) lines")
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

    from dyffusion_amd.distributed import init_engine_comm, rows_per_rank, sample_sharded, shard_rows

    strong = args.ensemble_total > 0
    total_rows = args.ensemble_total if strong else world * args.nb
    nb = rows_per_rank(total_rows, world)  # rows every rank launches (uneven shards repeat a row, distributed.py)
    log(f"building model, {total_rows} rows over {world} rank(s), {nb} per rank")
    model, F, I = build_model(nb, use_graph=not args.no_graph)
    log("model built")
    # every rank holds the full (total_rows, ...) inputs (111 KB per row) and the same seed; sample_sharded makes it
    # sample its own block of global rows and all-gathers the forecast stack
    g = torch.Generator().manual_seed(100)
    x0 = torch.randn(total_rows, C, H, W, generator=g).to(dev)
    static = torch.rand(total_rows, CS, H, W, generator=g).to(dev)
    gather = os.environ.get("DYF_BENCH_GATHER", "1") == "1"  # =0: time the rollouts without the exchange (A/B)
    lo, hi = shard_rows(total_rows, world, rank)

    def step():
        if world > 1 and gather:
            preds = sample_sharded(model, x0, static, exchange=exchange)
            assert preds[f"t{HORIZON}_preds"].shape[0] == total_rows
        else:
            model.set_row_offset(lo)
            preds = model.sample(x0[lo:hi], static_condition=static[lo:hi])
        return preds

    model.seed(2)
    model._ensure_engine((H, W), nb)
    log("engine created, weights uploaded")
    # N > 1 over RCCL: the ENGINE owns the communicator (dyf_comm_init) and issues the one all-gather of the forecast stack
    # itself, on the rollout's stream (dyf_sample_gather); DYF_BENCH_EXCHANGE=torch selects the torch.distributed route
    exchange = os.environ.get("DYF_BENCH_EXCHANGE", "engine" if world > 1 and dist.get_backend() == "nccl" else "torch")
    if world > 1 and gather and exchange == "engine":
        try:
            init_engine_comm(model, (H, W), total_rows)
            ok = 1
        except Exception as ex:  # e.g. no librccl for dlopen: every rank falls back to the torch.distributed route together
            log(f"engine-owned communicator unavailable ({type(ex).__name__}: {ex})")
            ok = 0
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 1:
            log("engine-owned RCCL communicator initialised")
        else:
            exchange = "torch"
            model._engine_comm_world = 1
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

    fields = total_rows * HORIZON * args.steps
    eng = model._engine
    n_f, n_i = eng.forward_counts()
    flops_rollout_row = n_f * eng.net_flops(0) + n_i * eng.net_flops(1)
    exec_rollout_row = n_f * eng.net_flops_executed(0) + n_i * eng.net_flops_executed(1)
    result = {
        "metric": "sampled fields/sec (h-step rollout)", "value": round(fields / dt, 3), "unit": "fields/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * dt / args.steps, 3),
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None, "dtype": "bf16",
        "data": "synthetic",
        "config": {"workload": "BASELINE configs[1]: Navier-Stokes 221x42, C=3+2 static ch, unet_simple dim 64 @256^2, "
                               "DYffusion h=16 cold sampling + refine, interpolator MC dropout p=0.15, hipGraph rollout",
                   "rows_per_gpu": nb, "total_rows": total_rows, "net_forwards_per_rollout": n_f + n_i,
                   "parallelism": f"ensemble-sharded dp{world}" + (f" + ONE all-gather of the forecast stack per step ({exchange}-owned exchange)" if gather and world > 1 else ""),
                   # gflop_per_field / whole_rollout_tflops: the REFERENCE's dense 2*MAC count (what the CPU path executes);
                   # executed_*: the contractions this engine runs (sparse last decoder block, stem composed into enc0,
                   # readout at the 4 neighbours the final resample reads)
                   "gflop_per_field": round(flops_rollout_row / HORIZON / 1e9, 2),
                   "whole_rollout_tflops": round(total_rows * flops_rollout_row * args.steps / dt / 1e12, 2),
                   "executed_gflop_per_field": round(exec_rollout_row / HORIZON / 1e9, 2),
                   "executed_whole_rollout_tflops": round(total_rows * exec_rollout_row * args.steps / dt / 1e12, 2)},
    }
    if rank == 0:
        # roofline of the dominant kernel: the last decoder block's 3x3 conv (256 -> 64 ch @256^2, 40 % of a forward),
        # HIP events on the launch stream, operands = live workspace activations
        # Dominant kernel: conv_halo_rows_kernel<0> (dense form; dec3 + dec4 = 27 % of a forward's time), largest launch dec4.
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

        result["roofline"] = layer_roofline(10, "conv_halo_rows_kernel<0> (dec4: fused x2-upsample + 3x3 conv, 256->128 ch, 64^2->128^2)")
        result["roofline"]["traffic"], result["roofline"]["traffic_source"] = pmc_traffic(nb)
        result["roofline_dec5_sparse"] = layer_roofline(
            11, "conv_halo_rows_mixed_kernel = conv_halo_rows_kernel<1, SH> (dec5: fused x2-upsample + 3x3 conv, 256->64 ch, 128^2->256^2, "
                "104 of 256 output columns as 3 x 16 + 4 list entries per phase)")
        if world == 1 and not args.no_extra_configs:
            # BASELINE configs[2] and configs[4] at their named shapes on this GPU (not the headline metric: extra keys)
            # the NS engine is CLOSED first (dyf_engine_destroy; `del` alone would not: the network modules hold it too): a live engine's
            # captured graph keeps a hardware queue, and the three concurrent row groups of the OISST rollout then share the 4
            # queues of the process with it (3 660 vs 2 870 fields/s measured)
            eng.close()
            del model, preds
            torch.cuda.empty_cache()
            for key, fn in (("config2_oisst", bench_oisst), ("config4_synth512", bench_synth512)):
                try:
                    result[key] = fn(dev)
                except Exception as ex:  # the headline line must survive a failure here
                    result[key] = {"error": f"{type(ex).__name__}: {ex}"}
                torch.cuda.empty_cache()
        if world == 1 and not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(F, I)
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
