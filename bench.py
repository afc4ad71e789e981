"""Headline benchmark: sampled fields/sec of the DYffusion h-step rollout (BASELINE.json metric).

  python bench.py [--gpus N] [--steps K] [--warmup W] [--nb NB] [--ensemble-total M]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one `DYffusion.sample` call = one full h=16 rollout (16 forecaster + 44 interpolator forwards,
cold sampling, refine pass, MC dropout ON in the interpolator) over NB ensemble rows resident in HBM, executed by
libdyffusion_hip.so as a captured hipGraph.  Workload = BASELINE.json configs[1]: Navier-Stokes 221x42, C=3 (+2
static channels), unet_simple dim 64 @256^2, bf16 MFMA / fp32 accumulate.

Beside the headline line (N = 1) the JSON carries: `batch_curve` + `strong_scaling_projection` (what sharding a FIXED ensemble over 8
GPUs would give, from this GPU's own small-batch rates), `config1_ns_c2` (the 2-channel NS variant), `config3_ns_ar64` (BASELINE
configs[3]), `config2_oisst` (+ its batch curve), `config4_synth512`, `train_step` (both backbones) and `cpu_baseline`.
For N > 1: the NS line (weak by default) + a `strong` object (50- and 80-row ensembles split over the ranks) + `config2_oisst` /
`config4_synth512` on 300 / 8 rows sharded N ways, every one through the engine-owned all-gather, with `nranks_seen` = ncclCommCount.

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


def pmc_traffic(nb, forms):
    """HBM bytes per launch of the dominant kernel.  PMC counters cannot be read from inside this process: the number is
    REPLAYED from the newest committed profiles/*_pmc_traffic.json (rocprofv3 --pmc passes, FETCH_SIZE and WRITE_SIZE in separate
    passes, corrected as MI355X_MICROARCH.md prescribes; the file records the command that produced it), scaled by rows -- and
    only while the kernel it was measured on is still the one this run launched for that layer: `forms` is the engine's form log
    of a rollout of THIS process ({kernel form: {rows: launches}}); a record whose `kernel` is not in it at `nb` rows is stale and
    NOT replayed.  Returns (bytes | None, description of the source)."""
    import glob

    try:
        newest = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")))[-1]
        with open(newest) as f:
            rec = json.load(f)
        name = os.path.basename(newest)
        kern = rec.get("kernel")
        if not kern or kern not in forms or nb not in forms[kern]:
            return None, (f"profiles/{name} was measured on {kern!r}; this run's rollout launched {sorted(forms)} -- the record is "
                          f"stale and was NOT replayed (re-run tools/pmc_traffic.sh)")
        cmd = rec.get("command", "command not recorded")
        return round(rec["hbm_bytes_per_row"] * nb), (f"replayed, not measured in this run: profiles/{name} (separate rocprofv3 --pmc "
                                                      f"passes of `{cmd}`; kernel {kern} still launched at {nb} rows), scaled to {nb} rows")
    except Exception as ex:
        return None, f"no usable profiles/*_pmc_traffic.json ({type(ex).__name__})"


_T0 = time.perf_counter()


def log(msg):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench +{time.perf_counter() - _T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


def build_model(nb, use_graph=True, channels=C, **extra):
    import dyffusion_amd as D

    c = channels
    F = D.UNet(num_input_channels=c, num_output_channels=c, num_conditional_channels=CS, spatial_shape=(H, W), **MODEL_KW)
    I = D.UNet(num_input_channels=2 * c, num_output_channels=c, num_conditional_channels=CS, spatial_shape=(H, W), **MODEL_KW)
    F.load_state_dict(random_state(F, 0))
    I.load_state_dict(random_state(I, 1))
    m = D.DYffusion(F, D.InterpolatorHandle(I, HORIZON), max_batch=nb, use_graph=use_graph, **dict(DIFFUSION_KW, **extra))
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
    try:
        ms, launches, fl, by = eng.time_kernel_in_rollout(kind, rows)
    except Exception as ex:
        if "not launched" not in str(ex):
            raise
        ms, launches, fl, by = 0.0, 0, 0.0, 0.0
    if launches == 0 or ms <= 0:  # e.g. the GroupNorm chain: fused into the convs (csrc/gn_fused.h), no launch of its own is left
        return {"kernel": name, "rows_per_launch": rows, "row_groups": groups, "launches": 0,
                "note": "no launch of this kernel class in the rollout"}
    r = {"kernel": name, "rows_per_launch": rows, "row_groups": groups, "avg_ms": round(ms, 4), "launches": launches,
         "algorithmic_bytes_per_launch": by,
         "hbm_gbps": round(by / ms / 1e6, 1), "hbm_frac": round(by / ms / 1e6 / 8000.0, 4)}
    if fl > 0:
        r.update({"bound": "mfma", "achieved": round(fl / ms / 1e9, 2), "peak": peak_tflops, "unit": "TFLOP/s",
                  "frac": round(fl / ms / 1e9 / peak_tflops, 4), "flops_per_launch": fl})
    else:
        r.update({"bound": "hbm", "achieved": r["hbm_gbps"], "peak": 8000.0, "unit": "GB/s", "frac": r["hbm_frac"]})
    return r


HBM_KERNELS = ("stem16_rows_kernel", "up2x_quad_kernel", "up2x_epilogue_kernel", "readout_dma_kernel", "layernorm_c_vec_kernel",
               "up2x_nearest_vec_kernel", "gn_apply_walk_kernel", "gn_apply_part_kernel")


def hbm_kernels(eng, nb):
    """north_star: "rocprof-reported HBM GB/s for the norm/activation kernels".  For every HBM-bound kernel the rollout launches (norm,
    activation, resample, readout: the launchers that open a KernelProf scope, csrc/common.h): HIP events around each of its launches in
    ONE eager rollout (dyf_time_named_kernel_in_rollout), ALGORITHMIC bytes (every operand once, 16-bit activations) / time -> GB/s and
    the fraction of the 8 TB/s HBM3E peak.  With row groups the launches cover one group's rows (as `_resnet_roofline`).  The rocprofv3
    kernel tables under profiles/ hold the same average durations."""
    rows = -(-nb // eng.row_groups)
    out = {}
    for name in HBM_KERNELS:
        ms, n, by = eng.time_named_kernel_in_rollout(name, rows)
        if n > 0 and ms > 0:
            out[name] = {"launches": n, "rows_per_launch": rows, "avg_us": round(1e3 * ms / n, 2),
                         "algorithmic_bytes_per_launch": round(by / n), "gbps": round(by / ms / 1e6, 1),
                         "frac_of_hbm_peak": round(by / ms / 1e6 / 8000.0, 4), "share_of_rollout_ms": round(ms, 3)}
    return out


def _time_rollouts(model, x0, reps, static=None):
    kw = {} if static is None else {"static_condition": static}
    model.sample(x0, **kw)  # captures the graph
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = model.sample(x0, **kw)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    assert all(bool(torch.isfinite(v).all()) for v in out.values()), "non-finite forecast"
    return dt


def projection(curve, total, parts=8):
    """What sharding a FIXED ensemble of `total` rows over `parts` GPUs would give against one GPU running all of it, from this GPU's
    own rates: parts * f(ceil(total / parts)) / f(total), f = fields/s at that many rows per GPU (the all-gather of the forecast stack,
    ~0.1 ms per MB over xGMI, is not in it).  The ceiling is total / ceil(total / parts) (50 rows on 8 GPUs: 7.14)."""
    per = -(-total // parts)
    if per not in curve or total not in curve:
        return None
    # the sharded job finishes when the slowest GPU -- one that runs `per` rows -- does: time = per * h / f(per); one GPU: total * h / f(total)
    return {"rows_total": total, "gpus": parts, "rows_per_gpu": per, "speedup": round((total / curve[total]) / (per / curve[per]), 3),
            "ceiling": round(total / per, 3)}


def ns_batch_curve(model, dev, nbs=(1, 4, 7, 10, 15, 25, 38, 50, 80, 120, 200)):
    """fields/s of the headline NS workload at smaller row counts on the SAME engine (graph per batch size; kernel forms are chosen
    per call from tile counts: split-K / implicit-GEMM forms at small batches) -- the rows one GPU gets when an ensemble is sharded."""
    g = torch.Generator().manual_seed(101)
    curve = {}
    for nb in nbs:
        x0 = torch.randn(nb, C, H, W, generator=g).to(dev)
        st = torch.rand(nb, CS, H, W, generator=g).to(dev)
        dt = _time_rollouts(model, x0, 2 if nb > 80 else 3 if nb >= 25 else 5, st)  # (> 80 rows: the engine is re-created at that size)
        curve[nb] = round(nb * HORIZON / dt, 1)
    log(f"NS batch curve (fields/s): {curve}")
    return curve


def bench_ns_c2(dev, nb):
    """The 2-channel Navier-Stokes variant (BASELINE.json writes '221x42x2ch'; the reference's data has 3 channels)."""
    m, _, _ = build_model(nb, channels=2)
    m.seed(2)
    g = torch.Generator().manual_seed(100)
    x0, st = torch.randn(nb, 2, H, W, generator=g).to(dev), torch.rand(nb, CS, H, W, generator=g).to(dev)
    dt = _time_rollouts(m, x0, 3, st)
    res = {"workload": "Navier-Stokes 221x42 with C=2 dynamics channels (+2 static), otherwise BASELINE configs[1]", "dtype": "bf16",
           "rows": nb, "fields_per_s": round(nb * HORIZON / dt, 1), "ms_per_rollout": round(1e3 * dt, 2)}
    log(f"NS C=2 NB={nb}: {res['ms_per_rollout']} ms per rollout -> {res['fields_per_s']} fields/s")
    m._engine.close()
    return res


def bench_ns_ar64(model, dev, b=4, n=20):
    """BASELINE configs[3] on ONE GPU: Navier-Stokes long rollout, prediction_horizon 64 with horizon 16 = FOUR autoregressive outer
    iterations of the headline rollout re-feeding t16 (forecasting_multi_horizon.py:114-229), the datamodule's boundary conditions
    (obstacle mask + parabolic inflow, physical_systems_benchmark.py:245-297) applied to every field by the device op."""
    import dyffusion_amd as D

    nb = b * n
    exp = D.MultiHorizonForecastingDYffusion(model, num_predictions=n)
    g = torch.Generator().manual_seed(3)
    dyn = torch.randn(b, 65, C, H, W, generator=g).to(dev)
    static = torch.rand(b, CS, H, W, generator=g).to(dev)
    meta = {"fixed_mask": (torch.rand(b, C, H, W, generator=g) < 0.06), "in_velocity": 1.0 + torch.rand(b, generator=g),
            "vertices": torch.rand(b, 2, H, W, generator=g) * 0.41}
    bc = D.PhysicalSystemsBoundaryConditions("navier-stokes", model._ensure_engine((H, W), nb))
    batch = {"dynamics": dyn, "condition": static, "metadata": meta}

    def run():
        batch["dynamics"] = dyn.clone()  # evaluation_step rescales the batch's dynamics after the first outer iteration
        return exp.evaluation_step(batch, prediction_horizon=64, boundary_conditions=bc, t0=torch.zeros(b), dt=torch.full((b,), 0.01))

    run()
    torch.cuda.synchronize()
    reps = 2
    t0 = time.perf_counter()
    for _ in range(reps):
        out = run()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    keys = [k for k in out if k.endswith("preds")]
    assert len(keys) == 64 and all(bool(torch.isfinite(out[k]).all()) for k in keys)
    res = {"workload": "BASELINE configs[3] shapes, 1 GPU: Navier-Stokes prediction_horizon 64 = 4 autoregressive outer iterations of the "
                       "h=16 rollout (240 network forwards), refine on, device boundary conditions on every field, ensemble of "
                       f"{n} x batch {b}", "dtype": "bf16", "rows": nb, "fields_per_s": round(nb * 64 / dt, 1),
           "ms_per_forecast": round(1e3 * dt, 2), "ms_per_outer_iteration": round(1e3 * dt / 4, 2)}
    log(f"NS AR-64 rows={nb}: {res['ms_per_forecast']} ms per 64-step forecast -> {res['fields_per_s']} fields/s")
    return res


def oisst_model(nb, dtype=None, row_groups=None):
    import dyffusion_amd as D

    kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
    F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
    I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2,
               attn_dropout=0.6, **kw)
    # conv gains halved (as for the 512^2 pair): the T=32 recursion of a random-init pair must stay inside fp16's range
    F.load_state_dict(_resnet_state(F, 0, 0.5))
    I.load_state_dict(_resnet_state(I, 1, 0.5))
    dtype = dtype or os.environ.get("DYF_BENCH_OISST_DTYPE", D.default_dtype_for(I))
    m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                    additional_interpolation_steps=25, refine_intermediate_predictions=False, max_batch=nb, dtype=dtype,
                    row_groups=row_groups)
    m.seed(2)
    return m, F, I, dtype


OISST_WORKLOAD = ("BASELINE configs[2] shapes: OISST 60x60x1, unet.Unet dim 64 mults (1,2,4), DYffusion h=7 k=25 (T=32), data+noise, "
                  "MC dropout on, hipGraph rollout")


def bench_oisst(dev, nb=300, reps=3):
    """BASELINE configs[2] shapes on ONE GPU: OISST 60x60x1, ResNet-UNet pair dim 64 mults (1,2,4), DYffusion h=7, k=25 (T=32: 32
    forecaster + 61 interpolator forwards), data+noise, MC dropout on, hipGraph; NB rows (50 members x 6 tiles)."""
    m, F, I, dtype = oisst_model(nb)
    x0 = torch.randn(nb, 1, 60, 60, generator=torch.Generator().manual_seed(3)).to(dev)
    dt = _time_rollouts(m, x0, reps)
    eng = m._engine
    nf, ni = eng.forward_counts()
    fl = nf * eng.net_flops(0) + ni * eng.net_flops(1)
    res = {"workload": OISST_WORKLOAD + ", 1 GPU", "dtype": dtype, "rows": nb, "row_groups": eng.row_groups,
           "net_forwards_per_rollout": nf + ni, "fields_per_s": round(nb * 7 / dt, 1), "ms_per_rollout": round(1e3 * dt, 2),
           "gflop_per_field": round(fl / 7 / 1e9, 2), "whole_rollout_tflops": round(nb * fl / dt / 1e12, 1),
           "roofline": _resnet_roofline(eng, 0, nb, "level-0 3x3 weight-standardised convs 64->64 @60x60 WITH the GroupNorm + FiLM + SiLU + "
                                                    "dropout (+ residual) of their Block fused into the epilogue (conv_gn16_kernel: 16 x 16 tiles, 3 workgroups per CU)",
                                        PEAK_BF16_TFLOPS),
           "roofline_groupnorm": _resnet_roofline(eng, 2, nb, "separate GroupNorm(8)+FiLM+SiLU+dropout(+residual) launches, 64 ch @60x60",
                                                  PEAK_BF16_TFLOPS),
           "hbm_kernels": hbm_kernels(eng, nb)}
    log(f"OISST NB={nb} ({dtype}): {res['ms_per_rollout']} ms per rollout -> {res['fields_per_s']} fields/s")
    eng.close()
    return res


def oisst_batch_curve(dev, first, nbs=(38, 75, 150)):
    """fields/s of the OISST workload at the row counts one GPU gets when 300 rows are sharded 8 / 4 / 2 ways: an engine of its own
    per point (the default row groups follow max_batch: 3 from 72 rows on, none below)."""
    curve = dict(first)
    for nb in nbs:
        m, _, _, _ = oisst_model(nb)
        x0 = torch.randn(nb, 1, 60, 60, generator=torch.Generator().manual_seed(3)).to(dev)
        dt = _time_rollouts(m, x0, 3)
        curve[nb] = round(nb * 7 / dt, 1)
        m._engine.close()
    log(f"OISST batch curve (fields/s): {curve}")
    return dict(sorted(curve.items()))


def synth512_model(nb):
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
    return m


SYNTH512_WORKLOAD = ("BASELINE configs[4] shapes: synthetic 512x512x4, unet.Unet dim 64 mults (1,2,4), DYffusion h=32, fp16 MFMA "
                     "conv/attention, MC dropout on, hipGraph rollout")


def bench_synth512(dev, nb=4, reps=1):
    """BASELINE configs[4] shapes on ONE GPU: synthetic 512x512x4ch, ResNet-UNet pair dim 64 mults (1,2,4) (bottleneck attention
    over 128^2 = 16 384 tokens), DYffusion h=32 (32 + 61 forwards), fp16, MC dropout on, hipGraph; NB rows."""
    m = synth512_model(nb)
    x0 = torch.randn(nb, 4, 512, 512, generator=torch.Generator().manual_seed(4)).to(dev)
    dt = _time_rollouts(m, x0, reps)
    eng = m._engine
    nf, ni = eng.forward_counts()
    fl = nf * eng.net_flops(0) + ni * eng.net_flops(1)
    res = {"workload": SYNTH512_WORKLOAD + ", 1 GPU", "dtype": "fp16", "rows": nb,
           "net_forwards_per_rollout": nf + ni, "fields_per_s": round(nb * 32 / dt, 1), "ms_per_rollout": round(1e3 * dt, 2),
           "gflop_per_field": round(fl / 32 / 1e9, 2), "whole_rollout_tflops": round(nb * fl / dt / 1e12, 1),
           "roofline": _resnet_roofline(eng, 1, nb, "flash_attention4_kernel (16 384 tokens, 4 heads x 32; dropout on the "
                                                    "probabilities in the interpolator's launches)", PEAK_BF16_TFLOPS),
           "roofline_conv": _resnet_roofline(eng, 0, nb, "level-0 3x3 weight-standardised convs 64->64 @512x512", PEAK_BF16_TFLOPS),
           "hbm_kernels": hbm_kernels(eng, nb)}
    log(f"512^2 NB={nb} (fp16): {res['ms_per_rollout']} ms per rollout -> {res['fields_per_s']} fields/s")
    eng.close()
    return res


PEAK_FP32_MFMA_TFLOPS = 157.3  # v_mfma_f32_32x32x2_f32 dense peak, MI355X_MICROARCH.md


def bench_train_steps(dev):
    """The TRAINING step of the forecaster objective (`DYffusion.p_losses` in train mode + `loss.backward()`: 2 interpolator + 2
    forecaster recorded forwards, backward through both forecaster passes and once through the frozen interpolator; dyffusion.py:496-567)
    for both backbones, fp32 activations on the fp32 matrix cores (csrc/train*.hip).  FLOP model: conv / matmul 2*MAC of a forward,
    x (4 forwards + 2 x 2 forecaster backward passes at 2 forward-equivalents each + 1 interpolator input-gradient pass at 1)."""
    import dyffusion_amd as D

    out = {}
    # ---- unet_simple at the NS shapes, B = 32
    B = 32
    m, F, I = build_model(B, use_graph=False, lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")
    m.train()
    g = torch.Generator().manual_seed(0)
    xt_last, cond = torch.randn(B, C, H, W, generator=g).to(dev), torch.randn(B, C, H, W, generator=g).to(dev)
    static = torch.rand(B, CS, H, W, generator=g).to(dev)
    t = torch.randint(0, HORIZON, (B,), generator=g).to(dev)

    def timed(step, reps):
        """One untimed step, then `reps` steps timed one by one (a step ends in a host read of the loss, so it is synchronous anyway):
        the MEDIAN -- a step is thousands of eager launches, and one host hiccup in a mean of two or three moved the line by 30 %."""
        step()
        torch.cuda.synchronize()
        each = []
        for _ in range(reps):
            t0 = time.perf_counter()
            loss = step()
            torch.cuda.synchronize()
            each.append(time.perf_counter() - t0)
        timed.each = [round(1e3 * v, 1) for v in each]
        return sorted(each)[len(each) // 2], loss

    def step_ns():
        o = m.p_losses(xt_last, cond, t, static_condition=static)
        o["loss"].backward()
        for p_ in m.model.parameters():
            p_.grad = None
        return float(o["loss"])

    dt, loss = timed(step_ns, 3)
    fwd = m._engine.net_flops(0)  # both nets: ~48.2 GF per row
    fl = B * fwd * (4 + 2 * 2 * 2 + 1)
    out["unet_simple_ns"] = {"workload": f"p_losses + backward, NS 221x42 shapes (unet_simple dim 64 @256^2), B={B}, both loss terms, fp32",
                             "batch": B, "ms_per_step": round(1e3 * dt, 1), "loss": round(loss, 4), "achieved": round(fl / dt / 1e12, 1),
                             "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3),
                             "samples_per_s": round(B / dt, 1), "each_ms": timed.each}
    log(f"train step unet_simple B={B}: {1e3 * dt:.1f} ms")
    # the same step with the training convs' operands rounded to 16 bits while they are staged (opt-in: train_precision=16; fp32
    # tensors and master weights, fp32 accumulation; csrc/train_halo16.hip + train_gemm.hip t_gemm_mfma16; how far the gradients
    # move: tests/test_gpu_training.py test_training_step_with_16bit_conv_operands_tracks_the_fp32_step)
    m._engine.train_set_precision("16-mixed")  # the reference's trainer.precision=16: C-ABI dyf_train_set_precision
    try:
        dt16, loss16 = timed(step_ns, 5)
    finally:
        m._engine.train_set_precision(32)
    out["unet_simple_ns_16bit_operands"] = {"workload": out["unet_simple_ns"]["workload"].replace(", fp32", ", fp32 tensors, conv operands "
                                                                                                    "rounded to bf16 in the kernels (train_precision=16)"),
                                            "batch": B, "ms_per_step": round(1e3 * dt16, 1), "loss": round(loss16, 4),
                                            "achieved": round(fl / dt16 / 1e12, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                                            "frac": round(fl / dt16 / 1e12 / PEAK_BF16_TFLOPS, 4),  # against the 16-bit MFMA peak: the operands are 16-bit
                                            "samples_per_s": round(B / dt16, 1), "speedup_vs_fp32": round(dt / dt16, 2), "each_ms": timed.each}
    log(f"train step unet_simple B={B}, 16-bit conv operands: {1e3 * dt16:.1f} ms")
    m._engine.close()
    del m
    # ---- unet.Unet at the OISST shapes: B = 8 (round 3's point) and B = 64, the reference's training batch
    # (src/configs/experiment/oisst_pacific.yaml:11).  The step is ~4 300 small launches: at B = 8 it is bound by the HOST's launch
    # rate (56 ms on one box, 119 ms on another), at B = 64 by the kernels.
    kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
    F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
    I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2,
               attn_dropout=0.6, **kw)
    F.load_state_dict(_resnet_state(F, 0, 0.5))
    I.load_state_dict(_resnet_state(I, 1, 0.5))
    for B, key in ((8, "unet_resnet_oisst"), (64, "unet_resnet_oisst_b64")):
        m2 = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                         additional_interpolation_steps=25, lambda_reconstruction=0.5, lambda_reconstruction2=0.5, loss_function="l1",
                         max_batch=B)
        m2.train()
        x2, c2 = torch.randn(B, 1, 60, 60, generator=g).to(dev), torch.randn(B, 1, 60, 60, generator=g).to(dev)
        t2 = torch.randint(0, m2.num_timesteps, (B,), generator=g).to(dev)

        def step_rn():
            o = m2.p_losses(x2, c2, t2, static_condition=None)
            o["loss"].backward()
            for p_ in m2.model.parameters():
                p_.grad = None
            return float(o["loss"])

        dt, loss = timed(step_rn, 3)
        fl = B * (m2._engine.net_flops(0) * (2 + 2 * 2 * 2) + m2._engine.net_flops(1) * (2 + 1))
        out[key] = {"workload": f"p_losses + backward, OISST 60x60 shapes (unet.Unet dim 64 mults (1,2,4)), B={B}, both loss terms, fp32",
                    "batch": B, "ms_per_step": round(1e3 * dt, 1), "loss": round(loss, 4), "achieved": round(fl / dt / 1e12, 1),
                    "peak": PEAK_FP32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / dt / 1e12 / PEAK_FP32_MFMA_TFLOPS, 3),
                    "samples_per_s": round(B / dt, 1), "each_ms": timed.each}
        log(f"train step unet.Unet B={B}: {1e3 * dt:.1f} ms")
        if B == 64:  # the same step with the conv operands rounded to 16 bits while staged (opt-in; same launchers as the NS step)
            m2._engine.train_set_precision("16-mixed")
            try:
                dt16, loss16 = timed(step_rn, 5)
            finally:
                m2._engine.train_set_precision(32)
            out[key + "_16bit_operands"] = {"workload": out[key]["workload"].replace(", fp32", ", fp32 tensors, conv operands rounded to "
                                                                                     "bf16 in the kernels (train_precision=16)"),
                                            "batch": B, "ms_per_step": round(1e3 * dt16, 1), "loss": round(loss16, 4),
                                            "loss_fp32_operands": round(loss, 4), "achieved": round(fl / dt16 / 1e12, 1),
                                            "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s", "frac": round(fl / dt16 / 1e12 / PEAK_BF16_TFLOPS, 4),
                                            "samples_per_s": round(B / dt16, 1), "speedup_vs_fp32": round(dt / dt16, 2), "each_ms": timed.each}
            log(f"train step unet.Unet B={B}, 16-bit conv operands: {1e3 * dt16:.1f} ms (loss {loss16:.4f} vs {loss:.4f})")
        m2._engine.close()
        del m2
    return out


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
    interpolator forward is timed at 16 / 32 / 64 / 128 threads (those that the host has; all of them on hosts of <= 96 threads; the
    256-thread setting of the GPU boxes was measured once at 17 s per forward against 0.13 s on 32 and is not re-timed on every run)
    and the fastest setting is used -- the sweep is reported in the line (`thread_sweep_s`)."""
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
        sweep = {}
        for nt in ([int(forced)] if forced else sorted({min(t, ncpu) for t in (16, 32, 64, 128)} | ({ncpu} if ncpu <= 96 else set()))):
            torch.set_num_threads(nt)
            xi = torch.cat([x4[:1], x4[:1]], 1)
            i_fn(xi, torch.ones(1), c4[:1])  # warm-up (thread pool, mkldnn primitives)
            t0 = time.perf_counter()
            i_fn(xi, torch.ones(1), c4[:1])
            dt = time.perf_counter() - t0
            log(f"cpu baseline: one interpolator forward on {nt} threads {dt:.3f} s")
            sweep[nt] = round(dt, 4)
            if best is None or dt < best[1]:
                best = (nt, dt)
        torch.set_num_threads(best[0])
        # 1 warm-up rollout (untimed: thread pool, oneDNN primitive cache, allocator), then >= 3 timed NB = 1 rollouts, each
        # timed on its own so that the spread is in the line; NB = 4 (reported beside it, never the headline unless faster): 1 rollout
        res, per = {}, {}
        sampler.sample_loop(f_fn, i_fn, x4[:1], c4[:1], cfg)
        for nb, budget, min_reps, max_reps in ((1, 16.0, 3, 4), (4, 10.0, 1, 1)):  # (one NB = 4 rollout is 30 s on the GPU boxes' hosts)
            times, t_all = [], time.perf_counter()
            while len(times) < min_reps or (time.perf_counter() - t_all < budget and len(times) < max_reps):
                t0 = time.perf_counter()
                sampler.sample_loop(f_fn, i_fn, x4[:nb], c4[:nb], cfg)
                times.append(time.perf_counter() - t0)
            dt = sum(times)
            res[nb] = (len(times), dt, len(times) * nb * HORIZON / dt)
            per[nb] = [round(nb * HORIZON / t, 4) for t in times]
            log(f"cpu baseline NB={nb}: {len(times)} timed rollout(s) in {dt:.1f} s = {res[nb][2]:.3f} fields/s (each: {per[nb]})")
    top = max(res, key=lambda k: res[k][2])
    return {"value": round(res[top][2], 4), "unit": "fields/s", "cores": best[0], "kind": "port",
            "host_cores": ncpu, "cpu_model": cpu_model(), "thread_sweep_s": {str(k): v for k, v in sweep.items()},
            "fields_per_s_nb1": round(res[1][2], 4), "fields_per_s_nb4": round(res[4][2], 4),
            "fields_per_s_each_rollout": {"nb1": per[1], "nb4": per[4]},
            "spread_nb1": round((max(per[1]) - min(per[1])) / (sum(per[1]) / len(per[1])), 4),
            "sample": f"1 untimed warm-up rollout, then full h={HORIZON} rollouts (60 network forwards each), fp32, MC dropout on: NB=1 x{res[1][0]} "
                      f"in {res[1][1]:.1f} s, NB=4 x{res[4][0]} in {res[4][1]:.1f} s (value = the faster of the two means); {best[0]} of {ncpu} host "
                      f"threads (fastest of {'/'.join(str(k) for k in sweep)} on one forward)"}


def self_launch(n):
    """Replace this process by `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port P
    bench.py <the same arguments>` (P: a free port picked here).  When the host shows fewer GPUs than ranks (the one-GPU rehearsal of
    the N > 1 branch: ranks wrap onto the devices there are) RCCL cannot build the communicator ("Duplicate GPU detected"), so
    torch.distributed is pointed at gloo unless the caller chose a backend."""
    import socket

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: what RCCL needs on this host driver
    env.setdefault("OMP_NUM_THREADS", "8")
    if torch.cuda.device_count() < n and "DYF_DIST_BACKEND" not in env:
        env["DYF_DIST_BACKEND"] = "gloo"
        log(f"{torch.cuda.device_count()} visible GPU(s) for {n} ranks: ranks share devices, torch.distributed over gloo")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log("self-launch: " + " ".join(cmd))
    sys.stdout.flush()
    sys.stderr.flush()
    os.execve(sys.executable, cmd, env)


def claim_stdout():
    """File descriptor 1 belongs to the ONE JSON line.  Whatever else writes to stdout in this process -- gloo's "[Gloo] Rank 1 is
    connected to ..." notes, RCCL's version banner at communicator creation, a stray print of a library -- is sent to stderr from here
    on (at the descriptor level, so C and C++ code is covered); the returned file object is the original stdout."""
    sys.stdout.flush()
    keep = os.dup(1)
    os.dup2(2, 1)
    return os.fdopen(keep, "w")


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
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="only the headline NS line (+ rooflines): no batch curve, no other configs, no training steps")
    ap.add_argument("--no-graph", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if "WORLD_SIZE" in os.environ or args.gpus < 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: the launcher's rank count and --gpus disagree")
        # plain `python bench.py --gpus N` (no launcher): re-execute this file under torch.distributed.run, one rank per GPU --
        # the same command line the driver uses for N > 1; rank 0 of the child prints the ONE JSON line on the inherited stdout
        return self_launch(args.gpus)
    json_out = claim_stdout()
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
    model, F, I = build_model(max(nb, args.nb if world > 1 else nb), use_graph=not args.no_graph)
    log("model built")
    gather = os.environ.get("DYF_BENCH_GATHER", "1") == "1"  # =0: time the rollouts without the exchange (A/B)
    want_engine_comm = world > 1 and gather and os.environ.get("DYF_BENCH_EXCHANGE", "engine" if dist.get_backend() == "nccl" else "torch") == "engine"

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    def sharded_run(mdl, x_full, s_full, hw, horizon, steps, warmup, gather=gather):
        """K timed steps of `mdl` on the rows of x_full sharded over the ranks: every rank samples its block of global rows (same seed,
        global-row dropout streams) and, N > 1, the forecast stack is all-gathered inside EVERY step (one collective, issued by the
        engine on the rollout's stream when it owns a communicator).  Returns (seconds MAX over ranks, exchange used, ncclCommCount)."""
        rows = x_full.shape[0]
        lo, hi = shard_rows(rows, world, rank)
        exch, seen = "none", 0
        # (growing the engine replaces it and drops its communicator: do that HERE, on every rank alike, before the communicator check)
        mdl._ensure_engine(hw, rows_per_rank(rows, world))
        if world > 1 and gather:
            exch = "torch"
            if want_engine_comm:
                if mdl.engine_comm_world() == world or init_engine_comm(mdl, hw, rows):
                    exch, seen = "engine", mdl._engine.comm_count()
                else:
                    log(f"engine-owned communicator unavailable ({getattr(mdl, '_comm_error', '?')}): torch.distributed exchange")

        def step():
            if world > 1 and gather:
                preds = sample_sharded(mdl, x_full, s_full, exchange=exch)
                assert next(iter(preds.values())).shape[0] == rows
                return preds
            mdl.set_row_offset(lo)
            return mdl.sample(x_full[lo:hi], **({} if s_full is None else {"static_condition": s_full[lo:hi]}))

        for _ in range(warmup):
            step()
            torch.cuda.synchronize()
        fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            preds = step()
        fence()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        assert all(bool(torch.isfinite(v).all()) for v in preds.values()), "non-finite forecast"
        return dt, exch, seen

    def rank0_alone(make, x_full, s_full, steps, warmup=1):
        """The N = 1 equivalent measured INSIDE the N > 1 job, same node, same process: rank 0 samples all rows of x_full by itself
        (no exchange) while the other ranks wait at the fence -- on an engine of its own (`make(rows)`, sized for those rows exactly as
        a 1-GPU job would size it; the sharded model and its communicator are not touched).  Seconds for `steps` steps (on every rank)."""
        kw = {} if s_full is None else {"static_condition": s_full}
        dt = 0.0
        fence()
        if rank == 0:
            mdl = make(x_full.shape[0])
            mdl.seed(2)
            for _ in range(warmup):
                mdl.sample(x_full, **kw)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                mdl.sample(x_full, **kw)
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            mdl._engine.close()
            del mdl
            torch.cuda.empty_cache()
        fence()
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    def strong_entry(mdl, x_full, s_full, hw, horizon, steps, alone=None):
        """One fixed ensemble sharded over the ranks: fields/s with the exchange, the same steps WITHOUT the exchange (its own cost =
        the difference), and rank 0 alone on the whole ensemble -> a same-node speed-up."""
        rows = x_full.shape[0]
        d, ex, seen = sharded_run(mdl, x_full, s_full, hw, horizon, steps, 1)
        ent = {"total_rows": rows, "rows_per_gpu": rows_per_rank(rows, world), "scaling": "strong",
               "rows_by_rank": [b - a for a, b in (shard_rows(rows, world, r) for r in range(world))],
               "fields_per_s": round(rows * horizon * steps / d, 1), "ms_per_step": round(1e3 * d / steps, 3), "exchange": ex, "nranks_seen": seen,
               "row_groups": mdl._engine.row_groups}
        if gather:
            d0, _, _ = sharded_run(mdl, x_full, s_full, hw, horizon, steps, 1, gather=False)
            ent["ms_per_step_without_exchange"] = round(1e3 * d0 / steps, 3)
            ent["exchange_ms_per_step"] = round(1e3 * (d - d0) / steps, 3)
        if alone is not None:
            d1 = rank0_alone(alone, x_full, s_full, steps)
            ent["rank0_alone_fields_per_s"] = round(rows * horizon * steps / d1, 1)
            ent["rank0_alone_ms_per_step"] = round(1e3 * d1 / steps, 3)
            ent["speedup_vs_rank0_alone"] = round(d1 / d, 3)
            ent["ceiling"] = round(rows / rows_per_rank(rows, world), 3)
        return ent

    # every rank holds the full (total_rows, ...) inputs (111 KB per row) and the same seed
    g = torch.Generator().manual_seed(100)
    x0 = torch.randn(total_rows, C, H, W, generator=g).to(dev)
    static = torch.rand(total_rows, CS, H, W, generator=g).to(dev)
    model.seed(2)
    model._ensure_engine((H, W), max(nb, args.nb if world > 1 else nb))
    log("engine created, weights uploaded")
    dt, exchange, nranks_seen = sharded_run(model, x0, static, (H, W), HORIZON, args.steps, args.warmup)
    log(f"timed region done: {dt:.3f} s for {args.steps} steps")

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
    if world > 1:
        result["nranks_seen"] = nranks_seen  # ncclCommCount of the engine's communicator (0: the torch.distributed route ran)
        result["rows_by_rank"] = [b - a for a, b in (shard_rows(total_rows, world, r) for r in range(world))]
        result["devices_visible"] = ndev
        if gather:  # the all-gather's own cost: the same K steps without it
            d0, _, _ = sharded_run(model, x0, static, (H, W), HORIZON, args.steps, 1, gather=False)
            result["ms_per_step_without_exchange"] = round(1e3 * d0 / args.steps, 3)
            result["exchange_ms_per_step"] = round(1e3 * (dt - d0) / args.steps, 3)
        # N = 1 equivalent inside this run: rank 0 alone on ONE rank's rows (weak scaling: the per-GPU work is what stays fixed) or on
        # the whole fixed ensemble (strong scaling)
        xa, sa = (x0, static) if strong else (x0[:nb], static[:nb])
        d1 = rank0_alone(lambda r: build_model(r, use_graph=not args.no_graph)[0], xa, sa, args.steps)
        result["rank0_alone"] = {"rows": xa.shape[0], "fields_per_s": round(xa.shape[0] * HORIZON * args.steps / d1, 1),
                                 "ms_per_step": round(1e3 * d1 / args.steps, 3),
                                 "speedup_of_this_line": round((fields / dt) / (xa.shape[0] * HORIZON * args.steps / d1), 3),
                                 "ideal": world if not strong else round(total_rows / nb, 3)}

    extras = not args.no_extra_configs
    if world > 1 and extras:
        # ---- N > 1: the other multi-GPU workloads of BASELINE.json, each sharded over the ranks through the same exchange.
        # STRONG scaling of the headline workload: a fixed ensemble (the reference's 50 members; its 80-row evaluation batch) split over
        # the ranks -- what north_star's ">= 6x at 8 GPUs" is about; the weak line above keeps 80 rows per GPU.
        # 200 rows = the reference's NS TEST call (eval_batch_size 4 x num_predictions 50: experiment/navier_stokes.yaml:12, mode/test.yaml:9)
        result["strong"] = {}
        for m_rows in [int(v) for v in os.environ.get("DYF_BENCH_STRONG_ROWS", "50,80,200").split(",")]:
            gs = torch.Generator().manual_seed(100)
            xs, ss = torch.randn(m_rows, C, H, W, generator=gs).to(dev), torch.rand(m_rows, CS, H, W, generator=gs).to(dev)
            try:
                result["strong"][f"ensemble_{m_rows}"] = strong_entry(model, xs, ss, (H, W), HORIZON, max(3, args.steps),
                                                                      alone=lambda r: build_model(r, use_graph=not args.no_graph)[0])
            except Exception as ex:  # identical on every rank (same shapes)
                result["strong"][f"ensemble_{m_rows}"] = {"error": f"{type(ex).__name__}: {ex}"}
            log(f"strong scaling, {m_rows} rows over {world} ranks: {result['strong'][f'ensemble_{m_rows}']}")
        model._engine.close()
        del model
        torch.cuda.empty_cache()
        # (DYF_BENCH_OISST_ROWS / DYF_BENCH_SYNTH_ROWS shrink the two ensembles for the 2-ranks-on-one-GPU rehearsal of this branch,
        # tests/test_gpu_bench_multirank.py; the driver's SCALE runs use the defaults)
        oisst_rows, synth_rows = int(os.environ.get("DYF_BENCH_OISST_ROWS", "300")), int(os.environ.get("DYF_BENCH_SYNTH_ROWS", "8"))
        for key, make, shape, rows, horizon, wl in (
                ("config2_oisst", lambda r: oisst_model(r)[0], (1, 60, 60), oisst_rows, 7, OISST_WORKLOAD),
                ("config4_synth512", synth512_model, (4, 512, 512), synth_rows, 32, SYNTH512_WORKLOAD)):
            try:
                rpr = rows_per_rank(rows, world)
                mdl = make(rpr)
                xf = torch.randn(rows, *shape, generator=torch.Generator().manual_seed(3)).to(dev)
                mdl._ensure_engine(shape[1:], rpr)
                reps = 3 if key == "config2_oisst" else 1
                # (rank 0 alone on the whole ensemble only for OISST; the 512^2 ensemble alone is 8 x 16 s of rollout)
                ent = strong_entry(mdl, xf, None, shape[1:], horizon, reps, alone=make if key == "config2_oisst" else None)
                ent.update({"workload": wl + f", {rows} rows sharded over {world} GPUs", "ms_per_rollout": ent["ms_per_step"]})
                result[key] = ent
                log(f"{key} on {world} ranks: {result[key]}")
                mdl._engine.close()
                del mdl
            except Exception as ex:  # identical on every rank (same shapes): the headline line must survive
                result[key] = {"error": f"{type(ex).__name__}: {ex}"}
            torch.cuda.empty_cache()

    if rank == 0 and world == 1:
        # roofline of the dominant kernel: conv_halo_rows_kernel<0> (dense form; dec3 + dec4 = 27 % of a forward's time), largest launch
        # dec4.  achieved = algorithmic FLOPs of one launch / average duration of that launch INSIDE the rollout: HIP events around every
        # dec4 conv (60 launches: 16 forecaster + 44 interpolator forwards, MC dropout on) of one eagerly launched rollout on the launch
        # stream (dyf_time_layer_in_rollout); the isolated back-to-back figure is kept beside it.  The sparse-column instance of the
        # same kernel (dec5, only the output columns the readout reads) is reported too.
        def layer_roofline(layer, name):
            _, fl, by = eng.time_conv_layer(1, layer, nb, iters=1)
            ms, launches = eng.time_layer_in_rollout(layer, nb)
            ms_iso, _, _ = eng.time_conv_layer(1, layer, nb, iters=10)
            log(f"{name}: {ms:.3f} ms per launch in the rollout ({launches} launches), {ms_iso:.3f} ms isolated back-to-back")
            return {"bound": "mfma", "kernel": name, "achieved": round(fl / ms / 1e9, 2), "peak": PEAK_BF16_TFLOPS,
                    "unit": "TFLOP/s", "frac": round(fl / ms / 1e9 / PEAK_BF16_TFLOPS, 4), "avg_ms": round(ms, 4),
                    "launches": launches, "avg_ms_isolated": round(ms_iso, 4), "flops_per_launch": fl,
                    "algorithmic_bytes_per_launch": by}

        eng.form_log(True)
        result["roofline"] = layer_roofline(10, "conv_halo_rows_kernel<0> (dec4: fused x2-upsample + 3x3 conv, 256->128 ch, 64^2->128^2)")
        forms = eng.form_log_read()  # the kernel forms of an eager rollout of THIS run (dyf_time_layer_in_rollout): guards the PMC replay
        eng.form_log(False)
        result["roofline"]["traffic"], result["roofline"]["traffic_source"] = pmc_traffic(nb, forms)
        result["roofline_dec5_sparse"] = layer_roofline(
            11, "conv_halo_rows_mixed_kernel = conv_halo_rows_kernel<1, SH> (dec5: fused x2-upsample + 3x3 conv, 256->64 ch, 128^2->256^2, "
                "104 of 256 output columns as 3 x 16 + 4 list entries per phase)")
        try:  # the HBM-bound kernels of the headline rollout (norm / activation / resample / readout)
            result["hbm_kernels"] = hbm_kernels(eng, nb)
        except Exception as ex:
            result["hbm_kernels"] = {"error": f"{type(ex).__name__}: {ex}"}
        if extras:
            def guarded(key, fn):
                try:
                    result[key] = fn()
                except Exception as ex:  # the headline line must survive a failure here
                    result[key] = {"error": f"{type(ex).__name__}: {ex}"}
                torch.cuda.empty_cache()

            # the NS engine is CLOSED before the OISST line (dyf_engine_destroy; `del` alone would not: the network modules hold it
            # too): a live engine's captured graph keeps a hardware queue, and the three concurrent row groups of the OISST rollout
            # then share the 4 queues of the process with it (3 660 vs 2 870 fields/s measured).  So: NS extras that
            # need this engine first, then close it, then OISST FIRST among the other engines.
            guarded("batch_curve", lambda: {"unit": "fields/s", "navier_stokes": ns_batch_curve(model, dev)})
            guarded("config3_ns_ar64", lambda: bench_ns_ar64(model, dev))
            model._engine.close()  # (the 120- / 200-row points of the curve re-created the engine: `eng` is the closed 80-row one)
            del model
            torch.cuda.empty_cache()
            guarded("config2_oisst", lambda: bench_oisst(dev))
            curve = result["batch_curve"]
            if "error" not in curve and "fields_per_s" in result["config2_oisst"]:
                try:
                    curve["oisst"] = oisst_batch_curve(dev, {300: result["config2_oisst"]["fields_per_s"]})
                except Exception as ex:
                    curve["oisst"] = {"error": f"{type(ex).__name__}: {ex}"}
                torch.cuda.empty_cache()
                oi = {k: v for k, v in curve["oisst"].items() if isinstance(k, int)}
                result["strong_scaling_projection"] = {
                    "what": "time of ONE GPU on the whole ensemble / time of the slowest of 8 GPUs on its ceil(rows / 8) share, from this "
                            "GPU's batch curve (the 1-collective all-gather of the forecast stack is not in it)",
                    # 200 rows = the reference's NS TEST call: eval_batch_size 4 x num_predictions 50 (experiment/navier_stokes.yaml:12,
                    # mode/test.yaml:9); 80 rows = its validation call (4 x 20); 50 = one batch item's members
                    "navier_stokes_200": projection(curve["navier_stokes"], 200), "navier_stokes_120": projection(curve["navier_stokes"], 120),
                    "navier_stokes_80": projection(curve["navier_stokes"], 80), "navier_stokes_50": projection(curve["navier_stokes"], 50),
                    "oisst_300": projection(oi, 300)}
            guarded("config1_ns_c2", lambda: bench_ns_c2(dev, nb))
            guarded("config4_synth512", lambda: bench_synth512(dev))
            guarded("train_step", lambda: bench_train_steps(dev))
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline(F, I)
    if rank == 0:
        json_out.write(json.dumps(result) + "\n")
        json_out.flush()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
