"""Timing of BASELINE configs[3]: Navier-Stokes long rollout, prediction_horizon = 64 with horizon = 16, i.e. four
autoregressive outer iterations of the headline rollout re-feeding t16 (forecasting_multi_horizon.py:114-229), with the
NS boundary condition applied to every field.  usage: python tools/bench_ns_ar64.py [B] [N]   (rows = N * B)"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import dyffusion_amd as D  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
N = int(sys.argv[2]) if len(sys.argv) > 2 else 20
nb = B * N
model, F, I = bench.build_model(nb)
exp = D.MultiHorizonForecastingDYffusion(model, num_predictions=N)
g = torch.Generator().manual_seed(3)
dyn = torch.randn(B, 65, bench.C, bench.H, bench.W, generator=g).cuda()
static = torch.rand(B, bench.CS, bench.H, bench.W, generator=g).cuda()
mask = (static[:, :1] > 0.05).float()  # obstacle mask: velocity is zero inside obstacles


def bc(preds, targets=None, metadata=None, time=None):  # physical_systems_benchmark.py:245-297, tensor form
    if preds.dim() == 5 or preds.shape[0] == B:  # (N, B, C, H, W) broadcasts against (B, 1, H, W)
        return preds * mask
    return preds * mask.repeat(preds.shape[0] // B, 1, 1, 1)


batch = {"dynamics": dyn, "condition": static}
exp.evaluation_step(batch, prediction_horizon=64, boundary_conditions=bc, return_targets=False)
torch.cuda.synchronize()
reps = 2
t0 = time.perf_counter()
for _ in range(reps):
    out = exp.evaluation_step(batch, prediction_horizon=64, boundary_conditions=bc, return_targets=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
keys = [k for k in out if k.endswith("preds")]
print(f"NS AR-64 rows={nb}: {dt * 1e3:.1f} ms per 64-step forecast (4 rollouts of 60 forwards) -> {nb * 64 / dt:.1f} fields/s, "
      f"{len(keys)} horizons, shape {tuple(out['t64_preds'].shape)}, finite={all(torch.isfinite(out[k]).all() for k in keys)}")
