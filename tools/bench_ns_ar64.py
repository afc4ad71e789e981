"""Timing of BASELINE configs[3]: Navier-Stokes long rollout, prediction_horizon = 64 with horizon = 16, i.e. four
autoregressive outer iterations of the headline rollout re-feeding t16 (forecasting_multi_horizon.py:114-229), with the
NS boundary condition applied to every field.  usage: python tools/bench_ns_ar64.py [B] [N]   (rows = N * B)"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
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
# the reference's NS boundary conditions (physical_systems_benchmark.py:245-297) on the GPU: obstacle / wall mask zeroed,
# parabolic inflow on the first grid row, one masked-write kernel per field (dyffusion_amd/boundary.py)
meta = {"fixed_mask": (torch.rand(B, bench.C, bench.H, bench.W, generator=g) < 0.06),
        "in_velocity": 1.0 + torch.rand(B, generator=g), "vertices": torch.rand(B, 2, bench.H, bench.W, generator=g) * 0.41}
bc = D.PhysicalSystemsBoundaryConditions("navier-stokes", model._ensure_engine((bench.H, bench.W), nb))
batch = {"dynamics": dyn, "condition": static, "metadata": meta}


def run():
    batch["dynamics"] = dyn.clone()  # evaluation_step scales the batch's dynamics by 1e6 after the first outer iteration
    return exp.evaluation_step(batch, prediction_horizon=64, boundary_conditions=bc, t0=torch.zeros(B), dt=torch.full((B,), 0.01))


run()
torch.cuda.synchronize()
reps = 2
t0 = time.perf_counter()
for _ in range(reps):
    out = run()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
keys = [k for k in out if k.endswith("preds")]
print(f"NS AR-64 rows={nb}: {dt * 1e3:.1f} ms per 64-step forecast (4 rollouts of 60 forwards) -> {nb * 64 / dt:.1f} fields/s, "
      f"{len(keys)} horizons, shape {tuple(out['t64_preds'].shape)}, finite={all(torch.isfinite(out[k]).all() for k in keys)}")
