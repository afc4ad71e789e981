"""NS headline workload (BASELINE configs[1]) at NB rows under the kernel-form switches of the environment (DYF_ROW_GROUPS, ...): fields/s of
hipGraph rollouts.  usage: python tools/bench_ns_rows.py [NB] [steps]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()
import bench  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 80
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model, F, I = bench.build_model(nb, use_graph=True)
model.seed(2)
g = torch.Generator().manual_seed(1)
x0 = torch.randn(nb, bench.C, bench.H, bench.W, generator=g).cuda()
st = torch.rand(nb, bench.CS, bench.H, bench.W, generator=g).cuda()
for _ in range(3):
    model.sample(x0, static_condition=st)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    out = model.sample(x0, static_condition=st)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
print(f"NS NB={nb} groups={model._engine.row_groups}: {1e3 * dt:.2f} ms per rollout -> {nb * bench.HORIZON / dt:.1f} fields/s, finite={bool(torch.isfinite(next(iter(out.values()))).all())}")
