"""Phase timeline of the fused-GroupNorm level-0 conv (conv_up_halo_kernel<5, 2>) inside an OISST forward -- needs the
HALO_EXP_TIMELINE experiment build: tools/build_variant.sh tl conv_up_halo.hip -DHALO_EXP_TIMELINE, DYF_LIB_F16=tools/variants/libvar_tl_f16.so.
usage: python tools/timeline_oisst.py ROWS [LAUNCH_INDEX]   -> per-phase cycle statistics of that launch's waves"""
import os
import struct
import sys

import numpy as np
import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 100
which = int(sys.argv[2]) if len(sys.argv) > 2 else 40
out = "/tmp/halo_tl.bin"
os.environ["DYF_TIMELINE_DUMP"] = f"{out}:{which}"
os.environ.setdefault("DYF_ROW_GROUPS", "1")
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()
import bench  # noqa: E402

m, F, I, dtype = bench.oisst_model(rows)
m._engine_opts["use_graph"] = False
x0 = torch.randn(rows, 1, 60, 60).cuda()
m.sample(x0)
torch.cuda.synchronize()
raw = open(out, "rb").read()
total, n, has_res, drop = struct.unpack("4i", raw[:16])
tl = np.frombuffer(raw[16:], dtype=np.uint64).reshape(-1, 4, 8).astype(np.int64)[:total]
print(f"launch {which}: {total} workgroups, {n} rows, residual={has_res}, dropout mode {drop}")
t0 = tl[:, :, 0].min()
names = ["start -> halo landed", "K loop", "statistics (phase A)", "sweep + barrier (phase B)", "epilogue (phase C)"]
seg = [(0, 1), (1, 2), (2, 3), (3, 4), (4, 5)]
for nm, (a, b) in zip(names, seg):
    d = (tl[:, :, b] - tl[:, :, a]).reshape(-1)
    print(f"  {nm:28s} mean {d.mean():9.0f}  p10 {np.percentile(d, 10):9.0f}  p50 {np.percentile(d, 50):9.0f}  p90 {np.percentile(d, 90):9.0f} cycles")
life = (tl[:, :, 5] - tl[:, :, 0]).reshape(-1)
print(f"  wave lifetime mean {life.mean():.0f}; launch span {(tl[:, :, 5].max() - t0)} cycles (s_memtime ticks)")
sw = (tl[:, 0, 6] - tl[:, 0, 3])
print(f"  wave 0: stats published -> coefficients ready (sweep) mean {sw.mean():.0f} p90 {np.percentile(sw, 90):.0f}")
st = (tl[:, 0, 0] - t0)
order = np.argsort(st)
print("  start times of workgroups (sorted) at 0/25/50/75/100 %:", [int(st[order[int(q * (total - 1))]]) for q in (0, .25, .5, .75, 1)])
en = (tl[:, :, 5].max(axis=1) - t0)
print("  end times at 0/25/50/75/100 %:", [int(np.sort(en)[int(q * (total - 1))]) for q in (0, .25, .5, .75, 1)])
