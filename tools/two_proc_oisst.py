"""two processes on ONE GPU (no process group needed): each runs OISST rollouts and prints per-call wall time.
usage: two_proc_oisst.py <rows> <row_groups> <calls>; env DYF_GN_FUSED"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import torch
import bench
rows, groups, calls = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
tag = sys.argv[4] if len(sys.argv) > 4 else "?"
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m, F, I, dtype = bench.oisst_model(rows, row_groups=groups)
x0 = torch.randn(rows, 1, 60, 60, generator=torch.Generator().manual_seed(3)).to(dev)
m._ensure_engine((60, 60), rows)
eng = m._engine
for c in range(calls):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    eng.form_log(True)
    out = m.sample(x0)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    fm = eng.form_log_read(); eng.form_log(False)
    nf = sum(sum(v.values()) for k, v in fm.items() if 'gn_fused' in k); ng = sum(sum(v.values()) for k, v in fm.items() if k.startswith('gn_'))
    print(f"[{tag} pid {os.getpid()}] call {c}: fused-form notes {nf}, gn_* notes {ng}; {dt*1e3:.1f} ms  groups {eng.row_groups} finite {all(bool(torch.isfinite(v).all()) for v in out.values())}", flush=True)
