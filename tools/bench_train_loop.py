"""Time whole training-loop iterations (p_losses incl. the weight refresh, backward incl. the gradient export, AdamW) at BASELINE
configs[1] shapes.  usage: python tools/bench_train_loop.py [B] [cuda]   (cuda: forecaster parameters resident on the GPU)"""
import os, sys, time, torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..")))
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench
kw = dict(bench.DIFFUSION_KW, lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")
bench.DIFFUSION_KW.clear(); bench.DIFFUSION_KW.update(kw)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
model, F, I = bench.build_model(B, use_graph=False)
if len(sys.argv) > 2 and sys.argv[2] == "cuda":
    F.cuda()
g = torch.Generator().manual_seed(0)
xt = torch.randn(B, 3, 221, 42, generator=g).cuda(); cond = torch.randn(B, 3, 221, 42, generator=g).cuda()
st = torch.rand(B, 2, 221, 42, generator=g).cuda(); t = torch.randint(0, 16, (B,), generator=g).cuda()
opt = torch.optim.AdamW(model.model.parameters(), lr=1e-4)
model.train()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(12):
    t0 = sync()
    opt.zero_grad()
    out = model.p_losses(xt, cond, t, static_condition=st)
    t1 = sync()
    out["loss"].backward()
    t2 = sync()
    opt.step()
    t3 = sync()
    print(f"it {it}: p_losses (incl. weight re-upload) {1e3*(t1-t0):.0f} ms, backward (incl. gradient export) {1e3*(t2-t1):.0f} ms, optimizer {1e3*(t3-t2):.0f} ms")
