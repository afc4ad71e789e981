"""Time the training step of the forecaster objective at BASELINE configs[1] shapes (NS 221x42, unet_simple dim 64 @256^2):
`DYffusion.p_losses` in training mode + `loss.backward()` = 2 interpolator + 2 forecaster recorded forwards and their
backward passes (fp32 VALU kernels, csrc/train.hip).  usage: python tools/bench_train_step.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
kw = dict(bench.DIFFUSION_KW, lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")
bench.DIFFUSION_KW.clear()
bench.DIFFUSION_KW.update(kw)
model, F, I = bench.build_model(B, use_graph=False)
model.train()
g = torch.Generator().manual_seed(0)
xt_last = torch.randn(B, bench.C, bench.H, bench.W, generator=g).cuda()
cond = torch.randn(B, bench.C, bench.H, bench.W, generator=g).cuda()
static = torch.rand(B, bench.CS, bench.H, bench.W, generator=g).cuda()
t = torch.randint(0, bench.HORIZON, (B,), generator=g).cuda()
best = None
for it in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = model.p_losses(xt_last, cond, t, static_condition=static)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    out["loss"].backward()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if it > 0 and (best is None or t2 - t0 < best[2] - best[0]):
        best = (t0, t1, t2)
    for p_ in model.model.parameters():
        p_.grad = None
t0, t1, t2 = best
out = model.p_losses(xt_last, cond, t, static_condition=static)
out["loss"].backward()
gn = float(torch.cat([p.grad.reshape(-1) for p in model.model.parameters()]).norm())
fl = model._engine.net_flops(0)
print(f"training step B={B}: forward {1e3 * (t1 - t0):.0f} ms, backward {1e3 * (t2 - t1):.0f} ms, loss {float(out['loss']):.4f}, "
      f"|grad| {gn:.4f}; ~{B * fl * 12 / (t2 - t0) / 1e12:.2f} TFLOP/s (4 forwards + 2 x 4 backward-equivalents)")
