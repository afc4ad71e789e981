"""Time the ResNet-UNet training step (BASELINE configs[2] shapes: OISST 60x60, unet.Unet dim 64 mults (1,2,4)): `DYffusion.p_losses`
in training mode + `loss.backward()` = 2 interpolator + 2 forecaster recorded forwards and their backward passes (csrc/train_resnet.inc).
usage: python tools/bench_train_step_resnet.py [B]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D  # noqa: E402
from bench import _resnet_state  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2, attn_dropout=0.6, **kw)
F.load_state_dict(_resnet_state(F, 0, 0.5))
I.load_state_dict(_resnet_state(I, 1, 0.5))
m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                additional_interpolation_steps=25, lambda_reconstruction=0.5, lambda_reconstruction2=0.5, loss_function="l1", max_batch=B)
m.train()
g = torch.Generator().manual_seed(0)
xt_last, cond = torch.randn(B, 1, 60, 60, generator=g).cuda(), torch.randn(B, 1, 60, 60, generator=g).cuda()
t = torch.randint(0, m.num_timesteps, (B,), generator=g).cuda()


def step():
    out = m.p_losses(xt_last, cond, t, static_condition=None)
    out["loss"].backward()
    return float(out["loss"])


step()
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
fl = 3 * 2 * (m._engine.net_flops(0) + m._engine.net_flops(1)) * B  # forward + 2x backward, 2 + 2 forwards
print(f"ResNet-UNet training step B={B}: {dt * 1e3:.1f} ms, loss {loss:.4f}, ~{fl / dt / 1e12:.1f} TFLOP/s (conv/matmul FLOPs, fwd + bwd)")
