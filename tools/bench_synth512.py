"""BASELINE configs[4] shapes: synthetic 512x512x4ch grid, ResNet-UNet dim 64 mults (1,2,4) (bottleneck attention over
128^2 = 16 384 tokens): time one interpolator forward.  usage: python tools/bench_synth512.py [NB]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
import dyffusion_amd as D  # noqa: E402
from bench import random_state  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 1
I = D.Unet(dim=64, dim_mults=(1, 2, 4), with_time_emb=True, num_input_channels=8, num_output_channels=4,
           block_dropout=0.1, attn_dropout=0.1)
sd = random_state(I, 1)
for k in sd:
    if k.endswith(".norm.g"):
        sd[k] = torch.ones_like(sd[k])
I.load_state_dict(sd)
x = torch.randn(nb, 8, 512, 512).cuda()
t = torch.full((nb,), 3.0).cuda()
with I.inference_dropout_scope(True):
    y = I(x, time=t)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    y = I(x, time=t)
    torch.cuda.synchronize()
dt = time.perf_counter() - t0
fl = I._engine.net_flops(0)
print(f"512^2 forward NB={nb}: {dt * 1e3:.1f} ms, {nb * fl / dt / 1e12:.1f} TFLOP/s ({fl / 1e9:.1f} GF/sample), finite={bool(torch.isfinite(y).all())}")
