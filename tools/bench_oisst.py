"""Timing of the OISST configuration (BASELINE configs[2] shapes): DYffusion with the ResNet-UNet pair, 60x60x1, h=7,
k=25 (T=32, 93 network forwards), data+noise, MC dropout on; NB rows on one GPU.  usage: python tools/bench_oisst.py [NB]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D  # noqa: E402
from bench import random_state  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2,
           attn_dropout=0.6, **kw)
for net, seed in ((F, 0), (I, 1)):
    sd = random_state(net, seed)
    for k in sd:
        if k.endswith(".norm.g"):
            sd[k] = torch.ones_like(sd[k])
        elif sd[k].dim() == 4:
            sd[k] = sd[k] * 0.5  # the T=32 recursion of a random-init pair must stay inside fp16's range (as bench.py)
    net.load_state_dict(sd)
m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                additional_interpolation_steps=25, refine_intermediate_predictions=False, max_batch=nb,
                use_graph=os.environ.get("DYF_NO_GRAPH", "0") != "1", dtype=os.environ.get("DYF_OISST_DTYPE") or None)
x0 = torch.randn(nb, 1, 60, 60).cuda()
m.sample(x0)
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = 3
for _ in range(reps):
    out = m.sample(x0)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
eng = m._engine
nf, ni = eng.forward_counts()
fl = nf * eng.net_flops(0) + ni * eng.net_flops(1)
print(f"OISST NB={nb}: {dt * 1e3:.1f} ms per rollout ({nf}+{ni} forwards) -> {nb * 7 / dt:.1f} fields/s, "
      f"{nb * fl / dt / 1e12:.1f} TFLOP/s algorithmic ({fl / 7 / 1e9:.1f} GF/field), finite={all(torch.isfinite(v).all() for v in out.values())}")
