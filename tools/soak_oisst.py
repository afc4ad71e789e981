"""Soak of the OISST rollout (fused-GroupNorm convs with the in-launch statistics exchange, row groups) over many batch sizes: every
rollout must be finite, repeatable bit for bit, and the engine must never have left the fused form (dyf_gn_fuse_state downgrades = 0).
usage: python tools/soak_oisst.py [repeats]"""
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()
import dyffusion_amd as D  # noqa: E402
from bench import random_state  # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2
kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2, attn_dropout=0.6, **kw)
for net, seed in ((F, 0), (I, 1)):
    sd = random_state(net, seed)
    for k in sd:
        if k.endswith(".norm.g"):
            sd[k] = torch.ones_like(sd[k])
        elif sd[k].dim() == 4:
            sd[k] = sd[k] * 0.5
    net.load_state_dict(sd)
bad = 0
t_all = time.perf_counter()
for max_batch in (300, 75, 38):
    m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                    additional_interpolation_steps=25, refine_intermediate_predictions=False, max_batch=max_batch)
    g = torch.Generator().manual_seed(3)
    x_all = torch.randn(max_batch, 1, 60, 60, generator=g).cuda()
    sizes = sorted({n for n in (1, 2, 3, 5, 8, 13, 19, 24, 27, 38, 50, 64, 71, 72, 75, 100, 128, 150, 199, 256, 300) if n <= max_batch})
    for nb in sizes:
        outs = []
        for r in range(reps):
            m.seed(11)
            out = m.sample(x_all[:nb])
            outs.append(torch.stack([out[k] for k in sorted(out)]).cpu())
        fin = bool(torch.isfinite(outs[0]).all())
        same = all(torch.equal(outs[0], o) for o in outs[1:])
        live, down = m._engine.gn_fuse_state()
        flag = "" if (fin and same and down == 0) else "   <-- PROBLEM"
        bad += bool(flag)
        print(f"max_batch {max_batch:3d} rows {nb:3d}: finite={fin} repeatable={same} fused_live={live} downgrades={down} groups={m._engine.row_groups}{flag}", flush=True)
    m._engine.close()
print(f"soak done in {time.perf_counter() - t_all:.1f} s, problems: {bad}")
sys.exit(1 if bad else 0)
