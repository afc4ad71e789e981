import os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
import dyffusion_amd as D
from tests.test_gpu_fp16 import _seeded_unet
DEV = "cuda:0"
for hw, mults, drop in (((64, 64), (1, 2), False), ((64, 64), (1, 2), True), ((128, 128), (1, 2, 4), False), ((128, 128), (1, 2, 4), True)):
    F_, _ = _seeded_unet(64, mults, 4, 4, seed=91)
    I_, _ = _seeded_unet(64, mults, 8, 4, seed=92, block_dropout=0.1 if drop else 0.0, attn_dropout=0.1 if drop else 0.0)
    hp = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=False, enable_interpolator_dropout=drop)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(2, 4, *hw, generator=g).to(DEV)
    def run(use_graph, dtype="fp16"):
        m = D.DYffusion(F_, D.InterpolatorHandle(I_, 4), max_batch=2, dtype=dtype, use_graph=use_graph, **hp)
        m.seed(5)
        a = {k: v.clone() for k, v in m.sample(x0).items()}
        return a
    for dtype in ("fp16", "bf16"):
        e1, e2, g1, g2 = run(False, dtype), run(False, dtype), run(True, dtype), run(True, dtype)
        def d(a, b): return max(float((a[k] - b[k]).abs().max()) for k in a)
        print(hw, mults, "dropout", drop, dtype, "eager-eager", d(e1, e2), "graph-graph", d(g1, g2), "eager-graph", d(e1, g1), flush=True)
