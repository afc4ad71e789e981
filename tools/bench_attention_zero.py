"""diagnosis: the attention core on all-zero / random operands (same instruction stream): a gap says the kernel is power-limited"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D
from dyffusion_amd.engine import net_config
nb, n = 4, 16384
cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=nb, use_graph=False)
g = torch.Generator().manual_seed(0)
r = torch.randn(nb, n, 384, generator=g)
for name, x in (("randn", r), ("zeros", torch.zeros_like(r)), ("randn*0.1", r * 0.1), ("randn", r)):
    qkv = x.to(eng.torch_dtype).cuda()
    for _ in range(3):
        y = eng.op_attention(qkv, 0.0)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        y = eng.op_attention(qkv, 0.0)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name}: {e0.elapsed_time(e1) / 20:.3f} ms")
