"""Time the flash-attention core (dyf_op_attention: 4 heads x 32 dims) at the bottleneck of BASELINE configs[4]: 16 384 tokens
(128 x 128), NB rows, without and with dropout on the probabilities.  usage: python tools/bench_attention.py [NB] [tokens] [p]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D  # noqa: E402
from dyffusion_amd.engine import net_config  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 4
n = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
p = float(sys.argv[3]) if len(sys.argv) > 3 else 0.1
eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=max(1, nb), use_graph=False)
g = torch.Generator().manual_seed(0)
qkv = torch.randn(nb, n, 384, generator=g).to(eng.torch_dtype).cuda()
fl = nb * 4 * 2 * 2 * n * n * 32
for pd in (0.0, p):
    for _ in range(10):  # the device idles in a low-power state: ten launches before the timed ones (measured: one warm-up launch
        y = eng.op_attention(qkv, pd)  # and ten timed ones read 0.84 ms where the steady state is 0.76 ms)
    torch.cuda.synchronize()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    ev0.record()
    for _ in range(reps):
        y = eng.op_attention(qkv, pd)
    ev1.record()
    torch.cuda.synchronize()
    ms = ev0.elapsed_time(ev1) / reps
    print(f"attention core NB={nb}, {n} tokens, 4 heads x 32, dropout p={pd}: {ms:.3f} ms, {fl / ms / 1e9:.1f} TFLOP/s = "
          f"{fl / ms / 1e9 / 2500:.3f} of the dense MFMA peak, finite={bool(torch.isfinite(y.float()).all())}")
