"""Sample the GPU clock and package power (rocm-smi) while a workload loops: the attention core on random / all-zero operands and
the NS rollout at 80 rows.  usage: python tools/sample_clocks.py > profiles/rXX_clocks_under_load.txt"""
import os
import re
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench  # noqa: E402
import dyffusion_amd as D  # noqa: E402
from dyffusion_amd.engine import net_config  # noqa: E402


def smi():
    out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out)
    pw = re.search(r"Power \(W\): ([0-9.]+)", out)
    return (int(sclk.group(1)) if sclk else -1), (float(pw.group(1)) if pw else -1.0)


def sample_while(name, fn, seconds=4.0):
    stop = threading.Event()
    samples = []

    def sampler():
        while not stop.is_set():
            samples.append(smi())
            time.sleep(0.05)

    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        fn()
        n += 1
        if n % 8 == 0:
            torch.cuda.synchronize()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    stop.set()
    th.join()
    s = samples[len(samples) // 3:]  # drop the ramp
    clk = sorted(c for c, _ in s if c > 0)
    pw = sorted(p for _, p in s if p > 0)
    med = lambda v: v[len(v) // 2] if v else -1
    print(f"{name:<58} {1e3 * dt / n:9.3f} ms per call   sclk median {med(clk)} MHz (min {clk[0] if clk else -1}, max {clk[-1] if clk else -1})   "
          f"power median {med(pw)} W   ({len(s)} samples)", flush=True)


print("# rocm-smi sclk / package power sampled while each workload loops for 4 s (first third of the samples dropped)")
idle = smi()
print(f"idle: sclk {idle[0]} MHz, power {idle[1]} W")
nb, n = 4, 16384
cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=nb, use_graph=False)
g = torch.Generator().manual_seed(0)
r = torch.randn(nb, n, 384, generator=g)
for name, x in (("attention core 16384 tokens, random operands", r), ("attention core 16384 tokens, all-zero operands", torch.zeros_like(r))):
    qkv = x.to(eng.torch_dtype).cuda()
    sample_while(name, lambda: eng.op_attention(qkv, 0.0))
qkv = r.to(eng.torch_dtype).cuda()
sample_while("attention core 16384 tokens, random, dropout 0.1", lambda: eng.op_attention(qkv, 0.1))
m, _, _ = bench.build_model(80)
m.seed(1)
x0 = torch.randn(80, bench.C, bench.H, bench.W, generator=g).cuda()
st = torch.rand(80, bench.CS, bench.H, bench.W, generator=g).cuda()
m.sample(x0, static_condition=st)
sample_while("NS rollout, 80 rows (hipGraph replay)", lambda: m.sample(x0, static_condition=st), 6.0)
z0, zs = torch.zeros_like(x0), torch.zeros_like(st)
sample_while("NS rollout, 80 rows, all-zero inputs", lambda: m.sample(z0, static_condition=zs), 6.0)
m._engine.close()
del m
mo, _, _, _ = bench.oisst_model(300)
xo = torch.randn(300, 1, 60, 60, generator=g).cuda()
mo.sample(xo)
sample_while("OISST rollout, 300 rows, 3 row groups (hipGraph replay)", lambda: mo.sample(xo), 6.0)
mo._engine.close()
del mo
ms = bench.synth512_model(4)
xs = torch.randn(4, 4, 512, 512, generator=g).cuda()
ms.sample(xs)
sample_while("512^2 rollout, 4 rows, fp16 (hipGraph replay)", lambda: ms.sample(xs), 6.0)
