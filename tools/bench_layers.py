"""Per-layer conv timing on the GPU through the C ABI (dyf_time_conv_layer): TFLOP/s of every UNetBlock conv of the
NS interpolator at batch NB.  usage: python tools/bench_layers.py [NB] [iters]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 50
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 10
model, F, I = bench.build_model(nb, use_graph=False)
g = torch.Generator().manual_seed(1)
x0 = torch.randn(nb, 3, 221, 42, generator=g).cuda()
st = torch.rand(nb, 2, 221, 42, generator=g).cuda()
model.sample(x0, static_condition=st)  # populate the workspace with realistic activations
torch.cuda.synchronize()
eng = model._engine
names = [f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]
tot_ms = tot_fl = 0.0
for layer, nm in enumerate(names):
    ms, fl, by = eng.time_conv_layer(1, layer, nb, iters)
    tot_ms += ms
    tot_fl += fl
    print(f"{nm}: {ms * 1e3:9.1f} us  {fl / ms / 1e9:8.1f} TFLOP/s  ({fl / 1e9:8.1f} GF, algo {by / 1e6:8.1f} MB -> {by / ms / 1e6:7.1f} GB/s)")
print(f"all convs: {tot_ms:.3f} ms, {tot_fl / tot_ms / 1e9:.1f} TFLOP/s")
