"""One training convolution through the test seam (dyf_train_conv_check), a few times -- for rocprofv3 (kernel time, HBM counters) of the
16-bit-operand tile + halo kernels (csrc/train_halo16.hip).  usage: python tools/bench_train_conv.py [kind n h w cin cout k s p] [reps]
kind 0 forward, 1 data gradient, 2 weight gradient; default: the NS decoder's last conv at 16 rows (forward, 16 x 256^2 x 128 -> 64)."""
import os
import sys

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
os.environ.setdefault("DYF_TRAIN_OPERANDS", "bf16")
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D  # noqa: E402
from dyffusion_amd.engine import net_config  # noqa: E402

args = [int(a) for a in sys.argv[1:]]
case = tuple(args[:9]) if len(args) >= 9 else (0, 16, 256, 256, 128, 64, 3, 1, 1)
reps = args[9] if len(args) >= 10 else 3
cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False)
for _ in range(reps):
    err, _, took = eng.train_conv_check(case[0], *case[1:], seed=1)
n, h, w, cin, cout = case[1:6]
px = n * h * w
print(f"kind {case[0]} {case[1:]}: rel max err {err:.2e}, took {took}; fp32 tensors: x {px * cin * 4 / 1e6:.1f} MB, z {px * cout * 4 / 1e6:.1f} MB")
