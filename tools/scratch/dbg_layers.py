"""Per-layer error of one full-size forward: engine vs fp32 oracle and vs the bf16 arithmetic model."""
import json, os, sys
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from oracle import nets, sampler
from tests.gpu_common import DEV, build_dyffusion, mirror_from_params, seeded_pair
from tests.helpers import jload, rel_rms
meta = jload("fullsize_checksums.json"); mk = meta["model"]
PF, PI = seeded_pair(64, 3, 2, seeds=(101, 102))
g = torch.Generator().manual_seed(1)
x0 = torch.randn(1, 3, 221, 42, generator=g); c = torch.rand(1, 2, 221, 42, generator=g)
net = mirror_from_params(PI, mk, 6, 2, 3)
t = torch.tensor([5.0])
xin = torch.cat([x0, x0.flip(0) * 0.5 + 0.1], 1)
os.environ["DYF_SPARSE_DEC5"] = "1"
y = net(xin.to(DEV), time=t.to(DEV), condition=c.to(DEV))
eng = net._engine
taps32, taps16 = {}, {}
with torch.no_grad():
    y32 = nets.unet_simple_forward(PI, mk, xin, t, c, taps=taps32)
    y16 = nets.unet_simple_forward_bf16_model(PI, mk, xin, t, c, taps=taps16)
names = [f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]
for li, nm in enumerate(names):
    a = eng.read_block_output(0, li, 1).cpu()
    ok = torch.isfinite(a)
    e32 = rel_rms(a[ok], taps32[nm][ok]); e16 = rel_rms(a[ok], taps16[nm][ok]); m = rel_rms(taps16[nm], taps32[nm])
    print(f"{nm}: engine-vs-fp32 {e32:.2e}  engine-vs-bf16model {e16:.2e}  bf16model-vs-fp32 {m:.2e}  computed {float(ok.float().mean()):.2f}")
print("output: engine-vs-fp32 %.2e engine-vs-bf16model %.2e bf16model-vs-fp32 %.2e" % (rel_rms(y.cpu(), y32), rel_rms(y.cpu(), y16), rel_rms(y16, y32)))
# rollout
hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
          sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=2)
out = m.sample(x0.to(DEV), static_condition=c.to(DEV))
with torch.no_grad():
    o32 = sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward(PF, mk, x, t, cond), lambda x, t, cond: nets.unet_simple_forward(PI, mk, x, t, cond), x0, c, hp)
    o16 = sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward_bf16_model(PF, mk, x, t, cond), lambda x, t, cond: nets.unet_simple_forward_bf16_model(PI, mk, x, t, cond), x0, c, hp)
for k in ("t1_preds", "t8_preds", "t16_preds"):
    print(k, "engine-vs-fp32 %.2e engine-vs-bf16model %.2e bf16model-vs-fp32 %.2e" % (rel_rms(out[k].cpu(), o32[k]), rel_rms(out[k].cpu(), o16[k]), rel_rms(o16[k], o32[k])))
