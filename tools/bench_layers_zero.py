"""experiment: per-layer conv timing with (a) normal random weights/activations, (b) conv weights zeroed (activations become constants
per channel: no operand toggling), same instruction stream -- is the halo kernel power-limited?"""
import os, sys, torch
ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench
nb, iters = 80, 30
mode = sys.argv[1]
model, F, I = bench.build_model(nb, use_graph=False)
if mode != "normal":
    for net in (F, I):
        sd = net.state_dict()
        for k, v in sd.items():
            if v.dim() == 4:  # conv weights
                sd[k] = torch.zeros_like(v) if mode == "zero" else torch.full_like(v, 0.01)
        net.load_state_dict(sd)
g = torch.Generator().manual_seed(1)
x0 = torch.randn(nb, 3, 221, 42, generator=g).cuda(); st = torch.rand(nb, 2, 221, 42, generator=g).cuda()
if mode != "normal":
    x0.zero_(); st.zero_()
model.sample(x0, static_condition=st); torch.cuda.synchronize()
eng = model._engine
names = [f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]
out = []
for layer, nm in enumerate(names):
    ms, fl, by = eng.time_conv_layer(1, layer, nb, iters)
    if nm in ("enc0", "enc1", "dec2", "dec3", "dec4", "dec5"):
        out.append(f"{nm}: {ms*1e3:.1f}us {fl/ms/1e9:.0f}TF")
print(mode, " ".join(out))
