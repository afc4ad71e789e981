import json, sys, os
import torch
sys.path.insert(0, os.path.abspath(os.path.join(os.path.dirname(__file__), "..", "..")))
from tests.gpu_common import DEV, build_dyffusion, seeded_pair
from tests.helpers import load_npz, rel_rms
HP4 = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
           sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
z = load_npz("stats_ens256.npz")
hp_meta = json.loads(str(z["hp"]))
PF, PI = seeded_pair(64, 3, 2, seeds=(101, 102))
x0, c = torch.from_numpy(z["x0"]), torch.from_numpy(z["c"])
for N in (1, 4, 64, 256):
    m = build_dyffusion(PF, PI, hp_meta["model"], 3, 2, HP4, max_batch=N)
    out = m.sample(x0.repeat(N, 1, 1, 1).to(DEV), static_condition=c.repeat(N, 1, 1, 1).to(DEV))
    o = out["t1_preds"].cpu()
    if N == 1:
        base = {k: v.cpu() for k, v in out.items()}
    bad = [(k, int((out[k].cpu() != base[k]).any(dim=(1, 2, 3)).sum())) for k in out]
    print("dropout off N", N, "rows differing from the N=1 result:", bad, flush=True)
# dropout on: per-row means
HP4["enable_interpolator_dropout"] = True
for N in (64, 256):
    m = build_dyffusion(PF, PI, hp_meta["model"], 3, 2, HP4, max_batch=N)
    m.seed(2024)
    out = m.sample(x0.repeat(N, 1, 1, 1).to(DEV), static_condition=c.repeat(N, 1, 1, 1).to(DEV))
    for k in sorted(out):
        v = out[k].double().cpu()
        mean_r, var_r = torch.from_numpy(z[f"mean::{k}"]).double()[0], torch.from_numpy(z[f"var::{k}"]).double()[0]
        d = (v - mean_r).pow(2).mean(dim=(1, 2, 3)).sqrt()
        print("N", N, k, "rms dev from ref mean per row: first8", [round(float(q), 3) for q in d[:8]], "blocks of 32:",
              [round(float(d[i:i + 32].mean()), 3) for i in range(0, N, 32)], "ref spread", float(var_r.mean().sqrt()),
              "ens var ratio", float(v.var(0).mean() / var_r.mean()), "mean diff rms", float((v.mean(0) - mean_r).pow(2).mean().sqrt()))
