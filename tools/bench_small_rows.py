"""Few-rows regime of the NS workload (the rows one GPU gets when an ensemble is sharded 8 ways): per-layer conv timings
(dyf_time_conv_layer, back-to-back launches of ONE layer on real activations) and whole rollouts at 1..20 rows, for the kernel-form
switches in the environment (DYF_HALO_SPLITK_FILL, DYF_UP_BORDER_SPLIT_ROWS, ...: read per launch / at graph capture).
usage: python tools/bench_small_rows.py [layers|rollouts|both] [tag]   -> one JSON line per measurement"""
import json
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import bench  # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else "both"
tag = sys.argv[2] if len(sys.argv) > 2 else ""
dev = torch.device("cuda", 0)
MAXB = int(os.environ.get("DYF_SMALL_MAXB", "80"))
model, F, I = bench.build_model(MAXB, use_graph=True)
model.seed(2)
g = torch.Generator().manual_seed(1)
names = [f"enc{i}" for i in range(6)] + [f"dec{i}" for i in range(6)]
env = {k: v for k, v in os.environ.items() if k.startswith("DYF_") and k not in ("DYF_SMALL_MAXB",)}

if what in ("layers", "both"):
    x0 = torch.randn(40, 3, 221, 42, generator=g).to(dev)
    st = torch.rand(40, 2, 221, 42, generator=g).to(dev)
    model.sample(x0, static_condition=st)  # populate the workspace with realistic activations
    torch.cuda.synchronize()
    eng = model._engine
    for nb in (1, 2, 4, 7, 10, 14, 20, 40, 80):
        row = {"tag": tag, "kind": "layers", "rows": nb, "env": env, "us": {}}
        for layer in range(12):
            ms, fl, by = eng.time_conv_layer(1, layer, nb, 30)
            row["us"][names[layer]] = round(ms * 1e3, 2)
        row["sum_us"] = round(sum(row["us"].values()), 1)
        print(json.dumps(row), flush=True)

if what in ("rollouts", "both"):
    curve = bench.ns_batch_curve(model, dev, nbs=(1, 2, 4, 7, 10, 14, 20, 25))
    print(json.dumps({"tag": tag, "kind": "rollouts", "env": env, "fields_per_s": curve,
                      "ms_per_rollout": {k: round(1e3 * k * bench.HORIZON / v, 3) for k, v in curve.items()}}), flush=True)
