"""What does bf16 arithmetic cost on the full-size NS rollout?  ("What's weak" #2.)

Runs BASELINE config 2 (NS 221x42, dim 64 @256^2, h=16, cold + refine, dropout off, NB=1; fixture G6 seeds) on the CPU
with the oracle in three arithmetic models and prints the rel-RMS of t1 / t8 / t16 against the fp32 oracle:

  autocast   torch.autocast("cpu", dtype=bfloat16) around both networks: what the reference itself computes under
             `trainer.precision=bf16` (SURVEY 8c quotes 3.5e-3 - 4.5e-3 for it).
  storage    fp32 convolutions on bf16-rounded operands, every layer output rounded to bf16 once (after the fused
             norm / FiLM / activation epilogue), sampler state fp32: the rounding points of the HIP engine.
  weights    bf16-rounded weights only (activations fp32): isolates the systematic part of the error.

Test infrastructure (imports oracle/); never imported by the product.  Output is committed under profiles/.
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from oracle import init as oinit  # noqa: E402
from oracle import nets, sampler  # noqa: E402
from tests.helpers import jload, load_npz, rel_rms  # noqa: E402


def main():
    torch.set_num_threads(os.cpu_count())
    meta, fields = jload("fullsize_checksums.json"), load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), meta["seeds"]["forecaster"])
    PI = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 8, 3), meta["seeds"]["interpolator"])
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)

    def fp32(P):
        return lambda x, t, cond: nets.unet_simple_forward(P, mk, x, t, cond)

    def autocast(P):
        def f(x, t, cond):
            with torch.autocast("cpu", dtype=torch.bfloat16):
                return nets.unet_simple_forward(P, mk, x, t, cond).float()
        return f

    def storage(P, **kw):
        return lambda x, t, cond: nets.unet_simple_forward_bf16_model(P, mk, x, t, cond, **kw)

    modes = {"fp32": fp32, "autocast": autocast, "storage": storage,
             "weights": lambda P: storage(P, round_act=False), "acts": lambda P: storage(P, round_w=False)}
    want = sys.argv[1:] or list(modes)
    res = {}
    with torch.no_grad():
        ref = None
        for name in ["fp32"] + [m for m in want if m != "fp32"]:
            t0 = time.time()
            out = sampler.sample_loop(modes[name](PF), modes[name](PI), x0, c, hp)
            if name == "fp32":
                ref = out
                res["fp32_vs_reference_fixture"] = {k: rel_rms(out[f"{k}_preds"], fields[k]) for k in ("t1", "t8", "t16")}
            else:
                res[name] = {k: rel_rms(out[f"{k}_preds"], ref[f"{k}_preds"]) for k in ("t1", "t4", "t8", "t12", "t16")}
                # one forward, same inputs: the per-forward error of this arithmetic model
                tt = torch.full((1,), 5.0)
                a = modes[name](PI)(torch.cat([x0, ref["t16_preds"]], 1), tt, c)
                b = fp32(PI)(torch.cat([x0, ref["t16_preds"]], 1), tt, c)
                res[name]["one_forward"] = rel_rms(a, b)
            print(name, json.dumps(res.get(name, res.get("fp32_vs_reference_fixture"))), f"{time.time() - t0:.0f}s", flush=True)
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
