"""On-device ensemble metrics: the surface of `src/utilities/evaluation.py` and of
`BaseExperiment._eval_ensemble_predictions` (`_base_experiment.py:617-640`) on GPU tensors, so the forecast stack never
makes the reference's `.cpu().numpy()` round trip.  The reductions run in `ensemble_metrics_kernel` (kernels.hip) through
`dyf_ensemble_metrics`; there is no CPU fallback."""
from collections import defaultdict
from typing import Dict, Optional

import torch
from torch import Tensor

from .engine import HipEngine


def evaluate_ensemble_prediction(predictions: Tensor, targets: Tensor, engine: HipEngine) -> Dict[str, float]:
    """evaluation.py:10-80 with ensemble_dim=0, mean_over_samples=True: predictions (n_members, n_samples, *),
    targets (n_samples, *), both on the engine's GPU -> {"ssr", "crps", "mse"}."""
    if predictions.shape[1] != targets.shape[0]:
        raise AssertionError(f"predictions.shape[1] ({predictions.shape[1]}) != targets.shape[0] ({targets.shape[0]})")
    if not predictions.is_cuda or not targets.is_cuda:
        raise ValueError("dyffusion_amd.metrics works on GPU tensors (the HIP engine computes them)")
    mse, ssr, crps = engine.ensemble_metrics(predictions, targets)
    return {"ssr": ssr, "crps": crps, "mse": mse}


def eval_ensemble_predictions(results: Dict[str, Tensor], engine: HipEngine, split: str = "val",
                              infix: Optional[str] = None) -> Dict[str, float]:
    """_base_experiment.py:617-640: per-horizon metrics `{split}/{infix}t{k}/{m}` for every `t{k}_preds`/`t{k}_targets`
    pair of an evaluation step's output, plus their average over horizons `{split}/{infix}avg/{m}`."""
    infix = "" if infix is None else infix
    out: Dict[str, float] = {}
    acc = defaultdict(list)
    for key in [k for k in results if k.endswith("preds")]:
        prefix = key.split("_")[0] if key != "preds" else ""
        tkey = f"{prefix}_targets" if prefix else "targets"
        if tkey not in results:
            continue
        for m, v in evaluate_ensemble_prediction(results[key], results[tkey], engine).items():
            out[f"{split}/{infix}{prefix}/{m}"] = v
            acc[f"{split}/{infix}avg/{m}"].append(v)
    for k, v in acc.items():
        out[k] = float(sum(v) / len(v))
    return out
