"""Lightning-checkpoint weight import: what `reload_model_from_config_and_ckpt` /
`get_checkpoint_from_path_or_wandb` (`src/interface.py:115-172`) do to get trained weights into the networks, reduced to
the key handling -- the engine's networks keep the reference's parameter names, so a checkpoint maps by prefix only:

  forecasting run   (MultiHorizonForecastingDYffusion):  state_dict["model.model.*"]              -> forecaster
                                                         state_dict["model.interpolator.model.*"] -> frozen interpolator copy
  interpolation run (InterpolationExperiment):           state_dict["model.*"]                    -> interpolator

The reference drops `model.interpolator*` on reload and takes the interpolator from its own run's checkpoint
(`interface.py:157-159`, `dyffusion.py:461-478`); `split_lightning_state_dict` returns both so either source can be used.
"""
import os
from typing import Any, Dict, Mapping, Optional, Union

import torch

_FORECASTER = "model.model."
_INTERPOLATOR_COPY = "model.interpolator.model."
_PLAIN = "model."


def _state_dict_of(ckpt: Union[str, os.PathLike, Mapping[str, Any]]) -> Mapping[str, torch.Tensor]:
    if isinstance(ckpt, (str, os.PathLike)):
        # weights_only: a Lightning .ckpt also pickles callbacks/hparams that need pytorch_lightning to unpickle
        ckpt = torch.load(ckpt, map_location="cpu", weights_only=True)
    if "state_dict" in ckpt and isinstance(ckpt["state_dict"], Mapping):
        return ckpt["state_dict"]
    return ckpt


def split_lightning_state_dict(ckpt: Union[str, os.PathLike, Mapping[str, Any]]) -> Dict[str, Dict[str, torch.Tensor]]:
    """-> {"forecaster": {...}, "interpolator": {...}} with the Lightning prefixes stripped; a role whose prefix does not
    occur is absent.  A plain network state dict (no prefixes) is returned under "model"."""
    sd = _state_dict_of(ckpt)
    out: Dict[str, Dict[str, torch.Tensor]] = {}
    is_forecasting = any(k.startswith(_FORECASTER) for k in sd)
    for k, v in sd.items():
        if k.startswith(_INTERPOLATOR_COPY):
            out.setdefault("interpolator", {})[k[len(_INTERPOLATOR_COPY):]] = v
        elif k.startswith(_FORECASTER):
            out.setdefault("forecaster", {})[k[len(_FORECASTER):]] = v
        elif k.startswith("model.interpolator"):
            continue  # non-network buffers of the wrapped InterpolationExperiment
        elif k.startswith(_PLAIN) and not is_forecasting:
            out.setdefault("interpolator", {})[k[len(_PLAIN):]] = v
        elif not k.startswith(_PLAIN):
            out.setdefault("model", {})[k] = v
    return out


def load_networks_from_checkpoints(forecaster, interpolator, forecaster_ckpt=None, interpolator_ckpt=None,
                                   strict: bool = True) -> Dict[str, Optional[int]]:
    """Fill the engine networks (dyffusion_amd.UNet / dyffusion_amd.Unet) from a forecasting-run checkpoint and,
    as the reference does, the interpolator from its own run's checkpoint when given (else from the frozen copy inside
    the forecasting checkpoint).  Returns {"epoch", "global_step"} of the forecasting checkpoint when present."""
    meta: Dict[str, Optional[int]] = {"epoch": None, "global_step": None}
    if forecaster_ckpt is not None:
        raw = forecaster_ckpt
        if isinstance(raw, (str, os.PathLike)):
            raw = torch.load(raw, map_location="cpu", weights_only=True)
        parts = split_lightning_state_dict(raw)
        if "forecaster" not in parts:
            raise KeyError("no 'model.model.*' keys: not a MultiHorizonForecastingDYffusion checkpoint")
        forecaster.load_state_dict(parts["forecaster"], strict=strict)
        if interpolator is not None and interpolator_ckpt is None and "interpolator" in parts:
            interpolator.load_state_dict(parts["interpolator"], strict=strict)
        if isinstance(raw, Mapping):
            meta["epoch"], meta["global_step"] = raw.get("epoch"), raw.get("global_step")
    if interpolator_ckpt is not None:
        parts = split_lightning_state_dict(interpolator_ckpt)
        src = parts.get("interpolator") or parts.get("model")
        if src is None:
            raise KeyError("no 'model.*' keys: not an InterpolationExperiment checkpoint")
        interpolator.load_state_dict(src, strict=strict)
    return meta
