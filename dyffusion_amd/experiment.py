"""Caller contract of the hot path: the `predict` / `predict_step` surface of
`src/experiment_types/_base_experiment.py:315-379,503-567,700-708` and the per-batch entry of
`forecasting_multi_horizon.py:282-342`, reduced to what drives `DYffusion.sample`."""
from typing import Any, Dict, Optional

import torch
from torch import Tensor, nn

from .dyffusion import DYffusion
from .unet_simple import _AttrDict


class InterpolatorHandle:
    """Duck type of the reference's InterpolationExperiment as DYffusion uses it (dyffusion.py:461-478):
    `.model`, `.window`, `.true_horizon`."""

    def __init__(self, model, horizon: int, window: int = 1):
        self.model, self.true_horizon, self.window = model, horizon, window

    def inference_dropout_scope(self, condition: bool, context=None):
        return self.model.inference_dropout_scope(condition, context)


class InterpolationExperiment(nn.Module):
    """Stage-1 interpolator experiment, evaluation path (`src/experiment_types/interpolation.py:12-141`): the network is
    asked for every intermediate step t = 1..h-1 given the first `window` frames and the last frame.  Also serves as the
    `interpolator` argument of `DYffusion` (`.model`, `.window`, `.true_horizon`, `inference_dropout_scope`)."""

    def __init__(self, model, horizon: int, window: int = 1, num_predictions: int = 1,
                 stack_window_to_channel_dim: bool = True, enable_inference_dropout: bool = False):
        super().__init__()
        assert horizon >= 2, "horizon must be >=2 for interpolation experiments"
        if not stack_window_to_channel_dim:
            raise NotImplementedError("stack_window_to_channel_dim=False (no shipped config uses it)")
        self.model = model
        self.horizon = self.true_horizon = horizon
        self.window = window
        self.hparams = _AttrDict(num_predictions=num_predictions, stack_window_to_channel_dim=True,
                                 enable_inference_dropout=enable_inference_dropout)

    @property
    def horizon_range(self):  # interpolation.py:22-27
        return list(range(1, self.horizon))

    def inference_dropout_scope(self, condition: bool, context=None):
        return self.model.inference_dropout_scope(condition, context)

    def get_ensemble_inputs(self, inputs_raw: Optional[Tensor], num_predictions: Optional[int] = None) -> Optional[Tensor]:
        n = num_predictions or self.hparams.num_predictions
        if inputs_raw is None or n <= 1:
            return inputs_raw
        return inputs_raw.unsqueeze(0).expand(n, *inputs_raw.shape).reshape(n * inputs_raw.shape[0], *inputs_raw.shape[1:])

    def get_inputs_from_dynamics(self, dynamics: Tensor) -> Tensor:  # interpolation.py:128-141
        assert dynamics.shape[1] == self.window + self.horizon, "dynamics must have shape (b, t, c, h, w)"
        b = dynamics.shape[0]
        past = dynamics[:, : self.window].reshape(b, -1, *dynamics.shape[-2:])  # "b window c lat lon -> b (window c) lat lon"
        return torch.cat([past, dynamics[:, -1]], dim=1)

    def get_evaluation_inputs(self, dynamics: Tensor) -> Tensor:
        return self.get_ensemble_inputs(self.get_inputs_from_dynamics(dynamics))

    # _base_experiment.py:315-379 with the interpolator as the model
    @torch.no_grad()
    def predict(self, inputs: Tensor, time: Tensor, num_predictions: Optional[int] = None,
                reshape_ensemble_dim: bool = True, **kwargs) -> Dict[str, Tensor]:
        n = num_predictions or self.hparams.num_predictions
        with self.model.inference_dropout_scope(condition=bool(self.hparams.enable_inference_dropout)):
            preds = self.model.predict_forward(inputs, time=time, **kwargs)
        if reshape_ensemble_dim and n > 1 and preds.shape[0] > 1:
            assert preds.shape[0] % n == 0
            preds = preds.reshape(n, preds.shape[0] // n, *preds.shape[1:])
        return {"preds": preds}

    # interpolation.py:69-127: one network call per intermediate step; returns the predictions, the targets and the MSE
    # of the (ensemble-mean) prediction per step and on average, computed on the GPU
    @torch.no_grad()
    def evaluation_step(self, batch: Dict[str, Any], split: str = "val") -> Dict[str, Any]:
        dynamics = batch["dynamics"]
        inputs = self.get_evaluation_inputs(dynamics)
        extra = {k: self.get_ensemble_inputs(v) for k, v in batch.items() if k != "dynamics"}
        out: Dict[str, Any] = {}
        mses = []
        for t_step in self.horizon_range:
            targets = dynamics[:, self.window + t_step - 1]
            time = torch.full((inputs.shape[0],), t_step, device=inputs.device, dtype=torch.long)
            preds = self.predict(inputs, time=time, **extra)["preds"]
            out[f"t{t_step}_preds"], out[f"t{t_step}_targets"] = preds, targets
            mean = preds.mean(dim=0) if preds.dim() == targets.dim() + 1 else preds
            mse = float(((mean - targets) ** 2).mean())
            out[f"{split}/t{t_step}/ipol/mse"] = mse
            mses.append(mse)
        out[f"{split}/{self.horizon}h_avg/ipol/mse"] = float(sum(mses) / len(mses))
        return out


class MultiHorizonForecastingDYffusion(nn.Module):
    def __init__(self, model: DYffusion, num_predictions: int = 1, window: int = 1, horizon: Optional[int] = None):
        super().__init__()
        self.model = model
        self.hparams = _AttrDict(num_predictions=num_predictions)
        self.window = window
        self.horizon = horizon or model.hparams.timesteps

    # _base_experiment.py:503-538 -- "N B ... -> (N B) ...": ensemble-major rows (row = n*B + b)
    def get_ensemble_inputs(self, inputs_raw: Optional[Tensor], num_predictions: Optional[int] = None) -> Optional[Tensor]:
        n = num_predictions or self.hparams.num_predictions
        if inputs_raw is None or n <= 1:
            return inputs_raw
        return inputs_raw.unsqueeze(0).expand(n, *inputs_raw.shape).reshape(n * inputs_raw.shape[0], *inputs_raw.shape[1:])

    # _base_experiment.py:315-379,540-567
    def predict(self, inputs: Tensor, num_predictions: Optional[int] = None, reshape_ensemble_dim: bool = True,
                **kwargs) -> Dict[str, Tensor]:
        n = num_predictions or self.hparams.num_predictions
        results = self.model.predict_forward(inputs, num_predictions=n, **kwargs)
        if torch.is_tensor(results):
            results = {"preds": results}
        if reshape_ensemble_dim:
            for k, v in list(results.items()):
                b = v.shape[0]
                if "preds" in k and b > 1 and n > 1:
                    assert b % n == 0, f"key={k}: b % #ens_mems = {b} % {n} != 0 ...Did you forget to create the input ensemble?"
                    results[k] = v.reshape(n, max(1, b // n), *v.shape[1:])
        return results

    # forecasting_multi_horizon.py:337-342 + :282-332 (first prediction step: tile, sample, cache all horizons)
    @torch.no_grad()
    def predict_step(self, batch: Dict[str, Any], batch_idx: int = 0, dataloader_idx: int = None) -> Dict[str, Any]:
        dynamics = batch["dynamics"]
        b = dynamics.shape[0]
        inputs = dynamics[:, : self.window].reshape(b, -1, *dynamics.shape[-2:])  # "b window c h w -> b (window c) h w"
        cond = batch.get("condition", None)
        n = self.hparams.num_predictions
        preds = self.predict(self.get_ensemble_inputs(inputs, n), condition=self.get_ensemble_inputs(cond, n),
                             num_predictions=n)
        return {k: v.detach().cpu().numpy() for k, v in preds.items()}

    # forecasting_multi_horizon.py:114-229 (prediction branch): autoregressive outer loop.  Every outer iteration is
    # one engine rollout (h fields); the last `window` predicted fields of every ensemble row become the next
    # iteration's initial condition; rows stay independent, so the (N*B) rows never leave the GPU between iterations.
    @torch.no_grad()
    def evaluation_step(self, batch: Dict[str, Any], prediction_horizon: Optional[int] = None, boundary_conditions=None,
                        t0: float = 0.0, dt: float = 1.0, return_targets: bool = True) -> Dict[str, Tensor]:
        """Returns {"t{k}_preds": (N, B, C, H, W)} (and "t{k}_targets" when the batch holds them) for
        k = 1..prediction_horizon.  `boundary_conditions(preds=, targets=, metadata=, time=)` is applied to every
        predicted field before it is returned / fed back, exactly where the reference applies it."""
        dynamics = batch["dynamics"]
        b = dynamics.shape[0]
        h = self.horizon
        prediction_horizon = prediction_horizon or h
        if self.window != 1:
            raise NotImplementedError("autoregressive evaluation is implemented for window == 1 (the shipped configs)")
        n = self.hparams.num_predictions
        n_outer = -(-prediction_horizon // h)
        cond = self.get_ensemble_inputs(batch.get("condition", None), n)
        inputs = self.get_ensemble_inputs(dynamics[:, : self.window].reshape(b, -1, *dynamics.shape[-2:]), n)
        out: Dict[str, Tensor] = {}
        total_t = t0
        for ar_step in range(n_outer):
            preds = self.model.predict_forward(inputs, condition=cond, num_predictions=n)
            last = None
            for t_step in range(1, h + 1):
                total_h = ar_step * h + t_step
                if total_h > prediction_horizon:
                    break
                total_t += dt
                p = preds[f"t{t_step}_preds"]
                p = p.reshape(n, b, *p.shape[1:]) if n > 1 else p
                tgt_idx = self.window + total_h - 1
                targets = dynamics[:, tgt_idx] if tgt_idx < dynamics.shape[1] else None
                if boundary_conditions is not None:
                    p = boundary_conditions(preds=p, targets=targets, metadata=batch.get("metadata", None), time=total_t)
                out[f"t{total_h}_preds"] = p
                if return_targets and targets is not None:
                    out[f"t{total_h}_targets"] = targets
                last = p
            if ar_step < n_outer - 1:  # "N B c h w -> (N B) c h w": next initial condition, already ensemble-tiled
                inputs = last.reshape(-1, *last.shape[-3:]).contiguous()
        return out
