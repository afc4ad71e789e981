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
    """Stage-1 interpolator experiment (`src/experiment_types/interpolation.py:12-167`): evaluation -- the network is
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


    # --------------------------------- training (stage 1)
    def get_loss(self, batch: Dict[str, Any]) -> Tensor:
        """interpolation.py:149-167: one uniformly drawn interpolation time per batch item, the frame at that time as the
        target, `model.get_loss(inputs, targets, time=t, **rest of the batch)`.  With the module in train mode the returned
        scalar's `.backward()` runs the engine's backward pass (UNet.get_loss)."""
        dynamics = batch["dynamics"]
        inputs = self.get_inputs_from_dynamics(dynamics)
        b = dynamics.shape[0]
        possible_times = torch.tensor(self.horizon_range, device=dynamics.device, dtype=torch.long)
        t = possible_times[torch.randint(len(possible_times), (b,), device=dynamics.device, dtype=torch.long)]
        targets = dynamics[torch.arange(b, device=dynamics.device), self.window + t - 1]
        return self.model.get_loss(inputs=inputs, targets=targets, time=t, **{k: v for k, v in batch.items() if k != "dynamics"})

    def training_step(self, batch: Dict[str, Any], batch_idx: int = 0):  # _base_experiment.py:440-470 without the logging
        loss = self.get_loss(batch)
        return {"loss": loss}


class MultiHorizonForecastingDYffusion(nn.Module):
    """Caller contract of the hot path: `src/experiment_types/forecasting_multi_horizon.py` (AbstractMultiHorizonForecasting-
    Experiment :20-229, MultiHorizonForecastingDYffusion :398-424) over `_base_experiment.py` (`predict` :315-379, ensemble
    tiling / reshape :503-567, `evaluation_step` :484-492, `predict_step` :700-708), reduced to what drives `DYffusion.sample`.
    `datamodule` is any object with the two methods `evaluation_step` touches: `boundary_conditions(preds, targets, metadata,
    time)` and `get_boundary_condition_kwargs(batch, batch_idx, split)` (e.g. built on `PhysicalSystemsBoundaryConditions`)."""

    def __init__(self, model: DYffusion, num_predictions: int = 1, window: int = 1, horizon: Optional[int] = None,
                 autoregressive_steps: int = 0, prediction_horizon: Optional[int] = None, datamodule=None):
        super().__init__()
        assert autoregressive_steps >= 0, f"Autoregressive steps must be >= 0, but is {autoregressive_steps}"
        if autoregressive_steps > 0:
            assert prediction_horizon is None, "Cannot use ``prediction_horizon`` with autoregressive_steps > 0"
        self.model = model
        self.hparams = _AttrDict(num_predictions=num_predictions, autoregressive_steps=autoregressive_steps)
        self.window = window
        self.horizon = horizon or model.hparams.timesteps
        self._prediction_horizon = prediction_horizon
        self._datamodule = datamodule
        self._predict_step_outputs = []

    # ---- forecasting_multi_horizon.py:45-100
    @property
    def true_horizon(self) -> int:
        return self.horizon

    @property
    def horizon_range(self):
        return list(range(1, self.horizon + 1))

    @property
    def prediction_timesteps(self):
        return self.horizon_range

    @property
    def prediction_horizon(self) -> int:
        if self._prediction_horizon:
            return self._prediction_horizon
        return self.horizon * (self.hparams.autoregressive_steps + 1)

    @property
    def num_autoregressive_steps(self) -> int:
        n = self.hparams.autoregressive_steps
        if n == 0 and self._prediction_horizon is not None:
            n = max(1, -(-self._prediction_horizon // self.true_horizon)) - 1
        return n

    @property
    def datamodule(self):
        return self._datamodule

    # _base_experiment.py:503-538 -- "N B ... -> (N B) ...": ensemble-major rows (row = n*B + b)
    def get_ensemble_inputs(self, inputs_raw: Optional[Tensor], num_predictions: Optional[int] = None) -> Optional[Tensor]:
        n = num_predictions or self.hparams.num_predictions
        if inputs_raw is None or n <= 1:
            return inputs_raw
        return inputs_raw.unsqueeze(0).expand(n, *inputs_raw.shape).reshape(n * inputs_raw.shape[0], *inputs_raw.shape[1:])

    # --------------------------------- training (stage 2)
    def get_loss(self, batch: Dict[str, Any]):
        """forecasting_multi_horizon.py:412-420: the first `window` frames (stacked on channels) are the inputs, the last frame the
        target, every other batch entry (the static condition) goes through; `DYffusion.get_loss` draws the diffusion steps and
        evaluates `p_losses`.  Returns the loss dict (`"loss"` is the scalar to call `.backward()` on in training mode).
        (In the training split the reference does not tile inputs into an ensemble; nor does this.)"""
        dynamics = batch["dynamics"]
        b = dynamics.shape[0]
        inputs = dynamics[:, : self.window].reshape(b, -1, *dynamics.shape[-2:])  # "b window c lat lon -> b (window c) lat lon"
        extra = {k: v for k, v in batch.items() if k not in ("dynamics", "metadata")}
        return self.model.get_loss(inputs=inputs, targets=dynamics[:, -1], **extra)

    def training_step(self, batch: Dict[str, Any], batch_idx: int = 0):  # _base_experiment.py:440-470 without the logging
        out = self.get_loss(batch)
        return out if isinstance(out, dict) else {"loss": out}

    # _base_experiment.py:315-379,540-567
    def predict(self, inputs: Tensor, num_predictions: Optional[int] = None, reshape_ensemble_dim: bool = True,
                **kwargs) -> Dict[str, Tensor]:
        """`inputs` are already ensemble-tiled ((N*B, ...)); returns `t{i}_preds` reshaped to (N, B, ...).  As in the
        reference, `num_predictions` only travels to the sampler; the reshape uses the module's own ensemble size."""
        n = self.hparams.num_predictions
        results = self.model.predict_forward(inputs, num_predictions=num_predictions or n, **kwargs)
        if torch.is_tensor(results):
            results = {"preds": results}
        if reshape_ensemble_dim:
            first = next(v for k, v in results.items() if "preds" in k)
            if first.shape[0] > 1 and n > 1 and first.shape[0] % n == 0:
                for k, v in list(results.items()):
                    if "targets" not in k and "true" not in k:
                        b = v.shape[0]
                        assert b % n == 0, f"key={k}: b % #ens_mems = {b} % {n} != 0 ...Did you forget to create the input ensemble?"
                        results[k] = v.reshape(n, max(1, b // n), *v.shape[1:])
        return results

    # _base_experiment.py:484-492
    @torch.no_grad()
    def evaluation_step(self, batch: Dict[str, Any], batch_idx: int = 0, split: str = "predict", **kwargs) -> Dict[str, Any]:
        if self.datamodule is not None and "boundary_conditions" not in kwargs:
            kwargs["boundary_conditions"] = self.datamodule.boundary_conditions
            kwargs.update(self.datamodule.get_boundary_condition_kwargs(batch, batch_idx, split))
        return self._evaluation_step(batch, batch_idx, split, **kwargs)

    # forecasting_multi_horizon.py:114-229 (prediction branch; metrics belong to the training harness): autoregressive
    # outer loop.  Every outer iteration is ONE engine rollout (all h fields at once -- the reference's per-horizon
    # `get_preds_at_t_for_batch` pops them from a cache, :319-329); the last `window` predicted fields of every ensemble row
    # become the next iteration's initial condition, so the (N*B) rows never leave the GPU between iterations.
    @torch.no_grad()
    def _evaluation_step(self, batch: Dict[str, Any], batch_idx: int = 0, split: str = "predict", dataloader_idx: int = None,
                         return_outputs: bool = True, boundary_conditions=None, t0=0.0, dt=1.0,
                         prediction_horizon: Optional[int] = None, as_numpy: bool = False) -> Dict[str, Any]:
        """Returns {"t{k}_targets": (B, C, H, W), "t{k}_preds": (N, B, C, H, W)} for k = 1..prediction_horizon (torch tensors on
        the GPU, or numpy arrays like the reference with `as_numpy`).  `boundary_conditions(preds=, targets=, metadata=, time=)`
        is applied to every predicted field before it is returned / fed back, exactly where the reference applies it."""
        dynamics = batch["dynamics"].clone()
        b = dynamics.shape[0]
        h = self.true_horizon
        prediction_horizon = prediction_horizon or self.prediction_horizon
        if split == "val" and dataloader_idx in (0, None):
            n_outer = 1  # forecasting_multi_horizon.py:134-136: "Simple evaluation without autoregressive steps", no length check
        else:
            assert split in ("val", "test", "predict")
            n_outer = max(1, -(-prediction_horizon // h))  # = num_autoregressive_steps + 1 (forecasting_multi_horizon.py:71-76,141)
            if dynamics.shape[1] < prediction_horizon:
                raise ValueError(f"Prediction horizon {prediction_horizon} is larger than {dynamics.shape}[1]")
        n = self.hparams.num_predictions
        cond = self.get_ensemble_inputs(batch.get("condition", None), n)
        out: Dict[str, Any] = {}
        conv = (lambda v: None if v is None else v.detach().cpu().numpy()) if as_numpy else (lambda v: v)
        inputs = None
        total_t = t0
        for ar_step in range(n_outer):
            if inputs is None:  # transform_inputs: "b window c lat lon -> b (window c) lat lon", then the ensemble tiling
                inputs = self.get_ensemble_inputs(batch["dynamics"][:, : self.window].reshape(b, -1, *dynamics.shape[-2:]), n)
            preds = self.predict(inputs, condition=cond, num_predictions=None if ar_step == 0 else 1)
            ar_window = []  # predictions at the last `window` horizon steps: the next outer iteration's initial condition
            for t_step in self.prediction_timesteps:
                total_h = ar_step * h + t_step
                if total_h > prediction_horizon:
                    break
                total_t = total_t + dt
                p = preds[f"t{t_step}_preds"]
                targets = dynamics[:, self.window + total_h - 1]
                if boundary_conditions is not None:
                    p = boundary_conditions(preds=p, targets=targets, metadata=batch.get("metadata", None), time=total_t)
                if return_outputs:
                    out[f"t{total_h}_targets"] = conv(targets)
                    out[f"t{total_h}_preds"] = conv(p)
                if t_step in self.horizon_range[-self.window:]:  # forecasting_multi_horizon.py:194-197
                    ar_window.append(p.reshape(-1, *p.shape[-3:]))
            if ar_step < n_outer - 1:  # "(N B) window c h w -> (N B) (window c) h w": already ensemble-tiled (:218-220)
                inputs = torch.cat(ar_window, dim=1).contiguous()
                batch["dynamics"] *= 1e6  # as the reference: "become completely dummy after first multistep prediction"
        return out

    # _base_experiment.py:700-708
    # --------------------------------- validation / test epochs: ensemble metrics on the device
    def ensemble_logging_infix(self, split: str) -> str:  # _base_experiment.py:609-615 (no logging_infix, no input noise)
        return f"{self.hparams.num_predictions}ens_mems/"

    def validation_step(self, batch: Dict[str, Any], batch_idx: int = 0, dataloader_idx: int = None, **kwargs):
        """_base_experiment.py:603-607: the evaluation step's fields stay on the GPU (no `torch_to_numpy`)."""
        results = self.evaluation_step(batch, batch_idx, split="val", dataloader_idx=dataloader_idx, **kwargs)
        self.__dict__.setdefault("_validation_step_outputs", []).append(results)
        return results

    def test_step(self, batch: Dict[str, Any], batch_idx: int = 0, dataloader_idx: int = None, **kwargs):
        results = self.evaluation_step(batch, batch_idx, split="test", **kwargs)
        self.__dict__.setdefault("_test_step_outputs", []).append(results)
        return results

    def _eval_ensemble_predictions(self, outputs, split: str) -> Dict[str, float]:
        """_base_experiment.py:569-640: concatenate the steps' outputs (predictions (N, B, ...) along B, targets along their batch
        axis) and evaluate CRPS / spread-skill ratio / ensemble-mean MSE per horizon and on average -- with
        `dyffusion_amd.metrics` (ensemble_metrics_kernel), not after a `.cpu().numpy()` round trip.  Returns what the reference
        logs: `{split}/{N}ens_mems/t{k}/{crps|ssr|mse}` and `.../avg/...`."""
        from .metrics import eval_ensemble_predictions
        n = self.hparams.num_predictions
        if n <= 1 or not outputs:  # use_ensemble_predictions(split), :497-498
            return {}
        results = {}
        for key, first in outputs[0].items():
            if not torch.is_tensor(first):
                continue
            axis = 1 if (first.shape[0] == n and first.dim() >= 2 and "targets" not in key and "true" not in key) else 0
            results[key] = torch.cat([o[key] for o in outputs], dim=axis)
        return eval_ensemble_predictions(results, self.model._engine, split=split, infix=self.ensemble_logging_infix(split))

    def on_validation_epoch_end(self) -> Dict[str, float]:
        outs, self.__dict__["_validation_step_outputs"] = self.__dict__.get("_validation_step_outputs", []), []
        return self._eval_ensemble_predictions(outs, split="val")

    def on_test_epoch_end(self, calc_ensemble_metrics: bool = True) -> Dict[str, float]:
        outs, self.__dict__["_test_step_outputs"] = self.__dict__.get("_test_step_outputs", []), []
        return self._eval_ensemble_predictions(outs, split="test") if calc_ensemble_metrics else {}

    def predict_step(self, batch: Dict[str, Any], batch_idx: int = 0, dataloader_idx: int = None, **kwargs) -> None:
        results = self.evaluation_step(batch, batch_idx, split="predict", as_numpy=True, **kwargs)
        self._predict_step_outputs.append(results)

    def on_predict_epoch_end(self) -> Dict[str, Any]:
        """Concatenate the per-batch outputs along the batch dimension (predictions: dim 1 of (N, B, ...))."""
        import numpy as np

        outs, self._predict_step_outputs = self._predict_step_outputs, []
        if not outs:
            return {}
        return {k: np.concatenate([o[k] for o in outs], axis=1 if ("preds" in k and outs[0][k].ndim == 5) else 0)
                for k in outs[0]}
