"""Host-side mirror of the reference's Navier-Stokes backbone `src/models/unet_simple.py:85-197` (UNet).

Same constructor keywords, same parameter/buffer names (so a Lightning checkpoint's `state_dict` loads unchanged)
and the same `forward(inputs, time=None, condition=None)` signature -- but `forward` does not run any torch
operator: it hands device pointers to the HIP engine (dyf_net_forward).  The torch.nn modules below are parameter
containers only.
"""
from contextlib import contextmanager
from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from . import _lib as L
from .engine import (EngineLoss, default_dtype_for, mark_weights_modified, HipEngine, collect_train_results, net_config, sync_train_weights, sync_weights,
                     upload_weights)


class _AttrDict(dict):
    __getattr__ = dict.get

    def __setattr__(self, k, v):
        self[k] = v


def _block_params(cin: int, cout: int, kernel: int, decoder: bool, time_dim: Optional[int], groupnorm: bool,
                  dropout: float) -> nn.Module:
    """Parameter container named like UNetBlock (unet_simple.py:13-65): ops.{0|1} conv, ops.{1|2} norm, time_mlp.1."""
    blk = nn.Module()
    conv = nn.Conv2d(cin, cout, kernel, bias=True)
    norm = nn.GroupNorm(8, cout) if groupnorm else nn.BatchNorm2d(cout)
    blk.ops = nn.Sequential(nn.Identity(), conv, norm) if decoder else nn.Sequential(conv, norm)
    blk.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, 2 * cout)) if time_dim is not None else None
    blk.dropout = nn.Dropout(dropout)
    return blk


class UNet(nn.Module):
    def __init__(self, dim: int, with_time_emb: bool = False, outer_sample_mode: str = "bilinear",
                 upsample_dims: Optional[Sequence[int]] = (256, 256), dropout: float = 0.0, input_dropout: float = 0.0,
                 num_input_channels: int = None, num_output_channels: int = None, num_conditional_channels: int = 0,
                 spatial_shape: Sequence[int] = None, loss_function: str = "mean_squared_error",
                 datamodule_config=None, name: str = "", verbose: bool = True):
        super().__init__()
        if outer_sample_mode not in ("bilinear", "nearest"):
            raise NotImplementedError("the HIP engine implements outer_sample_mode 'bilinear' and 'nearest'")
        self.hparams = _AttrDict(dim=dim, with_time_emb=with_time_emb, outer_sample_mode=outer_sample_mode,
                                 upsample_dims=None if upsample_dims is None else tuple(upsample_dims), dropout=dropout,
                                 input_dropout=input_dropout, num_input_channels=num_input_channels,
                                 num_output_channels=num_output_channels,
                                 num_conditional_channels=num_conditional_channels, spatial_shape=spatial_shape,
                                 loss_function=loss_function, name=name)
        self.name, self.verbose = name, verbose
        self.num_input_channels = num_input_channels
        self.num_output_channels = num_output_channels
        self.num_conditional_channels = num_conditional_channels
        self.spatial_shape = None if spatial_shape is None else tuple(spatial_shape)
        self.outer_sample_mode = outer_sample_mode
        cin = num_input_channels + num_conditional_channels
        self.time_dim = 2 * dim if with_time_emb else None
        # parameter containers, names as in the reference state_dict
        self.time_emb_mlp = (nn.Sequential(nn.Identity(), nn.Linear(dim, self.time_dim), nn.GELU(),
                                           nn.Linear(self.time_dim, self.time_dim)) if with_time_emb else None)
        self.init_conv = nn.Conv2d(cin, dim, 1)
        self.dropout_input = nn.Dropout(input_dropout)
        d = dim
        enc = [(d, 2 * d, 4), (2 * d, 2 * d, 4), (2 * d, 4 * d, 4), (4 * d, 8 * d, 4), (8 * d, 8 * d, 2), (8 * d, 8 * d, 2)]
        dec = [(8 * d, 8 * d, 1), (16 * d, 8 * d, 1), (16 * d, 4 * d, 3), (8 * d, 2 * d, 3), (4 * d, 2 * d, 3), (4 * d, d, 3)]
        self.input_ops = nn.ModuleList([_block_params(a, b, k, False, self.time_dim, i == 5, dropout)
                                        for i, (a, b, k) in enumerate(enc)])
        self.output_ops = nn.ModuleList([_block_params(a, b, k, True, self.time_dim, False, dropout) for a, b, k in dec])
        self.readout = nn.Sequential(nn.ConvTranspose2d(d, num_output_channels, 4, stride=2, padding=1))
        for m in self.modules():  # unet_simple.py:156-162
            if isinstance(m, (nn.Conv2d, nn.ConvTranspose2d)):
                m.weight.data.normal_(0.0, 0.02)
            elif isinstance(m, nn.BatchNorm2d):
                m.weight.data.normal_(1.0, 0.02)
                m.bias.data.fill_(0)
        self.requires_grad_(False)
        self.eval()
        self._engine: Optional[HipEngine] = None
        self._engine_slot = L.NET_FORECASTER
        self._engine_key = None
        self._weights_version = 0
        self._mc_dropout = False

    # ------------------------------------------------------------------ engine plumbing
    def engine_net_config(self) -> L.NetConfig:
        hp = self.hparams
        return net_config(in_channels=self.num_input_channels, cond_channels=self.num_conditional_channels,
                          out_channels=self.num_output_channels, dim=hp.dim, with_time_emb=hp.with_time_emb,
                          upsample_dims=hp.upsample_dims, dropout=hp.dropout, input_dropout=hp.input_dropout,
                          outer_sample_mode=hp.outer_sample_mode)

    @property
    def has_dropout(self) -> bool:
        return self.hparams.dropout > 0 or self.hparams.input_dropout > 0

    def attach_engine(self, engine: HipEngine, slot: int):
        """Used by DYffusion: both networks of a pair live in one engine."""
        self._engine, self._engine_slot = engine, slot
        self._engine_key = "attached"
        upload_weights(self, engine, slot)

    def _apply(self, fn, *args, **kwargs):  # .cuda() / .to() / .float() replace buffer objects: drop state_version's cache
        self.__dict__.pop("_version_tensors", None)
        return super()._apply(fn, *args, **kwargs)

    def mark_weights_modified(self):
        """Call after editing weights in a way `Tensor._version` does not record (e.g. `p.data.copy_(ema)`): the next forward /
        sample / training step re-uploads them."""
        mark_weights_modified(self)

    def load_state_dict(self, state_dict, strict: bool = True, **kw):
        res = super().load_state_dict(state_dict, strict=strict, **kw)
        if self._engine is not None:
            upload_weights(self, self._engine, self._engine_slot)
        return res

    def _own_engine(self, nb: int, hw) -> HipEngine:
        key = (tuple(hw), nb)
        if self._engine is None or (self._engine_key != "attached" and
                                    (self._engine_key[0] != key[0] or self._engine_key[1] < nb)):
            cfg = self.engine_net_config()
            self._engine = HipEngine(cfg, cfg, hw[0], hw[1], max_batch=nb, use_graph=False, dtype=default_dtype_for(self),
                                     train_precision=getattr(self, "train_precision", None))
            self._engine_slot = L.NET_FORECASTER
            self._engine_key = key
            upload_weights(self, self._engine, self._engine_slot)
        return self._engine

    # ------------------------------------------------------------------ reference API
    def forward(self, inputs: Tensor, time: Tensor = None, condition: Tensor = None, return_time_emb: bool = False,
                **kwargs) -> Tensor:
        if self.num_conditional_channels > 0:
            if condition is None:
                raise ValueError("condition must be given when num_conditional_channels > 0")
        else:
            assert condition is None
        eng = self._own_engine(inputs.shape[0], inputs.shape[-2:])
        sync_weights(self, eng, self._engine_slot)  # parameters modified in place since the last upload (optimizer, EMA swap)
        mode = 1 if (self._mc_dropout and self.has_dropout) else 0
        return eng.net_forward(self._engine_slot, inputs, time if self.hparams.with_time_emb else None, condition,
                               dropout_mode=mode)

    def predict_forward(self, inputs: Tensor, metadata=None, **kwargs):
        return self(inputs, **kwargs)

    def get_loss(self, inputs: Tensor, targets: Tensor, condition: Tensor = None, metadata=None,
                 predictions_mask: Optional[Tensor] = None, return_predictions: bool = False, **kwargs):
        """`BaseModel.get_loss` (_base_model.py:108-138): predict, then `criterion(predictions, targets)` with the network's
        `loss_function`.  In eval mode a plain forward; with the module in TRAIN mode the engine's recorded fp32 forward
        (batch-statistics BatchNorm with running-statistics update, Dropout active, csrc/train.hip) -- the returned scalar's
        `.backward()` runs dyf_train_backward and accumulates into `param.grad`."""
        kind = self.hparams.loss_function
        if not self.training:
            predictions = self(inputs, condition=condition, **kwargs)
            eng = self._engine
            p = predictions if predictions_mask is None else predictions[predictions_mask]
            loss = predictions.new_tensor(eng.criterion(p.contiguous(), targets, kind))
            return (loss, predictions) if return_predictions else loss
        if self.num_conditional_channels > 0 and condition is None:
            raise ValueError("condition must be given when num_conditional_channels > 0")
        eng = self._own_engine(inputs.shape[0], inputs.shape[-2:])
        sync_train_weights(self, eng, self._engine_slot)  # the fp32 training copy only
        time = kwargs.get("time") if self.hparams.with_time_emb else None
        pred = eng.train_forward(self._engine_slot, 0, inputs, None if time is None else time.float(), condition,
                                 batch_stats=True, dropout=self.has_dropout)
        # _base_model.py:132-135: criterion(predictions[predictions_mask], targets) -- the mean runs over the selected elements
        mask = None
        if predictions_mask is not None:  # boolean index over the leading dimensions, as torch's predictions[predictions_mask]
            mask = predictions_mask.to(device=pred.device, dtype=torch.bool)
            mask = mask.reshape(tuple(mask.shape) + (1,) * (pred.dim() - mask.dim())).expand_as(pred)
        value = eng.criterion(pred, targets, kind) if mask is None else \
            eng.criterion(pred[mask].contiguous(), targets.reshape(-1), kind)  # predictions[mask] and targets hold the same elements
        eng.train_step_id += 1
        self._train_state = dict(eng=eng, pred=pred, targets=targets.float().contiguous(), kind=kind, step_id=eng.train_step_id,
                                 mask=mask)
        if not hasattr(self, "_grad_anchor"):
            self._grad_anchor = torch.zeros((), requires_grad=True)
        loss = EngineLoss.apply(self._grad_anchor, self, float(value))
        return (loss, pred) if return_predictions else loss

    def _train_backward(self, upstream: float):
        st = self._train_state
        eng = st["eng"]
        if st.get("mask") is None:
            d = eng.criterion_grad(st["pred"], st["targets"], st["kind"], upstream)
        else:  # gradient of the masked mean: zero outside the mask
            d = torch.zeros_like(st["pred"])
            d[st["mask"]] = eng.criterion_grad(st["pred"][st["mask"]].contiguous(), st["targets"].reshape(-1), st["kind"], upstream)
        eng.train_backward(0, d, want_dinputs=False, param_grads=True)
        collect_train_results(self, eng, self._engine_slot, 1)

    @contextmanager
    def inference_dropout_scope(self, condition: bool, context=None):
        """_base_model.py:148-161: only the Dropout layers are switched; BatchNorm stays in eval mode."""
        assert isinstance(condition, bool), f"Condition must be a boolean, got {condition}"
        prev = self._mc_dropout
        if condition:
            self._mc_dropout = True
        try:
            yield None
        finally:
            self._mc_dropout = prev

    def enable_inference_dropout(self):
        self._mc_dropout = True

    def disable_inference_dropout(self):
        self._mc_dropout = False
