"""Host-side mirror of the reference's OISST / synthetic backbone `src/models/unet.py:112-315` (Unet).

Same constructor keywords and the same parameter names as the reference's state_dict; `forward(x, time, condition)`
hands device pointers to the HIP engine (dyf_net_forward, arch = DYF_ARCH_UNET_RESNET).  The torch.nn modules are
parameter containers only -- no torch operator runs in forward().
"""
from contextlib import contextmanager
from typing import Optional, Sequence

import torch
from torch import Tensor, nn

from . import _lib as L
from .engine import (EngineLoss, HipEngine, collect_train_results, default_dtype_for, mark_weights_modified, resnet_net_config,
                     sync_train_weights, sync_weights, upload_weights)
from .unet_simple import _AttrDict

HEADS, DIM_HEAD = 4, 32


def _resnet_block(cin: int, cout: int, time_dim: Optional[int], groups: int, double: bool = True) -> nn.Module:
    """Names of ResnetBlock (unet.py:79-98): mlp.1, block{1,2}.{proj,norm}, residual_conv (block2 = Identity without double_conv_layer)."""
    blk = nn.Module()
    blk.mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, cout * 2)) if time_dim is not None else None
    for i, ci in ((1, cin), (2, cout)) if double else ((1, cin),):
        b = nn.Module()
        b.proj = nn.Conv2d(ci, cout, 3, padding=1)
        b.norm = nn.GroupNorm(groups, cout)
        setattr(blk, f"block{i}", b)
    blk.residual_conv = nn.Conv2d(cin, cout, 1) if cin != cout else nn.Identity()
    return blk


class _Gain(nn.Module):
    def __init__(self, dim):
        super().__init__()
        self.g = nn.Parameter(torch.ones(1, dim, 1, 1))


def _attention(dim: int, linear: bool) -> nn.Module:
    """Names of Residual(PreNorm(dim, fn=[Linear]Attention, norm=LayerNorm)): fn.norm.g, fn.fn.to_qkv[.1], fn.fn.to_out."""
    inner = nn.Module()
    qkv = nn.Conv2d(dim, HEADS * DIM_HEAD * 3, 1, bias=False)
    inner.to_qkv = nn.Sequential(nn.Identity(), qkv) if linear else qkv
    inner.to_out = nn.Conv2d(HEADS * DIM_HEAD, dim, 1)
    pre = nn.Module()
    pre.fn = inner
    pre.norm = _Gain(dim)
    res = nn.Module()
    res.fn = pre
    return res


class Unet(nn.Module):
    default_engine_dtype = "fp16"  # engine.default_dtype_for: the ResNet-UNet is held to 1e-2 per field in fp16 (bf16: 4e-2 .. 5e-2)

    def __init__(self, dim, init_dim=None, dim_mults=(1, 2, 4, 8), num_conditions: int = 0, resnet_block_groups=8,
                 with_time_emb: bool = False, block_dropout: float = 0.0, block_dropout1: float = 0.0,
                 attn_dropout: float = 0.0, input_dropout: float = 0.0, double_conv_layer: bool = True,
                 learned_variance=False, learned_sinusoidal_cond=False, learned_sinusoidal_dim=16,
                 outer_sample_mode: str = None, upsample_dims: tuple = None, keep_spatial_dims: bool = False,
                 init_kernel_size: int = 7, init_padding: int = 3, init_stride: int = 1,
                 num_input_channels: int = None, num_output_channels: int = None, num_conditional_channels: int = 0,
                 spatial_shape: Sequence[int] = None, loss_function: str = "mean_squared_error", datamodule_config=None,
                 name: str = "", verbose: bool = True):
        super().__init__()
        # (outer_sample_mode / upsample_dims: the reference's own constructor raises AttributeError for them -- unet.py:155 reads
        # self.outer_sample_mode, which is never set -- so there is no behaviour to reproduce)
        # learned_variance: accepted -- in the reference it only feeds `default_out_dim` (unet.py:233), which `out_dim =
        # default(output_channels, ...)` never uses because output_channels = num_output_channels or input_channels is always set:
        # a no-op there, a no-op here.  init_dim != dim: the reference's own forward fails (final_res_block is built for 2 * dim
        # channels, unet.py:236, but receives 2 * init_dim): nothing to reproduce.  init_stride != 1 (a half-resolution output) is
        # the one real option left out.
        unsupported = dict(init_dim=init_dim not in (None, dim),
                           outer_sample_mode=outer_sample_mode is not None, upsample_dims=upsample_dims is not None,
                           init_stride=init_stride != 1)
        bad = [k for k, v in unsupported.items() if v]
        if bad:
            raise NotImplementedError(f"the HIP engine implements the shipped Unet settings only; unsupported: {bad}")
        self.hparams = _AttrDict(dim=dim, dim_mults=tuple(dim_mults), resnet_block_groups=resnet_block_groups,
                                 with_time_emb=with_time_emb, block_dropout=block_dropout, block_dropout1=block_dropout1,
                                 attn_dropout=attn_dropout, input_dropout=input_dropout, init_kernel_size=init_kernel_size,
                                 init_padding=init_padding, num_input_channels=num_input_channels,
                                 num_output_channels=num_output_channels, num_conditional_channels=num_conditional_channels,
                                 spatial_shape=spatial_shape, outer_sample_mode=None, upsample_dims=None,
                                 loss_function=loss_function, keep_spatial_dims=bool(keep_spatial_dims),
                                 double_conv_layer=bool(double_conv_layer), learned_sinusoidal_cond=bool(learned_sinusoidal_cond),
                                 learned_sinusoidal_dim=int(learned_sinusoidal_dim), learned_variance=bool(learned_variance))
        self.num_input_channels = num_input_channels
        self.num_conditional_channels = num_conditional_channels
        cin = num_input_channels + num_conditional_channels
        self.num_output_channels = num_output_channels or cin
        self.spatial_shape = None if spatial_shape is None else tuple(spatial_shape)
        self.time_dim = 2 * dim if with_time_emb else None
        g, td = resnet_block_groups, self.time_dim
        self.init_conv = nn.Conv2d(cin, dim, init_kernel_size, padding=init_padding)
        if with_time_emb and learned_sinusoidal_cond:  # misc.py:35-59: a `weights` parameter (dim / 2 frequencies), dim + 1 features
            assert learned_sinusoidal_dim % 2 == 0
            emb = nn.Module()
            emb.weights = nn.Parameter(torch.randn(learned_sinusoidal_dim // 2))
            self.time_emb_mlp = nn.Sequential(emb, nn.Linear(learned_sinusoidal_dim + 1, td), nn.GELU(), nn.Linear(td, td))
        else:
            self.time_emb_mlp = (nn.Sequential(nn.Identity(), nn.Linear(dim, td), nn.GELU(), nn.Linear(td, td))
                                 if with_time_emb else None)
        dims = [dim] + [dim * m for m in dim_mults]
        in_out = list(zip(dims[:-1], dims[1:]))
        self.downs, self.ups = nn.ModuleList(), nn.ModuleList()
        dbl = bool(double_conv_layer)
        for i, (a, b) in enumerate(in_out):
            last = i == len(in_out) - 1 or keep_spatial_dims
            down = nn.Conv2d(a, b, 3, padding=1) if last else nn.Conv2d(a, b, 4, 2, 1)
            self.downs.append(nn.ModuleList([_resnet_block(a, a, td, g, dbl), _resnet_block(a, a, td, g, dbl), _attention(a, True), down]))
        mid = dims[-1]
        self.mid_block1 = _resnet_block(mid, mid, td, g, dbl)
        self.mid_attn = _attention(mid, False)
        self.mid_block2 = _resnet_block(mid, mid, td, g, dbl)
        for i, (a, b) in enumerate(reversed(in_out)):
            last = i == len(in_out) - 1 or keep_spatial_dims
            up = nn.Conv2d(b, a, 3, padding=1) if last else nn.Sequential(nn.Identity(), nn.Conv2d(b, a, 3, padding=1))
            self.ups.append(nn.ModuleList([_resnet_block(b + a, b, td, g, dbl), _resnet_block(b + a, b, td, g, dbl), _attention(b, True), up]))
        self.final_res_block = _resnet_block(dim * 2, dim, td, g, dbl)
        self.final_conv = nn.Conv2d(dim, self.num_output_channels, 1)
        self.requires_grad_(False)
        self.eval()
        self._engine: Optional[HipEngine] = None
        self._engine_slot = L.NET_FORECASTER
        self._engine_key = None
        self._mc_dropout = False

    # ------------------------------------------------------------------ engine plumbing (same protocol as UNet)
    def engine_net_config(self) -> L.NetConfig:
        hp = self.hparams
        return resnet_net_config(in_channels=self.num_input_channels, cond_channels=self.num_conditional_channels,
                                 out_channels=self.num_output_channels, dim=hp.dim, dim_mults=hp.dim_mults,
                                 with_time_emb=hp.with_time_emb, block_dropout=hp.block_dropout,
                                 block_dropout1=hp.block_dropout1, attn_dropout=hp.attn_dropout,
                                 input_dropout=hp.input_dropout, groups=hp.resnet_block_groups,
                                 init_kernel_size=hp.init_kernel_size, init_padding=hp.init_padding,
                                 keep_spatial_dims=hp.keep_spatial_dims, double_conv_layer=hp.double_conv_layer,
                                 learned_sinusoidal_dim=hp.learned_sinusoidal_dim if hp.learned_sinusoidal_cond else 0)

    @property
    def has_dropout(self) -> bool:
        hp = self.hparams
        return hp.block_dropout > 0 or hp.block_dropout1 > 0 or hp.attn_dropout > 0 or hp.input_dropout > 0

    def attach_engine(self, engine: HipEngine, slot: int):
        self._engine, self._engine_slot, self._engine_key = engine, slot, "attached"
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
            # row_groups=1: this engine serves net_forward / the training step, never dyf_sample -- the default row groups would
            # cost a workspace and a packed weight copy each for nothing
            self._engine = HipEngine(cfg, cfg, hw[0], hw[1], max_batch=nb, use_graph=False, dtype=default_dtype_for(self),
                                     row_groups=1, train_precision=getattr(self, "train_precision", None))
            self._engine_slot, self._engine_key = L.NET_FORECASTER, key
            upload_weights(self, self._engine, self._engine_slot)
        return self._engine

    # ------------------------------------------------------------------ reference API
    def forward(self, x: Tensor, time: Tensor = None, condition: Tensor = None, return_time_emb: bool = False) -> Tensor:
        if self.num_conditional_channels > 0:
            if condition is None:
                raise ValueError("condition must be given when num_conditional_channels > 0")
        else:
            assert condition is None, "condition is not None but num_conditional_channels is 0"
        eng = self._own_engine(x.shape[0], x.shape[-2:])
        sync_weights(self, eng, self._engine_slot)  # parameters modified in place since the last upload
        mode = 1 if (self._mc_dropout and self.has_dropout) else 0
        return eng.net_forward(self._engine_slot, x, time if self.hparams.with_time_emb else None, condition,
                               dropout_mode=mode)

    def get_loss(self, inputs: Tensor, targets: Tensor, condition: Tensor = None, metadata=None, predictions_mask=None,
                 return_predictions: bool = False, **kwargs):
        """`BaseModel.get_loss` (_base_model.py:108-138): predict, then the network's criterion.  In eval mode a plain forward; with
        the module in TRAIN mode the engine's recorded fp32 forward (Dropout active; csrc/train_resnet.inc) -- the returned
        scalar's `.backward()` runs dyf_train_backward and accumulates into `param.grad`."""
        kind = getattr(self.hparams, "loss_function", "mean_squared_error")
        if not self.training:
            predictions = self(inputs, condition=condition, **kwargs)
            p = predictions if predictions_mask is None else predictions[predictions_mask]
            loss = predictions.new_tensor(self._engine.criterion(p.contiguous(), targets, kind))
            return (loss, predictions) if return_predictions else loss
        if self.num_conditional_channels > 0 and condition is None:
            raise ValueError("condition must be given when num_conditional_channels > 0")
        eng = self._own_engine(inputs.shape[0], inputs.shape[-2:])
        sync_train_weights(self, eng, self._engine_slot)
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

    def predict_forward(self, inputs: Tensor, metadata=None, **kwargs):
        return self(inputs, **kwargs)

    @contextmanager
    def inference_dropout_scope(self, condition: bool, context=None):
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
