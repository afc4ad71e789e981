"""Host-side mirror of the reference's spring-mesh backbone `src/models/simple_conv_net.py:58-131` (SimpleConvNet).

Same constructor keywords and parameter/buffer names as the reference (`convs.{i}.conv / norm / time_mlp.1`, `head`,
`time_emb_mlp`), same `forward(inputs, time=None, condition=None)`; `forward` hands device pointers to the HIP engine
(`dyf_net_forward`, arch DYF_ARCH_SIMPLE_CONV_NET).  The torch.nn modules are parameter containers only.
"""
from typing import Optional, Sequence

from torch import nn

from . import _lib as L
from .unet_simple import UNet, _AttrDict


def _conv_block(cin: int, cout: int, k: int, time_dim: Optional[int], dropout: float) -> nn.Module:
    """Parameter container named like ConvBlock (simple_conv_net.py:12-55): conv, norm, time_mlp.1."""
    blk = nn.Module()
    blk.conv = nn.Conv2d(cin, cout, k, padding=(k - 1) // 2)
    blk.norm = nn.BatchNorm2d(cout)
    blk.time_mlp = nn.Sequential(nn.SiLU(), nn.Linear(time_dim, 2 * cout)) if time_dim is not None else None
    blk.dropout = nn.Dropout(dropout)
    return blk


class SimpleConvNet(UNet):
    """Engine plumbing (attach_engine / load_state_dict / forward / MC-dropout scope) is shared with `UNet`."""

    def __init__(self, dim: int, with_time_emb: bool = False, net_normalization: str = "batch_norm",
                 kernel_sizes: Sequence[int] = (7, 3, 3), keep_spatial_shape: bool = True, residual: bool = True,
                 dropout: float = 0.0, num_input_channels: int = None, num_output_channels: int = None,
                 num_conditional_channels: int = 0, spatial_shape: Sequence[int] = None,
                 loss_function: str = "mean_squared_error", datamodule_config=None, name: str = "", verbose: bool = True):
        nn.Module.__init__(self)
        if net_normalization != "batch_norm" or not keep_spatial_shape or not residual:
            raise NotImplementedError("the HIP engine implements the shipped SimpleConvNet configuration: batch_norm, "
                                      "keep_spatial_shape=True, residual=True (model/cnn_simple.yaml)")
        if not 1 <= len(kernel_sizes) <= 6 or any(k % 2 == 0 for k in kernel_sizes):
            raise ValueError("kernel_sizes: 1..6 odd sizes")
        self.hparams = _AttrDict(dim=dim, with_time_emb=with_time_emb, net_normalization=net_normalization,
                                 kernel_sizes=tuple(int(k) for k in kernel_sizes), keep_spatial_shape=True, residual=True,
                                 dropout=dropout, input_dropout=0.0, upsample_dims=None,
                                 num_input_channels=num_input_channels, num_output_channels=num_output_channels,
                                 num_conditional_channels=num_conditional_channels, spatial_shape=spatial_shape,
                                 loss_function=loss_function, name=name)
        self.name, self.verbose = name, verbose
        self.num_input_channels = num_input_channels
        self.num_output_channels = num_output_channels
        self.num_conditional_channels = num_conditional_channels
        self.spatial_shape = None if spatial_shape is None else tuple(spatial_shape)
        cin = num_input_channels + num_conditional_channels
        self.time_dim = 2 * dim if with_time_emb else None
        self.time_emb_mlp = (nn.Sequential(nn.Identity(), nn.Linear(dim, self.time_dim), nn.GELU(),
                                           nn.Linear(self.time_dim, self.time_dim)) if with_time_emb else None)
        self.convs = nn.ModuleList([_conv_block(cin if i == 0 else dim, dim, k, self.time_dim, dropout)
                                    for i, k in enumerate(self.hparams.kernel_sizes)])
        self.head = nn.Conv2d(dim, num_output_channels, kernel_size=1, padding=0)
        self.requires_grad_(False)
        self.eval()
        self._engine = None
        self._engine_slot = L.NET_FORECASTER
        self._engine_key = None
        self._weights_version = 0
        self._mc_dropout = False

    def engine_net_config(self) -> L.NetConfig:
        hp = self.hparams
        cfg = L.NetConfig()
        cfg.arch = L.ARCH_SIMPLE_CONV_NET
        cfg.in_channels, cfg.cond_channels, cfg.out_channels = (self.num_input_channels, self.num_conditional_channels,
                                                                self.num_output_channels)
        cfg.dim, cfg.with_time_emb, cfg.upsample_h, cfg.upsample_w = hp.dim, int(bool(hp.with_time_emb)), 0, 0
        cfg.dropout, cfg.input_dropout = float(hp.dropout), 0.0
        cfg.n_mults = len(hp.kernel_sizes)  # kernel_sizes travel in n_mults / dim_mults (include/dyffusion_hip.h)
        for i, k in enumerate(hp.kernel_sizes):
            cfg.dim_mults[i] = int(k)
        return cfg
