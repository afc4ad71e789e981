"""ORACLE (test infrastructure) -- deterministic parameter construction shared by fixtures, tests and bench.

Full-size networks (10 M parameters each) are too large to commit, so fixtures that need them record only
(seed, checksums); the parameters are rebuilt from `seeded_state` wherever they are needed.  Shapes follow the
reference's state_dict layout: unet_simple.py:86-162 (UNet.__init__), simple_conv_net.py:59-110.
"""
import math
from typing import Dict, Tuple

import torch
from torch import Tensor


def unet_simple_param_shapes(dim: int, in_channels: int, out_channels: int, with_time_emb: bool = True) -> Dict[str, Tuple[int, ...]]:
    """state_dict keys -> shapes of src.models.unet_simple.UNet (in_channels = inputs + conditional)."""
    from .nets import unet_simple_layout

    s: Dict[str, Tuple[int, ...]] = {}
    tdim = 2 * dim
    if with_time_emb:
        s["time_emb_mlp.1.weight"] = (tdim, dim)
        s["time_emb_mlp.1.bias"] = (tdim,)
        s["time_emb_mlp.3.weight"] = (tdim, tdim)
        s["time_emb_mlp.3.bias"] = (tdim,)
    s["init_conv.weight"] = (dim, in_channels, 1, 1)
    s["init_conv.bias"] = (dim,)
    enc, dec = unet_simple_layout(dim)
    for group, blocks, conv_idx in (("input_ops", enc, 0), ("output_ops", dec, 1)):
        for li, (cin, cout, k, _, _, norm, _) in enumerate(blocks):
            pre = f"{group}.{li}"
            if with_time_emb:
                s[f"{pre}.time_mlp.1.weight"] = (2 * cout, tdim)
                s[f"{pre}.time_mlp.1.bias"] = (2 * cout,)
            s[f"{pre}.ops.{conv_idx}.weight"] = (cout, cin, k, k)
            s[f"{pre}.ops.{conv_idx}.bias"] = (cout,)
            n = f"{pre}.ops.{conv_idx + 1}"
            s[f"{n}.weight"] = (cout,)
            s[f"{n}.bias"] = (cout,)
            if norm == "bn":
                s[f"{n}.running_mean"] = (cout,)
                s[f"{n}.running_var"] = (cout,)
                s[f"{n}.num_batches_tracked"] = ()
    s["readout.0.weight"] = (dim, out_channels, 4, 4)
    s["readout.0.bias"] = (out_channels,)
    return s


def simple_conv_net_param_shapes(dim: int, in_channels: int, out_channels: int, kernel_sizes, with_time_emb: bool = True):
    s: Dict[str, Tuple[int, ...]] = {}
    tdim = 2 * dim
    if with_time_emb:
        s["time_emb_mlp.1.weight"] = (tdim, dim)
        s["time_emb_mlp.1.bias"] = (tdim,)
        s["time_emb_mlp.3.weight"] = (tdim, tdim)
        s["time_emb_mlp.3.bias"] = (tdim,)
    for li, k in enumerate(kernel_sizes):
        cin = in_channels if li == 0 else dim
        pre = f"convs.{li}"
        s[f"{pre}.conv.weight"] = (dim, cin, k, k)
        s[f"{pre}.conv.bias"] = (dim,)
        s[f"{pre}.norm.weight"] = (dim,)
        s[f"{pre}.norm.bias"] = (dim,)
        s[f"{pre}.norm.running_mean"] = (dim,)
        s[f"{pre}.norm.running_var"] = (dim,)
        s[f"{pre}.norm.num_batches_tracked"] = ()
        if with_time_emb:
            s[f"{pre}.time_mlp.1.weight"] = (2 * dim, tdim)
            s[f"{pre}.time_mlp.1.bias"] = (2 * dim,)
    s["head.weight"] = (out_channels, dim, 1, 1)
    s["head.bias"] = (out_channels,)
    return s


def seeded_state(shapes: Dict[str, Tuple[int, ...]], seed: int, gain: float = 1.4) -> Dict[str, Tensor]:
    """Deterministic, well-conditioned parameters: activations stay O(1) through the depth of the net so that
    parity checks are meaningful (the reference's own normal(0, 0.02) init makes every activation tiny)."""
    g = torch.Generator().manual_seed(seed)
    out: Dict[str, Tensor] = {}
    for key in sorted(shapes):
        shp = tuple(shapes[key])
        if key.endswith("num_batches_tracked"):
            out[key] = torch.zeros((), dtype=torch.int64)
        elif key.endswith("running_mean"):
            out[key] = 0.1 * torch.randn(shp, generator=g)
        elif key.endswith("running_var"):
            out[key] = 0.5 + torch.rand(shp, generator=g)
        elif len(shp) == 1 and key.endswith(".weight"):
            out[key] = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif len(shp) == 1:
            out[key] = 0.05 * torch.randn(shp, generator=g)
        else:
            if key.startswith("readout"):  # ConvTranspose2d weight is (Cin, Cout, kh, kw); 2x2 taps hit each output
                fan_in = shp[0] * 4
            else:
                fan_in = int(math.prod(shp[1:]))
            w = torch.randn(shp, generator=g) * (gain / math.sqrt(fan_in))
            if "time_mlp" in key:  # keep FiLM modulation moderate
                w = w * 0.5
            out[key] = w
    return out
