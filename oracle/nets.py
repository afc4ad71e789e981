"""ORACLE (test infrastructure, CPU fp32) -- functional restatement of the reference backbones.

Each function takes the reference's own `state_dict` (same key names) as a plain dict `P` of fp32
tensors and computes the forward pass with torch.nn.functional primitives; no nn.Module state.
Reference being restated:
  unet_simple_forward   <- /root/reference/src/models/unet_simple.py:13-82 (UNetBlock), :164-197 (UNet)
  time_embedding        <- /root/reference/src/models/modules/misc.py:20-32, :54-67
  simple_conv_net_forward <- /root/reference/src/models/simple_conv_net.py:12-55, :112-131
  resnet_unet_forward   <- /root/reference/src/models/unet.py:26-109, :266-315 and
                           /root/reference/src/models/modules/attention.py:7-73, net_norm.py:18-26
Parity: pinned against tests/golden/*.npz (outputs of the imported reference) in
tests/test_oracle_nets.py.  Third-party arithmetic (conv/batch-norm/bilinear/softmax) is PyTorch ATen,
as in the reference.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F
from torch import Tensor

LEAKY_SLOPE = 0.2  # unet_simple.py:10

# Every nn.Conv2d of the restated backbones goes through `conv2d` below: F.conv2d, unless a test has swapped in a model of the
# engine's arithmetic for the duration of a `with` block (oracle/losses.py `training_operand_rounding`: the bf16 operand rounding
# of the engine's mixed-precision training convolutions under torch.autograd).
_CONV2D = [F.conv2d]


def conv2d(x: Tensor, w: Tensor, bias: Optional[Tensor] = None, stride=1, padding=0) -> Tensor:
    return _CONV2D[0](x, w, bias, stride=stride, padding=padding)


# ----------------------------------------------------------------------------- dropout sources
class DropoutOff:
    """Dropout layers are identity (eval mode)."""

    def apply(self, x: Tensor, p: float) -> Tensor:
        return x


class DropoutSeeded:
    """Draw Bernoulli(1-p) keep-masks from a torch.Generator in call order; optionally record them.

    The golden generator patches the reference's nn.Dropout modules to draw from the same class, so the
    reference and the oracle consume identical mask streams when seeded identically.
    """

    def __init__(self, seed: int, record: bool = False):
        self.gen = torch.Generator().manual_seed(seed)
        self.record = record
        self.masks: List[Tensor] = []

    def apply(self, x: Tensor, p: float) -> Tensor:
        if p <= 0.0:
            return x
        keep = torch.bernoulli(torch.full(x.shape, 1.0 - p), generator=self.gen)
        if self.record:
            self.masks.append(keep.to(torch.uint8))
        return x * keep * (1.0 / (1.0 - p))


class DropoutFast:
    """Same distribution as DropoutSeeded, using the global RNG in place (cpu_baseline timing leg)."""

    def apply(self, x: Tensor, p: float) -> Tensor:
        if p <= 0.0:
            return x
        return F.dropout(x, p=p, training=True)


class DropoutFromList:
    """Replay recorded keep-masks (uint8, NCHW) in call order."""

    def __init__(self, masks: Sequence[Tensor]):
        self.masks = list(masks)
        self.pos = 0

    def apply(self, x: Tensor, p: float) -> Tensor:
        if p <= 0.0:
            return x
        keep = self.masks[self.pos].to(x.dtype)
        self.pos += 1
        assert keep.shape == x.shape, (keep.shape, x.shape)
        return x * keep * (1.0 / (1.0 - p))


# ----------------------------------------------------------------------------- shared pieces
def sinusoidal_features(t: Tensor, dim: int) -> Tensor:
    """misc.py:20-32: [sin(t*f_j), cos(t*f_j)], f_j = exp(-j*ln(1e4)/(dim/2-1))."""
    half = dim // 2
    freqs = torch.exp(torch.arange(half, dtype=torch.float32) * (-math.log(10000.0) / (half - 1)))
    ang = t.to(torch.float32)[:, None] * freqs[None, :]
    return torch.cat([ang.sin(), ang.cos()], dim=-1)


def time_embedding(P: Dict[str, Tensor], prefix: str, t: Tensor, dim: int) -> Tensor:
    """misc.py:54-67: sinusoid(dim) -> Linear -> GELU -> Linear; with learned_sinusoidal_cond (a `{prefix}.0.weights` parameter,
    misc.py:35-51) the features are [t, sin(2 pi t w), cos(2 pi t w)]."""
    if f"{prefix}.0.weights" in P:
        tt = t.float()[:, None]
        fr = tt * P[f"{prefix}.0.weights"][None, :] * 2 * math.pi
        e = torch.cat([tt, fr.sin(), fr.cos()], dim=-1)
    else:
        e = sinusoidal_features(t, dim)
    e = F.linear(e, P[f"{prefix}.1.weight"], P[f"{prefix}.1.bias"])
    e = F.gelu(e)
    return F.linear(e, P[f"{prefix}.3.weight"], P[f"{prefix}.3.bias"])


def film(P: Dict[str, Tensor], prefix: str, temb: Tensor):
    """SiLU -> Linear(time_dim, 2*C) -> (scale, shift) broadcast over space (unet_simple.py:21-23,72-78)."""
    ss = F.linear(F.silu(temb), P[f"{prefix}.1.weight"], P[f"{prefix}.1.bias"])
    scale, shift = ss[:, :, None, None].chunk(2, dim=1)
    return scale, shift


def bilinear_resize_explicit(x: Tensor, out_h: int, out_w: int) -> Tensor:
    """Bilinear resample with align_corners=False, no antialias, written out as gathers.

    This is the formula the HIP kernels implement; tests check it equals F.interpolate (ATen
    upsample_bilinear2d) which is what the reference calls (unet_simple.py:103,195).
      src = (dst + 0.5) * (in / out) - 0.5, clamped below at 0;  i0 = floor(src); i1 = min(i0 + 1, in - 1)
    """
    n, c, in_h, in_w = x.shape

    def axis(in_sz, out_sz):
        scale = in_sz / out_sz
        dst = torch.arange(out_sz, dtype=torch.float32)
        src = ((dst + 0.5) * scale - 0.5).clamp_min(0.0)
        i0 = src.floor().to(torch.int64).clamp_max(in_sz - 1)
        i1 = (i0 + 1).clamp_max(in_sz - 1)
        lam = src - i0.to(torch.float32)
        return i0, i1, lam

    y0, y1, ly = axis(in_h, out_h)
    x0, x1, lx = axis(in_w, out_w)
    top = x[:, :, y0, :]
    bot = x[:, :, y1, :]
    ly = ly[None, None, :, None]
    rows = top * (1.0 - ly) + bot * ly
    left = rows[:, :, :, x0]
    right = rows[:, :, :, x1]
    lx = lx[None, None, None, :]
    return left * (1.0 - lx) + right * lx


# ----------------------------------------------------------------------------- unet_simple.UNet (Navier-Stokes)
def unet_simple_layout(dim: int):
    """(cin, cout, kernel, stride, pad, norm, act) per block; unet_simple.py:119-139 with UNetBlock's arg mapping."""
    d = dim
    enc = [
        (d, 2 * d, 4, 2, 1, "bn", "leaky"),
        (2 * d, 2 * d, 4, 2, 1, "bn", "leaky"),
        (2 * d, 4 * d, 4, 2, 1, "bn", "leaky"),
        (4 * d, 8 * d, 4, 2, 1, "bn", "leaky"),
        (8 * d, 8 * d, 2, 2, 0, "bn", "leaky"),
        (8 * d, 8 * d, 2, 2, 0, "gn", "leaky"),
    ]
    # transposed blocks: Upsample(x2, bilinear) then Conv2d(kernel=size-1, stride 1, padding=pad)
    dec = [
        (8 * d, 8 * d, 1, 1, 0, "bn", "relu"),
        (16 * d, 8 * d, 1, 1, 0, "bn", "relu"),
        (16 * d, 4 * d, 3, 1, 1, "bn", "relu"),
        (8 * d, 2 * d, 3, 1, 1, "bn", "relu"),
        (4 * d, 2 * d, 3, 1, 1, "bn", "relu"),
        (4 * d, d, 3, 1, 1, "bn", "relu"),
    ]
    return enc, dec


def _norm(P, prefix, x, kind, bn_training: bool = False):
    if kind == "bn" and bn_training:  # module.train(): batch statistics (the running-statistics update is not modelled)
        return F.batch_norm(x, None, None, P[f"{prefix}.weight"], P[f"{prefix}.bias"], training=True, eps=1e-5)
    if kind == "bn":  # eval-mode BatchNorm2d: running statistics (BN always eval at sampling)
        return F.batch_norm(x, P[f"{prefix}.running_mean"], P[f"{prefix}.running_var"], P[f"{prefix}.weight"],
                            P[f"{prefix}.bias"], training=False, eps=1e-5)
    return F.group_norm(x, 8, P[f"{prefix}.weight"], P[f"{prefix}.bias"], eps=1e-5)


def unet_simple_forward(P: Dict[str, Tensor], cfg: dict, inputs: Tensor, time: Optional[Tensor] = None,
                        condition: Optional[Tensor] = None, dropout=None, taps: Optional[dict] = None,
                        bn_training: bool = False) -> Tensor:
    """unet_simple.py:181-197 + :164-179.  cfg keys: dim, upsample_dims (or None), outer_sample_mode,
    with_time_emb, dropout, input_dropout.  `taps` (optional dict) receives intermediate activations.  `bn_training`:
    BatchNorm2d as under module.train() (batch statistics) -- the training step of the forecaster."""
    dropout = dropout or DropoutOff()
    dim = cfg["dim"]
    mode = cfg.get("outer_sample_mode", "bilinear")
    x = torch.cat([inputs, condition], dim=1) if condition is not None else inputs
    temb = time_embedding(P, "time_emb_mlp", time, dim) if cfg.get("with_time_emb", False) else None
    native_hw = x.shape[-2:]
    if cfg.get("upsample_dims") is not None:
        x = F.interpolate(x, size=tuple(cfg["upsample_dims"]), mode=mode)
    x = conv2d(x, P["init_conv.weight"], P["init_conv.bias"])
    x = dropout.apply(x, cfg.get("input_dropout", 0.0))
    if taps is not None:
        taps["init"] = x
    enc, dec = unet_simple_layout(dim)
    p_drop = cfg.get("dropout", 0.0)
    skips = []
    for li, (_, _, k, s, pad, norm, act) in enumerate(enc):
        pre = f"input_ops.{li}"
        x = conv2d(x, P[f"{pre}.ops.0.weight"], P[f"{pre}.ops.0.bias"], stride=s, padding=pad)
        x = _norm(P, f"{pre}.ops.1", x, norm, bn_training)
        if temb is not None:
            scale, shift = film(P, f"{pre}.time_mlp", temb)
            x = x * (scale + 1) + shift
        x = F.leaky_relu(x, LEAKY_SLOPE)
        x = dropout.apply(x, p_drop)
        skips.append(x)
        if taps is not None:
            taps[f"enc{li}"] = x
    x = skips.pop()
    for li, (_, _, k, s, pad, norm, act) in enumerate(dec):
        pre = f"output_ops.{li}"
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
        x = conv2d(x, P[f"{pre}.ops.1.weight"], P[f"{pre}.ops.1.bias"], stride=1, padding=pad)
        x = _norm(P, f"{pre}.ops.2", x, norm, bn_training)
        if temb is not None:
            scale, shift = film(P, f"{pre}.time_mlp", temb)
            x = x * (scale + 1) + shift
        x = F.relu(x)
        x = dropout.apply(x, p_drop)
        if taps is not None:
            taps[f"dec{li}"] = x
        if skips:
            x = torch.cat([x, skips.pop()], dim=1)
    x = F.conv_transpose2d(x, P["readout.0.weight"], P["readout.0.bias"], stride=2, padding=1)
    if taps is not None:
        taps["readout"] = x
    return F.interpolate(x, size=tuple(native_hw), mode=mode)


def _r16(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def unet_simple_forward_bf16_model(P: Dict[str, Tensor], cfg: dict, inputs: Tensor, time: Optional[Tensor] = None,
                                   condition: Optional[Tensor] = None, dropout=None, round_act: bool = True,
                                   round_w: bool = True, taps: Optional[dict] = None) -> Tensor:
    """`unet_simple_forward` in the ARITHMETIC MODEL of the HIP engine: fp32 convolutions on bf16-rounded operands, every
    block output rounded to bf16 ONCE (after the fused norm / FiLM / activation / dropout epilogue), fp32 everything else.
    Not a restatement of the reference -- a measuring stick: the gap between this and `unet_simple_forward` is what bf16
    storage costs by itself (tests/measure_bf16_drift.py), the gap between the engine and this is the engine's own error."""
    dropout = dropout or DropoutOff()
    ra = _r16 if round_act else (lambda v: v)
    rw = _r16 if round_w else (lambda v: v)
    dim = cfg["dim"]
    x = torch.cat([inputs, condition], dim=1) if condition is not None else inputs
    temb = time_embedding(P, "time_emb_mlp", time, dim) if cfg.get("with_time_emb", False) else None
    native_hw = x.shape[-2:]
    if cfg.get("upsample_dims") is not None:
        x = F.interpolate(x, size=tuple(cfg["upsample_dims"]), mode="bilinear")
    x = ra(x)  # the resampled raw channels are stored in bf16; init_conv is composed into the first encoder conv
    x = F.conv2d(x, P["init_conv.weight"], P["init_conv.bias"])
    enc, dec = unet_simple_layout(dim)
    p_drop = cfg.get("dropout", 0.0)
    skips = []
    for li, (_, _, k, s, pad, norm, act) in enumerate(enc):
        pre = f"input_ops.{li}"
        x = F.conv2d(x, rw(P[f"{pre}.ops.0.weight"]), P[f"{pre}.ops.0.bias"], stride=s, padding=pad)
        x = _norm(P, f"{pre}.ops.1", x, norm)
        if temb is not None:
            scale, shift = film(P, f"{pre}.time_mlp", temb)
            x = x * (scale + 1) + shift
        x = ra(dropout.apply(F.leaky_relu(x, LEAKY_SLOPE), p_drop))
        skips.append(x)
        if taps is not None:
            taps[f"enc{li}"] = x
    x = skips.pop()
    for li, (_, _, k, s, pad, norm, act) in enumerate(dec):
        pre = f"output_ops.{li}"
        x = F.interpolate(x, scale_factor=2, mode="bilinear")
        if li < 3:
            x = ra(x)  # the small planes materialise the upsampled tensor (bf16); the large ones fuse it into the conv
        x = F.conv2d(x, rw(P[f"{pre}.ops.1.weight"]), P[f"{pre}.ops.1.bias"], stride=1, padding=pad)
        x = _norm(P, f"{pre}.ops.2", x, norm)
        if temb is not None:
            scale, shift = film(P, f"{pre}.time_mlp", temb)
            x = x * (scale + 1) + shift
        x = ra(dropout.apply(F.relu(x), p_drop))
        if taps is not None:
            taps[f"dec{li}"] = x
        if skips:
            x = torch.cat([x, skips.pop()], dim=1)
    x = F.conv_transpose2d(x, rw(P["readout.0.weight"]), P["readout.0.bias"], stride=2, padding=1)
    return F.interpolate(x, size=tuple(native_hw), mode="bilinear")


# ----------------------------------------------------------------------------- SimpleConvNet (spring-mesh plumbing)
def simple_conv_net_forward(P: Dict[str, Tensor], cfg: dict, inputs: Tensor, time: Optional[Tensor] = None,
                            condition: Optional[Tensor] = None, dropout=None) -> Tensor:
    """simple_conv_net.py:112-131 with ConvBlock :39-55.  cfg keys: dim, kernel_sizes, with_time_emb, dropout,
    residual."""
    dropout = dropout or DropoutOff()
    x = torch.cat([inputs, condition], dim=1) if condition is not None else inputs
    temb = time_embedding(P, "time_emb_mlp", time, cfg["dim"]) if cfg.get("with_time_emb", False) else None
    for li, k in enumerate(cfg["kernel_sizes"]):
        pre = f"convs.{li}"
        res = x
        w = P[f"{pre}.conv.weight"]
        x = conv2d(x, w, P[f"{pre}.conv.bias"], padding=(k - 1) // 2)
        x = F.batch_norm(x, P[f"{pre}.norm.running_mean"], P[f"{pre}.norm.running_var"], P[f"{pre}.norm.weight"],
                         P[f"{pre}.norm.bias"], training=False, eps=1e-5)
        if temb is not None:
            scale, shift = film(P, f"{pre}.time_mlp", temb)
            x = x * (scale + 1) + shift
        x = F.gelu(x)
        x = dropout.apply(x, cfg.get("dropout", 0.0))
        if cfg.get("residual", True) and w.shape[0] == w.shape[1]:
            x = x + res
    return conv2d(x, P["head.weight"], P["head.bias"])


# ----------------------------------------------------------------------------- unet.Unet (OISST / synthetic backbone)
def _ws_conv3x3(P, prefix, x):
    """WeightStandardizedConv2d (unet.py:26-40): per-out-channel (w - mean) * rsqrt(var + 1e-5), biased variance."""
    w = P[f"{prefix}.weight"]
    mean = w.mean(dim=(1, 2, 3), keepdim=True)
    var = w.var(dim=(1, 2, 3), unbiased=False, keepdim=True)
    return conv2d(x, (w - mean) * (var + 1e-5).rsqrt(), P[f"{prefix}.bias"], padding=1)


def _resnet_block(P, pre, x, temb, groups, p1, p2, dropout):
    """ResnetBlock (unet.py:79-109) = Block(FiLM) -> Block -> + residual_conv(x); Block = WS-conv, GroupNorm, FiLM, SiLU,
    Dropout (unet.py:58-76)."""
    h = _ws_conv3x3(P, f"{pre}.block1.proj", x)
    h = F.group_norm(h, groups, P[f"{pre}.block1.norm.weight"], P[f"{pre}.block1.norm.bias"], eps=1e-5)
    if temb is not None and f"{pre}.mlp.1.weight" in P:
        scale, shift = film(P, f"{pre}.mlp", temb)
        h = h * (scale + 1) + shift
    h = dropout.apply(F.silu(h), p1)
    if f"{pre}.block2.proj.weight" in P:  # double_conv_layer=False: block2 = nn.Identity() (unet.py:94)
        h = _ws_conv3x3(P, f"{pre}.block2.proj", h)
        h = F.group_norm(h, groups, P[f"{pre}.block2.norm.weight"], P[f"{pre}.block2.norm.bias"], eps=1e-5)
        h = dropout.apply(F.silu(h), p2)
    if f"{pre}.residual_conv.weight" in P:
        x = conv2d(x, P[f"{pre}.residual_conv.weight"], P[f"{pre}.residual_conv.bias"])
    return h + x


def _channel_layernorm(x, g):
    """unet.LayerNorm (unet.py:43-52): over the channel dim, biased variance, gain only, eps 1e-5 (fp32)."""
    var = x.var(dim=1, unbiased=False, keepdim=True)
    mean = x.mean(dim=1, keepdim=True)
    return (x - mean) * (var + 1e-5).rsqrt() * g


def _linear_attention(P, pre, x, heads, dim_head, p_attn, dropout):
    """Residual(PreNorm(LayerNorm, LinearAttention(rescale='qkv'))) (attention.py:7-44, net_norm.py:18-26, misc.py:8-14)."""
    b, c, hh, ww = x.shape
    n = hh * ww
    y = _channel_layernorm(x, P[f"{pre}.fn.norm.g"])
    y = dropout.apply(y, p_attn)
    qkv = conv2d(y, P[f"{pre}.fn.fn.to_qkv.1.weight"]).reshape(b, 3, heads, dim_head, n)
    q, k, v = qkv[:, 0], qkv[:, 1], qkv[:, 2]
    q = q.softmax(dim=-2) * dim_head ** -0.5
    k = k.softmax(dim=-1)
    v = v / n
    context = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", context, q).reshape(b, heads * dim_head, hh, ww)
    return conv2d(out, P[f"{pre}.fn.fn.to_out.weight"], P[f"{pre}.fn.fn.to_out.bias"]) + x


def _full_attention(P, pre, x, heads, dim_head, p_attn, dropout):
    """Residual(PreNorm(LayerNorm, Attention)) (attention.py:51-73): softmax(q*scale . k) over keys, dropout on the
    probabilities, times v."""
    b, c, hh, ww = x.shape
    n = hh * ww
    y = _channel_layernorm(x, P[f"{pre}.fn.norm.g"])
    qkv = conv2d(y, P[f"{pre}.fn.fn.to_qkv.weight"]).reshape(b, 3, heads, dim_head, n)
    q, k, v = qkv[:, 0] * dim_head ** -0.5, qkv[:, 1], qkv[:, 2]
    attn = torch.einsum("bhdi,bhdj->bhij", q, k).softmax(dim=-1)
    attn = dropout.apply(attn, p_attn)
    out = torch.einsum("bhij,bhdj->bhid", attn, v)                        # (b, h, n, d)
    out = out.permute(0, 1, 3, 2).reshape(b, heads * dim_head, hh, ww)    # "b h (x y) d -> b (h d) x y"
    return conv2d(out, P[f"{pre}.fn.fn.to_out.weight"], P[f"{pre}.fn.fn.to_out.bias"]) + x


def resnet_unet_forward(P: Dict[str, Tensor], cfg: dict, x: Tensor, time: Optional[Tensor] = None,
                        condition: Optional[Tensor] = None, dropout=None) -> Tensor:
    """unet.Unet.forward (unet.py:266-315).  cfg keys: dim, dim_mults, with_time_emb, block_dropout (second block),
    block_dropout1 (first block), attn_dropout, resnet_block_groups (8), init_kernel_size (7), init_padding (3);
    input_dropout (two sites: residual copy, then x), no outer resampling (the shipped OISST / synthetic settings).
    NOTE the condition goes FIRST in the channel concat here (unet.py:269), unlike unet_simple."""
    dropout = dropout or DropoutOff()
    dim, mults = cfg["dim"], tuple(cfg.get("dim_mults", (1, 2, 4)))
    groups = cfg.get("resnet_block_groups", 8)
    p2, p1, pa = cfg.get("block_dropout", 0.0), cfg.get("block_dropout1", 0.0), cfg.get("attn_dropout", 0.0)
    heads, dh = 4, 32
    keep = bool(cfg.get("keep_spatial_dims", False))  # unet.py:190,214: no down / up sampling, plain 3x3 convs instead
    if condition is not None:
        x = torch.cat([condition, x], dim=1)
    # (no outer resampler: the reference's Unet cannot be constructed with upsample_dims -- unet.py:155 reads an attribute that is never set)
    assert cfg.get("upsample_dims") is None
    x = conv2d(x, P["init_conv.weight"], P["init_conv.bias"], padding=cfg.get("init_padding", 3))
    # unet.py:276-277: two independent Dropouts on init_conv's output, the copy kept for the final residual first
    p_in = cfg.get("input_dropout", 0.0)
    r = dropout.apply(x, p_in) if p_in > 0 else x
    x = dropout.apply(x, p_in)
    temb = time_embedding(P, "time_emb_mlp", time, dim) if cfg.get("with_time_emb", False) else None
    nlev = len(mults)
    skips = []
    for li in range(nlev):
        pre = f"downs.{li}"
        x = _resnet_block(P, f"{pre}.0", x, temb, groups, p1, p2, dropout)
        skips.append(x)
        x = _resnet_block(P, f"{pre}.1", x, temb, groups, p1, p2, dropout)
        x = _linear_attention(P, f"{pre}.2", x, heads, dh, pa, dropout)
        skips.append(x)
        if li < nlev - 1 and not keep:
            x = conv2d(x, P[f"{pre}.3.weight"], P[f"{pre}.3.bias"], stride=2, padding=1)   # Downsample: k4 s2 p1
        else:
            x = conv2d(x, P[f"{pre}.3.weight"], P[f"{pre}.3.bias"], padding=1)
    x = _resnet_block(P, "mid_block1", x, temb, groups, p1, p2, dropout)
    x = _full_attention(P, "mid_attn", x, heads, dh, pa, dropout)
    x = _resnet_block(P, "mid_block2", x, temb, groups, p1, p2, dropout)
    for li in range(nlev):
        pre = f"ups.{li}"
        x = _resnet_block(P, f"{pre}.0", torch.cat([x, skips.pop()], dim=1), temb, groups, p1, p2, dropout)
        x = _resnet_block(P, f"{pre}.1", torch.cat([x, skips.pop()], dim=1), temb, groups, p1, p2, dropout)
        x = _linear_attention(P, f"{pre}.2", x, heads, dh, pa, dropout)
        if li < nlev - 1 and not keep:
            x = F.interpolate(x, scale_factor=2, mode="nearest")
            x = conv2d(x, P[f"{pre}.3.1.weight"], P[f"{pre}.3.1.bias"], padding=1)
        else:
            x = conv2d(x, P[f"{pre}.3.weight"], P[f"{pre}.3.bias"], padding=1)
    x = _resnet_block(P, "final_res_block", torch.cat([x, r], dim=1), temb, groups, p1, p2, dropout)
    return conv2d(x, P["final_conv.weight"], P["final_conv.bias"])
