"""numpy restatement of the engine's counter-based MC-dropout generator (dyffusion_amd/csrc/common.h: rng_row_key,
rng_layer_salt, rng_stream_key, rng_pair_word, keep_threshold16) -- test infrastructure.

The keep bit of an element is a pure function of (seed, forward index, GLOBAL batch row, dropout layer, element index
inside the row's NHWC tensor), so the masks of a whole rollout can be rebuilt on the host and fed to the oracle
(`oracle.nets.DropoutFromList`): RNG-mode outputs of the engine are then compared with the oracle on exactly those masks.
"""
import numpy as np
import torch

M32 = np.uint64(0xFFFFFFFF)
M24 = np.uint64(0xFFFFFF)


def _u(x):
    return np.asarray(x, dtype=np.uint64) & M32


def fmix32(h):
    h = _u(h)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def row_key(seed, fwd, grow):
    """rng_row_key(seed_lo, seed_hi, fwd, global row) -> (k0, k1); `grow` may be an array."""
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64((seed >> 32) & 0xFFFFFFFF)
    a = fmix32(lo ^ fmix32((hi + np.uint64(0x9E3779B9) * np.uint64(fwd + 1)) & M32))
    b = fmix32((a + np.uint64(0x85EBCA77) * (_u(grow) + np.uint64(1))) & M32)
    return b, fmix32(b ^ hi ^ np.uint64(0x27D4EB2F))


def layer_salt(layer):
    s = fmix32((np.uint64(0x9E3779B9) * np.uint64(layer + 1)) & M32)
    return s, fmix32((s + np.uint64(0x165667B1)) & M32)


def pair_word(pair_index, k0, k1):
    x = (_u(pair_index) * np.uint64(0x9E3779B1) + k0) & M32
    x ^= x >> np.uint64(15)
    x = ((x & M24) * np.uint64(0x735A2D) + k1) & M32
    x ^= x >> np.uint64(13)
    x = ((x & M24) * np.uint64(0x97E5B5)) & M32
    x ^= x >> np.uint64(16)
    return x


def keep_threshold16(p):
    keep = np.float32(np.float32(1.0) - np.float32(p)) * np.float32(65536.0)
    return np.uint64(65536 if keep >= 65536.0 else int(keep))


def row_mask_nhwc(shape_hwc, p, seed, fwd, layer, grow):
    """keep-mask (bool, H x W x C) of global batch row `grow` at dropout site `layer` of forward `fwd`."""
    n = int(np.prod(shape_hwc))
    e = np.arange(n, dtype=np.uint64)
    r0, r1 = row_key(seed, fwd, grow)
    s0, s1 = layer_salt(layer)
    w = pair_word(e >> np.uint64(1), r0 ^ s0, (r1 + s1) & M32)
    v = np.where(e & np.uint64(1), w >> np.uint64(16), w & np.uint64(0xFFFF))
    return (v < keep_threshold16(p)).reshape(shape_hwc)


def mask_nchw(nb, shape_hwc, p, seed, fwd, layer, row_offset=0):
    """uint8 keep-mask (NB, C, H, W) as the oracle's dropout layers see it."""
    rows = [row_mask_nhwc(shape_hwc, p, seed, fwd, layer, row_offset + r) for r in range(nb)]
    return torch.from_numpy(np.stack(rows, 0).astype(np.uint8)).permute(0, 3, 1, 2).contiguous()


def unet_simple_site_shapes(dim, uh, uw):
    """(H, W, C) of the 12 dropout sites (UNetBlock outputs) of unet_simple at a uh x uw resampled grid."""
    d = dim
    ch = [2 * d, 2 * d, 4 * d, 8 * d, 8 * d, 8 * d, 8 * d, 8 * d, 4 * d, 2 * d, 2 * d, d]
    hs, ws, h, w = [], [], uh, uw
    for i in range(6):
        h, w = h // 2, w // 2
        hs.append(h)
        ws.append(w)
    for i in range(6):
        h, w = h * 2, w * 2
        hs.append(h)
        ws.append(w)
    return [(hs[i], ws[i], ch[i]) for i in range(12)]


class EngineDropout:
    """Dropout source for `oracle.nets.unet_simple_forward(..., dropout=)` that replays the ENGINE's masks: call
    `begin_forward()` before every network forward that draws masks (same order as the engine's forward counter)."""

    INPUT_SITE = 12  # csrc/common.h DYF_INPUT_DROP_SITE: dropout_input on init_conv's output (only drawn when input_dropout > 0)

    def __init__(self, seed, dim, uh, uw, row_offset=0, first_forward=0, input_dropout=False):
        self.seed, self.row_offset = seed, row_offset
        blocks = unet_simple_site_shapes(dim, uh, uw)
        # (shape, layer id) in execution order: the oracle calls apply() for dropout_input first, then once per UNetBlock
        self.sites = ([((uh, uw, dim), self.INPUT_SITE)] if input_dropout else []) + [(sh, i) for i, sh in enumerate(blocks)]
        self.shapes = blocks
        self.fwd = first_forward - 1
        self.site = 0

    def begin_forward(self):
        self.fwd += 1
        self.site = 0

    def apply(self, x, p):
        if p <= 0.0:
            return x
        (h, w, c), layer = self.sites[self.site]
        assert tuple(x.shape[1:]) == (c, h, w), (x.shape, self.sites[self.site])
        keep = mask_nchw(x.shape[0], (h, w, c), p, self.seed, self.fwd, layer, self.row_offset).to(x.dtype)
        self.site += 1
        return x * keep * (1.0 / (1.0 - p))


class ResnetEngineDropout:
    """Dropout source for `oracle.nets.resnet_unet_forward(..., dropout=)` that replays the ENGINE's masks in the training step of
    the ResNet-UNet (csrc/train_resnet.inc): sites are numbered in execution order over the layers with p > 0; a feature map
    (b, C, H, W) is indexed NHWC inside its row, the bottleneck Attention's probabilities (b, heads, N, N) as (h * N + i) * N + j.
    Call `begin_forward()` before every network forward that draws masks."""

    def __init__(self, seed, attn_tokens, row_offset=0, first_forward=0):
        self.seed, self.row_offset, self.attn_tokens = seed, row_offset, attn_tokens
        self.fwd = first_forward - 1
        self.site = 0

    def begin_forward(self):
        self.fwd += 1
        self.site = 0

    def apply(self, x, p):
        if p <= 0.0:
            return x
        b = x.shape[0]
        probs = x.dim() == 4 and x.shape[1] == 4 and x.shape[2] == x.shape[3] == self.attn_tokens
        rows = []
        for r in range(b):
            if probs:
                m = row_mask_nhwc((4, self.attn_tokens, self.attn_tokens), p, self.seed, self.fwd, self.site, self.row_offset + r)
            else:
                c, h, w = x.shape[1:]
                m = row_mask_nhwc((h, w, c), p, self.seed, self.fwd, self.site, self.row_offset + r).transpose(2, 0, 1)
            rows.append(m)
        keep = torch.from_numpy(np.stack(rows, 0).astype(np.float32))
        self.site += 1
        return x * keep * (1.0 / (1.0 - p))
