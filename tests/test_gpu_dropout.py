"""-m gpu: the engine's counter-based MC-dropout generator.

(1) The keep-masks are reproduced on the host with numpy from the documented hash (csrc/common.h) and fed to the
    oracle: the engine's RNG-mode forward must match the oracle run with exactly those masks.
(2) Statistics: keep-rate = 1-p within sampling error; different forwards / seeds give different masks.
"""
import numpy as np
import pytest
import torch

from oracle import init as oinit
from oracle import nets
from tests.gpu_common import DEV, mirror_from_params
from tests.helpers import rel_rms

pytestmark = pytest.mark.gpu
M32 = np.uint64(0xFFFFFFFF)


def fmix32(h):
    h = h.astype(np.uint64)
    h ^= h >> np.uint64(16)
    h = (h * np.uint64(0x85EBCA6B)) & M32
    h ^= h >> np.uint64(13)
    h = (h * np.uint64(0xC2B2AE35)) & M32
    h ^= h >> np.uint64(16)
    return h


def pair_word(pair_index, key):
    """common.h rng_pair_word: Weyl sequence through two xorshift / 24-bit-multiply rounds."""
    m24 = np.uint64(0xFFFFFF)
    x = (pair_index.astype(np.uint64) * np.uint64(0x9E3779B1) + np.uint64(key)) & M32
    x ^= x >> np.uint64(15)
    x = ((x & m24) * np.uint64(0x735A2D)) & M32
    x ^= x >> np.uint64(13)
    x = ((x & m24) * np.uint64(0x97E5B5)) & M32
    x ^= x >> np.uint64(16)
    return x


def layer_key(seed, fwd, layer):
    lo, hi = np.uint64(seed & 0xFFFFFFFF), np.uint64(seed >> 32)
    inner = fmix32(np.array([(int(hi) + 0x9E3779B9 * (fwd * 64 + layer + 1)) & 0xFFFFFFFF], dtype=np.uint64))
    return fmix32(np.array([int(lo) ^ int(inner[0])], dtype=np.uint64))[0]


def host_mask_nhwc(shape_nhwc, p, seed, fwd, layer):
    n = int(np.prod(shape_nhwc))
    e = np.arange(n, dtype=np.uint64)
    key = layer_key(seed, fwd, layer)
    w = pair_word(e >> np.uint64(1), key)
    v = np.where(e & np.uint64(1), w >> np.uint64(16), w & np.uint64(0xFFFF))
    thresh = np.uint64(int((np.float32(1.0) - np.float32(p)) * np.float32(65536.0)))  # keep_threshold16()
    return (v < thresh).reshape(shape_nhwc)


def test_rng_mode_equals_oracle_with_host_reproduced_masks():
    p = 0.15
    cfg = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=p)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=36)
    g = torch.Generator().manual_seed(10)
    x, c, t = torch.randn(2, 3, 23, 11, generator=g), torch.rand(2, 2, 23, 11, generator=g), torch.tensor([2.0, 3.0])
    net = mirror_from_params(P, cfg, 3, 2, 3)
    with net.inference_dropout_scope(True):
        net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV))  # forward #0 (also builds the engine)
        net._engine.seed(1234567890123)
        y0 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()  # forward index 0 after seeding
        y1 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()  # forward index 1
    assert not torch.equal(y0, y1)
    # layer output shapes (NHWC) of the 12 blocks at a 64x64 resampled grid, dim 64
    sizes = [32, 16, 8, 4, 2, 1, 2, 4, 8, 16, 32, 64]
    chans = [128, 128, 256, 512, 512, 512, 512, 512, 256, 128, 128, 64]
    for fwd, y in ((0, y0), (1, y1)):
        masks = [torch.from_numpy(host_mask_nhwc((2, s, s, ch), p, 1234567890123, fwd, l).astype(np.uint8))
                 .permute(0, 3, 1, 2).contiguous() for l, (s, ch) in enumerate(zip(sizes, chans))]
        keep = np.mean([float(m.float().mean()) for m in masks[:3]])
        assert abs(keep - (1 - p)) < 0.01
        with torch.no_grad():
            want = nets.unet_simple_forward(P, cfg, x, t, c, dropout=nets.DropoutFromList(masks))
        err = rel_rms(y, want)
        print("rng-mode vs oracle(host masks) forward", fwd, "rel-rms", err)
        assert err <= 1.5e-2
