"""CPU: the numpy restatement of the engine's dropout generator (tests/rng_host.py) against the C++ source itself.

csrc/common.h's generator functions are `__host__ __device__`: a probe program including the header is compiled with
hipcc (host pass only runs) and prints keep-mask words for a few (seed, forward, row, layer) tuples; the numpy
restatement must reproduce them bit for bit.  This pins the checker that the `-m gpu` RNG-mode rollout tests rely on."""
import os
import shutil
import subprocess

import numpy as np
import pytest

from tests import rng_host as R

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
PROBE = r'''
#include "common.h"
#include <cstdio>
int main() {
    const unsigned long long seeds[3] = {0ull, 1234567890123ull, 0xFEDCBA9876543210ull};
    for (int s = 0; s < 3; ++s)
        for (unsigned fwd = 0; fwd < 3; ++fwd)
            for (unsigned row = 0; row < 70; row += 23)
                for (unsigned layer = 0; layer < 12; layer += 5) {
                    RngKey k = rng_stream_key(rng_row_key((uint32_t)seeds[s], (uint32_t)(seeds[s] >> 32), fwd, row), rng_layer_salt(layer));
                    unsigned acc = 0;
                    for (unsigned e = 0; e < 4096; ++e) acc = acc * 31u + (rng_keep(e, k, keep_threshold16(0.15f)) ? 1u : 0u);
                    printf("%d %u %u %u %u %u %u\n", s, fwd, row, layer, k.k0, k.k1, acc);
                }
    printf("T %u %u %u\n", keep_threshold16(0.15f), keep_threshold16(0.6f), keep_threshold16(0.0f));
    return 0;
}
'''


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc")
def test_numpy_generator_equals_cpp_source(tmp_path):
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    src = tmp_path / "probe.hip"
    src.write_text(PROBE)
    exe = tmp_path / "probe"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "dyffusion_amd", "csrc"),
                    str(src), "-o", str(exe)], check=True, capture_output=True)
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.strip().splitlines()
    seeds = [0, 1234567890123, 0xFEDCBA9876543210]
    n = 0
    for line in out:
        if line.startswith("T"):
            _, a, b, c = line.split()
            assert (int(a), int(b), int(c)) == (int(R.keep_threshold16(0.15)), int(R.keep_threshold16(0.6)), int(R.keep_threshold16(0.0)))
            continue
        s, fwd, row, layer, k0, k1, acc = (int(v) for v in line.split())
        r0, r1 = R.row_key(seeds[s], fwd, row)
        s0, s1 = R.layer_salt(layer)
        assert int(r0 ^ s0) == k0 and int((r1 + s1) & R.M32) == k1, line
        mask = R.row_mask_nhwc((4096,), 0.15, seeds[s], fwd, layer, row).astype(np.uint64)
        want = 0
        for bit in mask:
            want = (want * 31 + int(bit)) & 0xFFFFFFFF
        assert want == acc, line
        n += 1
    assert n == 3 * 3 * 4 * 3


def test_generator_statistics():
    """keep rate = 1-p within sampling error; streams of different rows / forwards / layers are uncorrelated."""
    p = 0.15
    a = R.row_mask_nhwc((1 << 18,), p, 7, 0, 0, 0)
    assert abs(a.mean() - (1 - p)) < 4 * np.sqrt(p * (1 - p) / a.size)
    for other in (R.row_mask_nhwc((1 << 18,), p, 7, 0, 0, 1), R.row_mask_nhwc((1 << 18,), p, 7, 1, 0, 0),
                  R.row_mask_nhwc((1 << 18,), p, 7, 0, 1, 0), R.row_mask_nhwc((1 << 18,), p, 8, 0, 0, 0)):
        corr = np.corrcoef(a.astype(np.float64), other.astype(np.float64))[0, 1]
        assert abs(corr) < 0.01
        for lag in (1, 2, 64):
            c2 = np.corrcoef(a[lag:].astype(np.float64), other[:-lag].astype(np.float64))[0, 1]
            assert abs(c2) < 0.01
    # neighbouring elements of one stream (the two halves of a pair word, consecutive pair words)
    for lag in (1, 2, 3, 128):
        assert abs(np.corrcoef(a[lag:].astype(np.float64), a[:-lag].astype(np.float64))[0, 1]) < 0.01


def test_joint_structure_the_epilogues_consume():
    """What a forward actually draws: the SAME pair index (element e of a row) under the 13 site keys of one forward (12 UNetBlocks +
    dropout_input), under the two consecutive forward indices of a paired interpolator launch, and under neighbouring global rows.
    All 13 x 2 x 2 streams of one element must be jointly independent: pairwise correlations at sampling-error level, every pair's
    joint keep rate = (1 - p)^2, and the number of sites that keep an element binomial(13, 1 - p) (mean, variance, and the tails an
    MFMA tile sees: 32 consecutive elements of 13 sites never all dropped / all kept more often than chance)."""
    p, n = 0.15, 1 << 16
    keep = 1.0 - p
    streams = {}
    for fwd in (5, 6):
        for row in (40, 41):
            for layer in range(13):
                streams[(fwd, row, layer)] = R.row_mask_nhwc((n,), p, 20260929, fwd, layer, row).astype(np.float64)
    keys = sorted(streams)
    M = np.stack([streams[k] for k in keys], 0)
    assert np.all(np.abs(M.mean(1) - keep) < 5 * np.sqrt(p * keep / n))
    C = np.corrcoef(M)
    off = C[~np.eye(len(keys), dtype=bool)]
    assert np.abs(off).max() < 0.02, np.abs(off).max()  # 52 x 51 pairs, sampling error 1/sqrt(n) = 0.004
    # joint keep rate of every pair of streams
    J = (M @ M.T) / n
    offj = J[~np.eye(len(keys), dtype=bool)]
    assert np.abs(offj - keep * keep).max() < 6 * np.sqrt(keep * keep * (1 - keep * keep) / n)
    # the 13 sites of one (forward, row): kept-site count per element is binomial(13, keep)
    S = np.stack([streams[(5, 40, l)] for l in range(13)], 0).sum(0)
    assert abs(S.mean() - 13 * keep) < 5 * np.sqrt(13 * p * keep / n)
    assert abs(S.var() - 13 * p * keep) < 0.05 * 13 * p * keep
    from math import comb
    for k in (13, 12, 8):
        want = comb(13, k) * keep ** k * p ** (13 - k)
        got = float((S == k).mean())
        assert abs(got - want) < 5 * np.sqrt(want * (1 - want) / n) + 1e-4, (k, got, want)
    # lagged cross-correlations between sites (an epilogue walks consecutive pair indices of all sites together)
    a = streams[(5, 40, 0)]
    for l in (1, 5, 12):
        b = streams[(5, 40, l)]
        for lag in (1, 2, 31, 32):
            assert abs(np.corrcoef(a[lag:], b[:-lag])[0, 1]) < 0.02
