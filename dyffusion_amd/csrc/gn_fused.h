// GroupNorm FUSED into the producing conv (ResnetBlock of src/models/unet.py:58-109): the conv's epilogue itself normalises,
// applies FiLM + SiLU + Dropout (+ the block's residual) and stores the finished activation -- the raw conv output, the
// statistics pass and the apply pass over it (gn_stats / gn_finalize / gn_apply_* kernels: 21 % of the OISST rollout's kernel time
// in round 3) no longer exist.
//
// GroupNorm needs the statistics of a whole (sample, group) before any element can be normalised, and a sample's pixels are
// spread over several workgroups of the conv (8 tiles of 16 x 32 pixels at 60 x 60).  So the workgroups of one sample meet
// INSIDE the launch:
//   1. every wave reduces (sum, sum of squares) of y = acc + bias over its pixels per 8-channel octet, straight from the fp32
//      accumulators, and publishes the 16 values as 8-byte {tag, value} granules (one aligned sc1 store each -- write-through, the
//      data IS the flag: MI355X_MICROARCH.md "Inter-workgroup visibility", recipe R2);
//   2. one wave per workgroup sweeps the granules of its sample's slots with sc1 loads until every tag matches, adds the values in
//      SLOT ORDER in fp64 (results do not depend on arrival order), finalises (mean, 1/std) per group and folds GroupNorm affine,
//      FiLM and the conv bias into one (A, C) pair per channel, parked in LDS;
//   3. all waves run the ordinary epilogue y = acc * A + C -> SiLU -> dropout -> (+ residual) -> 16-bit store.
// tag = (forward epoch << 8) | conv index: the epoch is a DEVICE word bumped by gn_epoch_bump_kernel at the head of every forward
// (kernel arguments are frozen under hipGraph replay; a word in memory is not), the conv index separates the convs of one forward
// that share the granule buffer.  Tags never repeat, so the buffer is zeroed once at allocation and never again.
//
// Residency: a workgroup waits only for workgroups of its own sample, which have neighbouring block ids (observed: blocks are
// dispatched in id order, XCD = id % 8); HIP does not promise that, so the sweep is BOUNDED (2 s): on time-out the wave raises the
// engine's host-visible error word, poisons its coefficients with NaN and goes on -- the launch always terminates, the failure is
// loud (NaN output + dyf_sample error) and the engine falls back to the three-kernel path.  The launchers use this form only
// when a sample's workgroups are few (<= 32) against the 512 resident ones.
#pragma once
#include "common.h"

struct GnFuse {
    unsigned long long* gran;  // [n][max_slots][cout / 8][2] granules {tag << 32 | float bits}; null = not fused
    const uint32_t* epoch;     // device word: forward counter (gn_epoch_bump_kernel)
    uint32_t conv_tag;         // index of this conv inside the forward (8 bits)
    int max_slots;             // slot stride of `gran` (>= the launch's slots per sample)
    int slots;                 // slots per sample of this launch (set by the launcher)
    int bm;                    // conv_igemm2_kernel: pixel rows of the workgroup tile the slots were counted for (0 / 256: the default, 128)
    int groups;
    const float* bias;         // conv bias [cout]
    const float* gamma;        // GroupNorm affine [cout]
    const float* beta;
    const float* film_a;       // (1 + scale) rows [row][film_stride] or null (second Block of a ResnetBlock: no FiLM)
    const float* film_c;
    int film_stride;
    int invariant;             // batch_invariant engines: only forms whose (mean, 1/std) bits do not depend on the sample's position in
                               // the launch (halo5, 2-D tiles, plane % 128 == 0: a sample's 128-row slabs start at its first pixel)
    uint32_t timeout_ticks;    // bound of a granule sweep in s_memrealtime ticks (100 MHz); 0 = GN_FUSE_TIMEOUT_TICKS (2 s)
    uint32_t test_tag_xor;     // test hook (dyf_debug_gn_fuse): the sweeps wait for tag ^ this -- a tag nobody publishes when nonzero, i.e.
                               // every sweep of the launch runs into its time-out, as if a workgroup of the sample never arrived
    uint32_t* err;             // host-visible (pinned, mapped) words: [0] nonzero after a sweep timed out (output NaN-poisoned),
                               // [1] nonzero after a sweep took > 1 024 passes (correct, but the workgroups are not co-scheduled)
};

#if defined(__HIP_DEVICE_COMPILE__)
typedef __attribute__((address_space(1))) unsigned long long gn_gu64;

__device__ __forceinline__ void gn_store_granule(unsigned long long* p, uint32_t tag, float v) {
    __hip_atomic_store((gn_gu64*)p, ((unsigned long long)tag << 32) | (unsigned long long)__float_as_uint(v), __ATOMIC_RELAXED,
                       __HIP_MEMORY_SCOPE_AGENT);  // ONE aligned 8-byte sc1 store
}

__device__ __forceinline__ double gn_shfl_xor_f64(double v, int d) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const uint32_t lo = (uint32_t)__shfl_xor((int)(uint32_t)u, d, 64), hi = (uint32_t)__shfl_xor((int)(uint32_t)(u >> 32), d, 64);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// ONE wave: wait for the `nslots` x 16 granules of one (sample, 64-channel block) -- base points at granule (slot 0, first octet
// of the block, sum), slots are `slot_stride` granules apart -- and return, in EVERY lane L, (mean, 1/std) of the group of channel
// L of the block.  cpg = channels per group (8, 16, 32 or 64), count = elements per (sample, group).
// Lane (v = L & 15, sg = L >> 4) adds the slots sg, sg + 4, ... of value v = 2 * octet + {0: sum, 1: sum of squares} in fp64; the
// four slot classes are combined by two exchanges (a + b is commutative: every lane gets the same bits).
// Bounded by TIME (s_memrealtime: a constant 100 MHz counter), not by passes: under time-slicing (two processes on one GPU) a sweep
// can legitimately take milliseconds; 2 s without a match is a dead launch.  The clock and the error word (another sweep of this
// engine already gave up: do not queue 2 s behind every later conv of the rollout) are looked at every 1 024 passes.
#ifndef GN_FUSE_TIMEOUT_TICKS
#define GN_FUSE_TIMEOUT_TICKS 200000000ull
#endif
template <int MAXJ>
__device__ __forceinline__ float2 gn_fuse_sweep(const unsigned long long* base, int slot_stride, int nslots, uint32_t tag, int cpg,
                                               double inv_count, uint32_t* err, int lane, uint32_t timeout_ticks = 0) {
    const unsigned long long limit = timeout_ticks ? (unsigned long long)timeout_ticks : GN_FUSE_TIMEOUT_TICKS;
    const int v = lane & 15, sg = lane >> 4;
    const gn_gu64* p = (const gn_gu64*)base + v;
    double part = 0.0;
    bool failed = false;
    unsigned long long t_start = 0;
    for (unsigned spins = 0;; ++spins) {
        bool ok = true;
        part = 0.0;
#pragma unroll
        for (int j = 0; j < MAXJ; ++j) {
            const int s = sg + 4 * j;
            if (s < nslots) {
                const unsigned long long x = __hip_atomic_load(p + (size_t)s * slot_stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ok = ok && (uint32_t)(x >> 32) == tag;
                part += (double)__uint_as_float((uint32_t)x);
            }
        }
        if (__all(ok)) break;
        if ((spins & 1023u) == 1023u) {  // wave-uniform
            const unsigned long long now = __builtin_amdgcn_s_memrealtime();
            if (t_start == 0) {
                t_start = now;
                // a sweep that needed 1 024 passes (> 1 ms; they take two or three when the chip is ours) means the sample's
                // workgroups are not co-scheduled -- a GPU shared with another process.  err[1]: the host then switches the
                // engine to the three-kernel GroupNorm path (no error: this launch still completes correctly)
                if (lane == 0 && err) __hip_atomic_store(err + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            const bool gave_up = err && __hip_atomic_load(err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0u;
            if (gave_up || now - t_start > limit) {
                failed = true;
                break;
            }
        }
        __builtin_amdgcn_s_sleep(8);
    }
    if (failed) {
        if (lane == 0 && err) __hip_atomic_store(err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        const float nan = __uint_as_float(0x7fc00000u);
        return make_float2(nan, nan);
    }
    part += gn_shfl_xor_f64(part, 16);
    part += gn_shfl_xor_f64(part, 32);
    // octets of one group: value index v = 2 * octet + k -> neighbours at xor 2, 4, 8
    for (int d = 2; d < (cpg >> 2); d <<= 1) part += gn_shfl_xor_f64(part, d);
    const double other = gn_shfl_xor_f64(part, 1);
    const double s1 = (v & 1) ? other : part, s2 = (v & 1) ? part : other;
    const double mean = s1 * inv_count;
    const double var = s2 * inv_count - mean * mean;
    const float mean_f = (float)mean, rstd_f = rsqrtf(fmaxf((float)var, 0.0f) + 1e-5f);
    const int src = 2 * (lane >> 3) & 15;  // a lane of the octet of channel `lane` (lanes 0..15 of the wave hold every octet)
    return make_float2(__shfl(mean_f, src, 64), __shfl(rstd_f, src, 64));
}

// (A, C) of channel `ch` (absolute) for sample row `frow` (FiLM row index): out = acc * A + C with the conv bias, GroupNorm affine
// and FiLM folded -- the operation order of gn_apply_*_kernel (unet_kernels.hip): A = gamma * rstd; C = beta - mean * A; then
// A *= (1 + scale); C = C * (1 + scale) + shift; then the bias: C += bias * A.
__device__ __forceinline__ float2 gn_fuse_coef(const GnFuse& g, int ch, int frow, float2 mr) {
    float A = g.gamma[ch] * mr.y;
    float C = fmaf(-mr.x, A, g.beta[ch]);
    if (g.film_a) {
        const float fa = g.film_a[(size_t)frow * g.film_stride + ch], fc = g.film_c[(size_t)frow * g.film_stride + ch];
        A *= fa;
        C = fmaf(C, fa, fc);
    }
    C = fmaf(g.bias[ch], A, C);
    return make_float2(A, C);
}
#endif
