// HBM/LDS-bound kernels of the ResNet-UNet backbone (src/models/unet.py + modules/attention.py) for gfx950.
#include "unet_kernels.h"

#include <algorithm>
#include <cstdlib>

__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ float wmax(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float bsum(float v, float* scratch) {
    v = wsum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// ------------------------------------------------------------------------------------------------ init_conv
// One thread per pixel, 16 output channels at a time.  Weights sit in LDS as [tap][cin][dim]: every lane reads the same
// address (broadcast), four channels per ds_read_b128; results leave as 16-byte stores (the first version issued one 2-byte
// store per channel and one ds_read_b32 per FMA: 1.1 ms per call on the OISST grid; this form 0.55 ms, LDS-issue bound --
// fetching the weights through the scalar unit instead was slower, 1.2 ms).
__global__ __launch_bounds__(256) void stem_conv_kernel(StemConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) float wsh[];  // [k*k*cin][dim]
    const int wcount = a.k * a.k * a.cin * a.dim;
    for (int i = threadIdx.x; i < wcount; i += blockDim.x) wsh[i] = a.wgt[i];
    __syncthreads();
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.n * a.h * a.w;
    if (pix >= total) return;
    const int n = (int)(pix / ((long long)a.h * a.w));
    const int rem = (int)(pix % ((long long)a.h * a.w));
    const int y = rem / a.w, x = rem % a.w;
    el16_t* out = a.out + (size_t)pix * a.dim;
    const bool vec = (a.dim & 15) == 0;
    for (int d0 = 0; d0 < a.dim; d0 += 16) {
        float acc[16];
#pragma unroll
        for (int t = 0; t < 16; ++t) acc[t] = (d0 + t < a.dim) ? a.bias[d0 + t] : 0.0f;
        for (int ky = 0; ky < a.k; ++ky) {
            const int iy = y + ky - a.pad;
            if ((unsigned)iy >= (unsigned)a.h) continue;
            for (int kx = 0; kx < a.k; ++kx) {
                const int ix = x + kx - a.pad;
                if ((unsigned)ix >= (unsigned)a.w) continue;
                int cbase = 0;
                for (int s = 0; s < a.nsrc; ++s) {
                    const float* src = a.src[s] + ((size_t)n * a.ch[s]) * a.h * a.w + (size_t)iy * a.w + ix;
                    for (int c = 0; c < a.ch[s]; ++c) {
                        const float v = src[(size_t)c * a.h * a.w];
                        const float* wr = wsh + ((size_t)(ky * a.k + kx) * a.cin + cbase + c) * a.dim + d0;
                        if (vec) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                const float4 w4 = *(const float4*)(wr + 4 * q);
                                acc[4 * q + 0] = fmaf(v, w4.x, acc[4 * q + 0]);
                                acc[4 * q + 1] = fmaf(v, w4.y, acc[4 * q + 1]);
                                acc[4 * q + 2] = fmaf(v, w4.z, acc[4 * q + 2]);
                                acc[4 * q + 3] = fmaf(v, w4.w, acc[4 * q + 3]);
                            }
                        } else {
#pragma unroll
                            for (int t = 0; t < 16; ++t)
                                if (d0 + t < a.dim) acc[t] = fmaf(v, wr[t], acc[t]);
                        }
                    }
                    cbase += a.ch[s];
                }
            }
        }
        if (vec) {
            uint4 o0, o1;
            o0.x = pack_el16x2(acc[0], acc[1]); o0.y = pack_el16x2(acc[2], acc[3]);
            o0.z = pack_el16x2(acc[4], acc[5]); o0.w = pack_el16x2(acc[6], acc[7]);
            o1.x = pack_el16x2(acc[8], acc[9]); o1.y = pack_el16x2(acc[10], acc[11]);
            o1.z = pack_el16x2(acc[12], acc[13]); o1.w = pack_el16x2(acc[14], acc[15]);
            *(uint4*)(out + d0) = o0;
            *(uint4*)(out + d0 + 8) = o1;
        } else {
#pragma unroll
            for (int t = 0; t < 16; ++t)
                if (d0 + t < a.dim) out[d0 + t] = f32_to_el16(acc[t]);
        }
    }
}

// MFMA form (dim = 64): D[channel][pixel] = W[channel][(cin, tap)] . X[(cin, tap)][pixel].  The contraction index is CHANNEL-major
// with the k*k taps of a channel padded to TS = ceil(k*k / 16) k16 steps (7 x 7: 49 -> 64): a step then gathers 8 taps per lane of
// ONE input channel plane, so
//   * the plane is a wave-uniform buffer resource (base = source + (sample, channel) plane, 4 * h * w bytes) and the per-lane part
//     of an address -- 4 (rem + dy w + dx), or 0xFFFFFFFF for a tap that falls into the zero padding: the buffer bounds check
//     returns 0 for it -- is the SAME for every channel: TS * 8 offsets computed once per pixel group, kept in registers;
//   * a step is 8 buffer loads with no address arithmetic, the hi / lo split and 6 MFMAs; the loads of channel c + 1 are issued
//     before the MFMAs of channel c.
// One wave owns 32 consecutive pixels of one sample (MFMA columns; neighbouring lanes = neighbouring pixels: coalesced; the
// k*k-fold reuse of an input lives in L1).  Inputs and weights are split into 16-bit hi + lo parts and multiplied as
// hi*hi + lo*hi + hi*lo, so the result matches the fp32 VALU form to ~2^-16 -- the network input is not rounded to 16 bits.
// The weight fragments ([cin * TS steps][2][hi/lo][64 lanes] x 16 B, <= 128 KB) sit in LDS.
// (First form, rounds 1-2: tap-major K with a per-element {offset, dy, dx, source} table in LDS: bounds checks, a source-pointer
// select and 64-bit address arithmetic per gathered value compiled into ~1 000 instructions per k16 step -- 1.31 ms per launch at
// 4 rows of 512^2 x 8 channels, VALU/branch-bound at one wave per SIMD.)
typedef __attribute__((ext_vector_type(16))) float st_f32x16;
constexpr int STEM_MAX_KSTEPS = 32;  // cin * TS <= 32 (7 x 7 x 8 channels): 128 KB of weight fragments in LDS

template <int TS>
__global__ __launch_bounds__(256) void stem_mfma_kernel(StemConvArgs a, int gps) {
    extern __shared__ __attribute__((aligned(16))) char st_smem[];
    const uint4* wf = (const uint4*)st_smem;  // [cin * TS][2][2][64] x 16 B
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    const int plane = a.h * a.w, kk = a.k * a.k;
    {
        uint4* wdst = (uint4*)st_smem;
        const uint4* wsrc = (const uint4*)a.wfrag;
        for (int i = tid; i < a.cin * TS * 4 * 64; i += 256) wdst[i] = wsrc[i];
    }
    int dydx[TS * 8];  // tap displacement of this lane's j-th value of step t: dy | dx << 16; dy = 0x7fff: no such tap
#pragma unroll
    for (int i = 0; i < TS * 8; ++i) {
        const int tap = (i >> 3) * 16 + hi * 8 + (i & 7);
        const int dy = tap / a.k - a.pad, dx = tap % a.k - a.pad;
        dydx[i] = tap < kk ? ((dy & 0xffff) | (dx << 16)) : 0x7fff;
    }
    const float *sp0 = a.src[0], *sp1 = a.src[1], *sp2 = a.src[2], *sp3 = a.src[3];
    const int ch0 = a.ch[0], ch1 = a.ch[1], ch2 = a.ch[2], ch3 = a.ch[3];
    __syncthreads();
    const int ngroups = a.n * gps;
    for (int g = blockIdx.x * 4 + wave; g < ngroups; g += gridDim.x * 4) {
        const int n = g / gps, rem = (g - n * gps) * 32 + l31;
        const bool pvalid = rem < plane;
        const int y = rem / a.w, x = rem - y * a.w;
        unsigned voff[TS * 8];
#pragma unroll
        for (int i = 0; i < TS * 8; ++i) {
            const int dy = (int)(short)(dydx[i] & 0xffff), dx = dydx[i] >> 16;
            const bool ok = pvalid && (unsigned)(y + dy) < (unsigned)a.h && (unsigned)(x + dx) < (unsigned)a.w;
            voff[i] = ok ? (unsigned)(rem + dy * a.w + dx) * 4u : 0xFFFFFFFFu;
        }
        st_f32x16 acc[2];
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rb][r] = 0.0f;
        float v[TS * 8], nv[TS * 8];
        int src = 0, cl = 0;  // (source, channel inside the source) of the channel being fetched: wave-uniform
        auto fetch = [&](float (&dst)[TS * 8]) {
            const float* sp = src == 0 ? sp0 : src == 1 ? sp1 : src == 2 ? sp2 : sp3;
            const int chs = src == 0 ? ch0 : src == 1 ? ch1 : src == 2 ? ch2 : ch3;
            const float* base = sp + ((size_t)n * chs + cl) * plane;
            const auto rs = __builtin_amdgcn_make_buffer_rsrc((void*)base, 0, plane * 4, 0x00020000);
#pragma unroll
            for (int i = 0; i < TS * 8; ++i) dst[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], 0, 0));
            if (++cl == chs) { cl = 0; ++src; }
        };
        fetch(v);
        for (int c = 0; c < a.cin; ++c) {
            if (c + 1 < a.cin) fetch(nv);
#pragma unroll
            for (int t = 0; t < TS; ++t) {
                uint32_t bh[4], bl[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    bh[q] = pack_el16x2(v[t * 8 + 2 * q], v[t * 8 + 2 * q + 1]);
                    bl[q] = pack_el16x2(v[t * 8 + 2 * q] - el16_lo(bh[q]), v[t * 8 + 2 * q + 1] - el16_hi(bh[q]));
                }
                const el16x8_t xh = __builtin_bit_cast(el16x8_t, make_uint4(bh[0], bh[1], bh[2], bh[3]));
                const el16x8_t xl = __builtin_bit_cast(el16x8_t, make_uint4(bl[0], bl[1], bl[2], bl[3]));
                const int step = c * TS + t;
#pragma unroll
                for (int rb = 0; rb < 2; ++rb) {
                    const el16x8_t wh = __builtin_bit_cast(el16x8_t, wf[((step * 2 + rb) * 2 + 0) * 64 + lane]);
                    const el16x8_t wl = __builtin_bit_cast(el16x8_t, wf[((step * 2 + rb) * 2 + 1) * 64 + lane]);
                    acc[rb] = DYF_MFMA_32x32x16(wh, xh, acc[rb], 0, 0, 0);
                    acc[rb] = DYF_MFMA_32x32x16(wl, xh, acc[rb], 0, 0, 0);
                    acc[rb] = DYF_MFMA_32x32x16(wh, xl, acc[rb], 0, 0, 0);
                }
            }
            if (c + 1 < a.cin) {
#pragma unroll
                for (int i = 0; i < TS * 8; ++i) v[i] = nv[i];
            }
        }
        // lane (pixel, hi) holds channels rb*32 + 8 (r >> 2) + 4 hi + (r & 3); groups 2 g2 / 2 g2 + 1 are exchanged between
        // lanes p and p + 32 so that every lane stores 8 consecutive channels (as in the conv epilogues)
        el16_t* op = a.out + ((size_t)n * plane + rem) * a.dim + hi * 8;
#pragma unroll
        for (int rb = 0; rb < 2; ++rb)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int cb = rb * 32 + g2 * 16 + 4 * hi;
                const float4 ba = *(const float4*)(a.bias + cb), bb = *(const float4*)(a.bias + cb + 8);
                const uint32_t p0 = pack_el16x2(acc[rb][g2 * 8 + 0] + ba.x, acc[rb][g2 * 8 + 1] + ba.y);
                const uint32_t p1 = pack_el16x2(acc[rb][g2 * 8 + 2] + ba.z, acc[rb][g2 * 8 + 3] + ba.w);
                const uint32_t q0 = pack_el16x2(acc[rb][g2 * 8 + 4] + bb.x, acc[rb][g2 * 8 + 5] + bb.y);
                const uint32_t q1 = pack_el16x2(acc[rb][g2 * 8 + 6] + bb.z, acc[rb][g2 * 8 + 7] + bb.w);
                const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                uint4 o;
                o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                if (pvalid) *(uint4*)(op + rb * 32 + g2 * 16) = o;
            }
    }
}

int stem_frag_steps(int k, int cin) {  // k16 steps of the MFMA stem, or 0 if this shape runs on the VALU form
    const int ts = (k * k + 15) / 16;
    if ((ts != 1 && ts != 2 && ts != 4) || cin * ts > STEM_MAX_KSTEPS) return 0;
    return cin * ts;
}

void pack_stem_frag(const float* wgt, int k, int cin, int dim, el16_t* out) {
    const int kk = k * k, ts = (kk + 15) / 16;
    size_t o = 0;
    for (int c = 0; c < cin; ++c)
        for (int t = 0; t < ts; ++t)
            for (int rb = 0; rb < dim / 32; ++rb)
                for (int hl = 0; hl < 2; ++hl)
                    for (int lane = 0; lane < 64; ++lane)
                        for (int j = 0; j < 8; ++j) {
                            const int tap = t * 16 + (lane >> 5) * 8 + j, ch = rb * 32 + (lane & 31);
                            const float w = tap < kk ? wgt[((size_t)tap * cin + c) * dim + ch] : 0.0f;
                            const el16_t h = f32_to_el16(w);
                            out[o++] = hl == 0 ? h : f32_to_el16(w - el16_to_f32(h));
                        }
}

hipError_t launch_stem_conv(const StemConvArgs& a, hipStream_t s) {
    const long long total = (long long)a.n * a.h * a.w;
    if (a.wfrag && a.dim == 64 && a.ksteps >= 1 && a.ksteps == stem_frag_steps(a.k, a.cin) && a.nsrc <= 4 &&
        (long long)a.h * a.w * 4 < (1ll << 31)) {
        const bool use_mfma = !(dyf_form("DYF_STEM_MFMA") && atoi(dyf_form("DYF_STEM_MFMA")) == 0);
        if (use_mfma) {
            const int ts = (a.k * a.k + 15) / 16;
            const size_t lds = (size_t)a.ksteps * 4 * 64 * 16;
            static bool attr = false;
            if (!attr) {
                const int cap = STEM_MAX_KSTEPS * 4 * 64 * 16;
                hipError_t e = hipFuncSetAttribute((const void*)stem_mfma_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_mfma_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                if (e == hipSuccess) e = hipFuncSetAttribute((const void*)stem_mfma_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, cap);
                if (e != hipSuccess) return e;
                attr = true;
            }
            const int gps = (a.h * a.w + 31) / 32;  // pixel groups per sample (the last one of a sample may be partial)
            const long long ngroups = (long long)a.n * gps;
            if (ngroups < (1ll << 31)) {
                // persistent: as many workgroups as fit the chip at once (each copies the fragments, up to 128 KB, into LDS first)
                const long long resident = 256ll * std::max<long long>(1, std::min<long long>(4, (160 * 1024) / (long long)lds));
                const unsigned grid = (unsigned)std::min<long long>((ngroups + 3) / 4, resident);
                dyf_form_note("stem_mfma_kernel", a.n);
                if (ts == 1) hipLaunchKernelGGL(stem_mfma_kernel<1>, dim3(grid), dim3(256), lds, s, a, gps);
                else if (ts == 2) hipLaunchKernelGGL(stem_mfma_kernel<2>, dim3(grid), dim3(256), lds, s, a, gps);
                else hipLaunchKernelGGL(stem_mfma_kernel<4>, dim3(grid), dim3(256), lds, s, a, gps);
                return hipGetLastError();
            }
        }
    }
    hipLaunchKernelGGL(stem_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256),
                       (size_t)a.k * a.k * a.cin * a.dim * sizeof(float), s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ GroupNorm + act
// One workgroup per (sample, group): wavefront reductions for mean / centred variance, then the fused epilogue.
__global__ __launch_bounds__(256) void gn_act_kernel(GnActArgs a) {
    __shared__ float scratch[16];
    const int n = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    const int cpg = a.c / a.groups;
    const int count = a.hw * cpg;
    const el16_t* x = a.x + (size_t)n * a.hw * a.c + g * cpg;
    float s = 0.0f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += el16_to_f32(x[(size_t)(i / cpg) * a.c + (i % cpg)]);
    const float mean = bsum(s, scratch) / (float)count;
    float v = 0.0f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const float d = el16_to_f32(x[(size_t)(i / cpg) * a.c + (i % cpg)]) - mean;
        v = fmaf(d, d, v);
    }
    const float rstd = rsqrtf(bsum(v, scratch) / (float)count + 1e-5f);
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int p = i / cpg, ch = g * cpg + (i % cpg);
        const size_t e = ((size_t)n * a.hw + p) * a.c + ch;
        float y = (el16_to_f32(a.x[e]) - mean) * rstd * a.gamma[ch] + a.beta[ch];
        if (a.film_a) {
            const size_t fi = (size_t)n * a.film_stride + ch;
            y = fmaf(y, a.film_a[fi], a.film_c[fi]);
        }
        y = apply_act(y, a.act);
        y = drop_apply(y, (uint32_t)e, row0, a.drop, key);
        if (a.residual) y += el16_to_f32(a.residual[e]);
        a.out[e] = f32_to_el16(y);
    }
}

// Vectorised two-kernel form (c % 8 == 0, 8-channel chunks never straddle a group): every lane owns one 16-byte chunk
// of one pixel.  (1) statistics: per-workgroup LDS partials -> one fp64 atomic per (sample, group) and workgroup;
// (2) apply: normalise + FiLM + SiLU + dropout (+ residual), one 16-B load and one 16-B store per lane.
// HBM traffic: 2 reads + 1 write of the tensor (+1 read of the residual) instead of 3 strided read passes.
// Deterministic: every wave owns a slot per group in LDS (one lane per group and wave writes it), the four slots are added in
// wave order, and each workgroup stores its partial to stats[n][group][sum | sum of squares][workgroup]; gn_finalize_kernel
// adds the workgroups' partials in index order.  (Atomic merges made GroupNorm -- and with it every rollout of the ResNet-UNet
// -- differ from run to run by a rounding flip that the sampling recursion amplifies.)
__global__ __launch_bounds__(256) void gn_stats_kernel(const el16_t* x, int hw, int c, int groups, double* stats) {
    __shared__ float part[4][64][2];  // [wave][group]; groups <= 64
    const int n = blockIdx.y;
    const int chunks = c >> 3, cpg = c / groups;
    for (int i = threadIdx.x; i < 4 * 64 * 2; i += blockDim.x) (&part[0][0][0])[i] = 0.0f;
    __syncthreads();
    const int cq = cpg >> 3;  // chunks per group
    const bool pow2 = (chunks & (chunks - 1)) == 0 && (cq & (cq - 1)) == 0 && chunks <= 256;
    if (pow2) {
        // a thread keeps ONE chunk column (q) and walks pixels: no index arithmetic in the loop, the group never changes, so
        // the partial sums stay in registers until the end
        const int q = threadIdx.x & (chunks - 1), row = threadIdx.x / chunks, rows = 256 / chunks;
        const el16_t* xp = x + (size_t)n * hw * c + q * 8;
        float s = 0.0f, ss = 0.0f;
        const int step = gridDim.x * rows;
        int p = blockIdx.x * rows + row;
        auto add = [&](const uint4& v) {
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float lo = el16_lo(w[t]), hi = el16_hi(w[t]);
                s += lo + hi;
                ss = fmaf(lo, lo, fmaf(hi, hi, ss));
            }
        };
        for (; p + 3 * step < hw; p += 4 * step) {  // four independent 16-B loads in flight
            const uint4 v0 = *(const uint4*)(xp + (size_t)p * c), v1 = *(const uint4*)(xp + (size_t)(p + step) * c);
            const uint4 v2 = *(const uint4*)(xp + (size_t)(p + 2 * step) * c), v3 = *(const uint4*)(xp + (size_t)(p + 3 * step) * c);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; p < hw; p += step) add(*(const uint4*)(xp + (size_t)p * c));
        // lanes l and l ^ d hold the same group when d < cq (neighbouring chunks of the group) or d >= chunks (same chunk,
        // another pixel row): butterfly over those strides, then one LDS slot per group and wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            if (d < cq || d >= chunks) {
                s += __shfl_xor(s, d, 64);
                ss += __shfl_xor(ss, d, 64);
            }
        }
        const int l = threadIdx.x & 63;
        if ((l & (cq - 1)) == 0 && (chunks >= 64 || l < chunks)) {
            const int g = (q * 8) / cpg;
            part[threadIdx.x >> 6][g][0] = s;   // exactly one lane of the wave holds group g
            part[threadIdx.x >> 6][g][1] = ss;
        }
    } else {
        const long long total = (long long)hw * chunks;
        for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
            const int q = (int)(idx % chunks);
            const long long p = idx / chunks;
            const uint4 v = *(const uint4*)(x + ((size_t)n * hw + p) * c + q * 8);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
            float s = 0.0f, ss = 0.0f;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float lo = el16_lo(w[t]), hi = el16_hi(w[t]);
                s += lo + hi;
                ss = fmaf(lo, lo, fmaf(hi, hi, ss));
            }
            const int g = (q * 8) / cpg;
            atomicAdd(&part[0][g][0], s);   // channel counts whose 16-byte chunks are not a power of two: order not fixed
            atomicAdd(&part[0][g][1], ss);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < groups * 2; i += blockDim.x) {
        const int g = i >> 1, k = i & 1;
        const double v = (double)part[0][g][k] + (double)part[1][g][k] + (double)part[2][g][k] + (double)part[3][g][k];
        stats[(((size_t)n * groups + g) * 2 + k) * GN_MAX_BLOCKS + blockIdx.x] = v;
    }
}

// (sum, sum of squares) in fp64 -> (mean, 1/std) in fp32, once per (sample, group) instead of once per lane of the apply pass
__global__ void gn_finalize_kernel(const double* stats, int count, int nblocks, double inv_cnt, float2* mr) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    double s = 0.0, ss = 0.0;
    for (int b = 0; b < nblocks; ++b) {  // fixed order
        s += stats[((size_t)i * 2 + 0) * GN_MAX_BLOCKS + b];
        ss += stats[((size_t)i * 2 + 1) * GN_MAX_BLOCKS + b];
    }
    const double mean = s * inv_cnt;
    const double var = ss * inv_cnt - mean * mean;
    mr[i] = make_float2((float)mean, rsqrtf(fmaxf((float)var, 0.0f) + 1e-5f));
}

__global__ __launch_bounds__(256) void gn_apply_kernel(GnActArgs a, const float2* mr) {
    const int chunks = a.c >> 3, cpg = a.c / a.groups;
    const long long total = (long long)a.n * a.hw * chunks;
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int q = (int)(idx % chunks);
    const long long pix = idx / chunks;
    const int n = (int)(pix / a.hw);
    const int g = (q * 8) / cpg;
    const float2 ms = mr[(size_t)n * a.groups + g];
    const float mean = ms.x, rstd = ms.y;
    const size_t e0 = (size_t)pix * a.c + q * 8;
    const uint4 v = *(const uint4*)(a.x + e0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    // GroupNorm affine and FiLM folded into one FMA per element: y = x * A + C
    const float4 g0 = *(const float4*)(a.gamma + q * 8), g1 = *(const float4*)(a.gamma + q * 8 + 4);
    const float4 b0 = *(const float4*)(a.beta + q * 8), b1 = *(const float4*)(a.beta + q * 8 + 4);
    float A[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float C[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        A[t] *= rstd;
        C[t] = fmaf(-mean, A[t], C[t]);
    }
    if (a.film_a) {
        const size_t fi = (size_t)n * a.film_stride + q * 8;
        const float4 fa0 = *(const float4*)(a.film_a + fi), fa1 = *(const float4*)(a.film_a + fi + 4);
        const float4 fc0 = *(const float4*)(a.film_c + fi), fc1 = *(const float4*)(a.film_c + fi + 4);
        const float fa[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
        const float fc[8] = {fc0.x, fc0.y, fc0.z, fc0.w, fc1.x, fc1.y, fc1.z, fc1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[t] *= fa[t];
            C[t] = fmaf(C[t], fa[t], fc[t]);
        }
    }
    float y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        const float xv = (t & 1) ? el16_hi(w[t >> 1]) : el16_lo(w[t >> 1]);
        y[t] = fmaf(xv, A[t], C[t]);
    }
    act_drop<8>(y, (uint32_t)e0, (uint32_t)((size_t)n * a.hw * a.c), a.act, a.drop, drop_row_key(a.drop, n));  // (activation, dropout mode) dispatched once
    if (a.residual) {
        const uint4 r = *(const uint4*)(a.residual + e0);
        const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int t = 0; t < 8; ++t)
            y[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
    }
    *(uint4*)(a.out + e0) = make_uint4(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]), pack_el16x2(y[4], y[5]),
                                       pack_el16x2(y[6], y[7]));
}

// Walk form of gn_apply_kernel (channel chunks per pixel a power of two <= 256): a thread keeps ONE 16-byte channel chunk and
// applies it to GP pixels of its sample: the per-chunk coefficient setup (GroupNorm affine x FiLM: 6 vector loads, ~50 VALU) is
// paid once instead of per pixel, and the GP tensor loads (and residual loads) are in flight together.
#ifndef GN_GP
#define GN_GP 4
#endif
constexpr int GP = GN_GP;
__global__ __launch_bounds__(256) void gn_apply_walk_kernel(GnActArgs a, const float2* mr) {
    const int chunks = a.c >> 3, cpg = a.c / a.groups;
    const int n = blockIdx.y;
    const int q = threadIdx.x & (chunks - 1), row = threadIdx.x / chunks, rows = 256 / chunks;
    const int g = (q * 8) / cpg;
    const float2 ms = mr[(size_t)n * a.groups + g];
    const float mean = ms.x, rstd = ms.y;
    const float4 g0 = *(const float4*)(a.gamma + q * 8), g1 = *(const float4*)(a.gamma + q * 8 + 4);
    const float4 b0 = *(const float4*)(a.beta + q * 8), b1 = *(const float4*)(a.beta + q * 8 + 4);
    float A[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float C[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        A[t] *= rstd;
        C[t] = fmaf(-mean, A[t], C[t]);
    }
    if (a.film_a) {
        const size_t fi = (size_t)n * a.film_stride + q * 8;
        const float4 fa0 = *(const float4*)(a.film_a + fi), fa1 = *(const float4*)(a.film_a + fi + 4);
        const float4 fc0 = *(const float4*)(a.film_c + fi), fc1 = *(const float4*)(a.film_c + fi + 4);
        const float fa[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
        const float fc[8] = {fc0.x, fc0.y, fc0.z, fc0.w, fc1.x, fc1.y, fc1.z, fc1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[t] *= fa[t];
            C[t] = fmaf(C[t], fa[t], fc[t]);
        }
    }
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
    const int p0 = blockIdx.x * rows * GP + row;
    uint4 v[GP], r[GP];
#pragma unroll
    for (int i = 0; i < GP; ++i) {
        const int p = min(p0 + i * rows, a.hw - 1);  // past the end: re-read the last pixel (not stored)
        const size_t e0 = ((size_t)n * a.hw + p) * a.c + q * 8;
        v[i] = *(const uint4*)(a.x + e0);
        if (a.residual) r[i] = *(const uint4*)(a.residual + e0);
    }
#pragma unroll
    for (int i = 0; i < GP; ++i) {
        const int p = p0 + i * rows;
        const size_t e0 = ((size_t)n * a.hw + min(p, a.hw - 1)) * a.c + q * 8;
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float y[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xv = (t & 1) ? el16_hi(w[t >> 1]) : el16_lo(w[t >> 1]);
            y[t] = fmaf(xv, A[t], C[t]);
        }
        act_drop<8>(y, (uint32_t)e0, row0, a.act, a.drop, key);
        if (a.residual) {
            const uint32_t rw[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int t = 0; t < 8; ++t) y[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
        }
        if (p < a.hw)
            *(uint4*)(a.out + e0) = make_uint4(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]), pack_el16x2(y[4], y[5]), pack_el16x2(y[6], y[7]));
    }
}

// gn_apply_walk_kernel with the statistics coming from the producing conv's epilogue (GnActArgs::part): the workgroup first
// finalises (mean, 1/std) of its sample's groups in LDS -- thread (value v = group / {sum, sum of squares}, slice) adds its share
// of the slots in index order in fp64, the slices are added in order -- so neither a statistics pass nor a finalize launch runs.
__global__ __launch_bounds__(256) void gn_apply_part_kernel(GnActArgs a) {
    __shared__ double red[256];
    __shared__ float2 mr_s[64];
    const int chunks = a.c >> 3, cpg = a.c / a.groups;
    const int n = blockIdx.y;
    const int q = threadIdx.x & (chunks - 1), row = threadIdx.x / chunks, rows = 256 / chunks;
    // the tensor loads go out first: the finalisation below (two dependent L2 round trips and three barriers) runs under them
    const int p0 = blockIdx.x * rows * GP + row;
    uint4 v[GP], r[GP];
#pragma unroll
    for (int i = 0; i < GP; ++i) {
        const int p = min(p0 + i * rows, a.hw - 1);  // past the end: re-read the last pixel (not stored)
        const size_t e0 = ((size_t)n * a.hw + p) * a.c + q * 8;
        v[i] = *(const uint4*)(a.x + e0);
        if (a.residual) r[i] = *(const uint4*)(a.residual + e0);
    }
    {
        const int nval = 2 * a.groups, opg = cpg >> 3;      // octets per group
        const int nsl = 256 / nval;                          // slices (groups <= 64 -> nsl >= 2)
        const int vi = threadIdx.x % nval, sl = threadIdx.x / nval;
        double acc = 0.0;
        if (sl < nsl) {
            const int g = vi >> 1, k = vi & 1;
            const float* p = a.part + (size_t)n * a.part_slots * chunks * 2;
            for (int s = sl; s < a.part_slots; s += nsl)
                for (int o = 0; o < opg; ++o) acc += (double)p[((size_t)s * chunks + g * opg + o) * 2 + k];
        }
        red[threadIdx.x] = acc;
        __syncthreads();
        if (threadIdx.x < nval) {
            double t = 0.0;
            for (int i = 0; i < nsl; ++i) t += red[i * nval + threadIdx.x];
            red[threadIdx.x] = t;
        }
        __syncthreads();
        if (threadIdx.x < a.groups) {
            const double inv = 1.0 / ((double)a.hw * cpg);
            const double mean = red[2 * threadIdx.x] * inv;
            const double var = red[2 * threadIdx.x + 1] * inv - mean * mean;
            mr_s[threadIdx.x] = make_float2((float)mean, rsqrtf(fmaxf((float)var, 0.0f) + 1e-5f));
        }
        __syncthreads();
    }
    const int g = (q * 8) / cpg;
    const float2 ms = mr_s[g];
    const float mean = ms.x, rstd = ms.y;
    const float4 g0 = *(const float4*)(a.gamma + q * 8), g1 = *(const float4*)(a.gamma + q * 8 + 4);
    const float4 b0 = *(const float4*)(a.beta + q * 8), b1 = *(const float4*)(a.beta + q * 8 + 4);
    float A[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float C[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        A[t] *= rstd;
        C[t] = fmaf(-mean, A[t], C[t]);
    }
    if (a.film_a) {
        const size_t fi = (size_t)n * a.film_stride + q * 8;
        const float4 fa0 = *(const float4*)(a.film_a + fi), fa1 = *(const float4*)(a.film_a + fi + 4);
        const float4 fc0 = *(const float4*)(a.film_c + fi), fc1 = *(const float4*)(a.film_c + fi + 4);
        const float fa[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
        const float fc[8] = {fc0.x, fc0.y, fc0.z, fc0.w, fc1.x, fc1.y, fc1.z, fc1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[t] *= fa[t];
            C[t] = fmaf(C[t], fa[t], fc[t]);
        }
    }
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
#pragma unroll
    for (int i = 0; i < GP; ++i) {
        const int p = p0 + i * rows;
        const size_t e0 = ((size_t)n * a.hw + min(p, a.hw - 1)) * a.c + q * 8;
        const uint32_t w[4] = {v[i].x, v[i].y, v[i].z, v[i].w};
        float y[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xv = (t & 1) ? el16_hi(w[t >> 1]) : el16_lo(w[t >> 1]);
            y[t] = fmaf(xv, A[t], C[t]);
        }
        act_drop<8>(y, (uint32_t)e0, row0, a.act, a.drop, key);
        if (a.residual) {
            const uint32_t rw[4] = {r[i].x, r[i].y, r[i].z, r[i].w};
#pragma unroll
            for (int t = 0; t < 8; ++t) y[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
        }
        if (p < a.hw)
            *(uint4*)(a.out + e0) = make_uint4(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]), pack_el16x2(y[4], y[5]), pack_el16x2(y[6], y[7]));
    }
}

// Small planes (a sample's tensor fits the registers of one 512- or 1024-lane workgroup: hw * c / 8 <= THREADS * MAXI chunks -- the 30x30x128
// and 15x15x256 levels of the OISST ResNet-UNet, whose convs do not produce statistics): ONE kernel, one read of the tensor from HBM.
// A lane keeps one 16-byte channel chunk column and, KEEP, MAXI pixels of it in registers (512 lanes x 16 chunks = 128 KB: the
// 15x15x256 level); larger samples (30x30x128 = 225 KB: 1024 lanes would have to hold it in 64 of their 128 registers, which
// spills) are walked twice, the second time out of L2.  The statistics are reduced like gn_stats_kernel's
// (lane sums in fp32, butterfly over the lanes of a group, one LDS slot per (wave, group), the waves added in order in fp64:
// deterministic), then normalise + FiLM + activation + dropout (+ residual) run from the registers.  Would replace gn_stats_kernel +
// gn_finalize_kernel + gn_apply_walk_kernel (5.2 + 4.9 + 11.9 us per GroupNorm at 100 rows of the OISST shapes).
// EXPERIMENT (DYF_GN_FUSED_SAMPLE=1), measured and not adopted: correct (tests/test_gpu_bench_forms.py and test_gpu_unet_resnet.py
// pass with it on) but no faster where the chip is full -- OISST rollout at 300 rows 3 740 / 3 791 fields/s with it, 3 797 without,
// 3 824 with DYF_GN_FUSED_REREAD=1 (both levels walked twice), same box: the three short launches it removes already overlap
// with the other row groups' kernels -- and slower where it is not: one workgroup per sample is 16 workgroups at 16 rows
// (595 against 730 fields/s).
template <int MAXI, int THREADS, bool KEEP>
__global__ __launch_bounds__(THREADS) void gn_fused_sample_kernel(GnActArgs a) {
    __shared__ float part[THREADS / 64][64][2];  // [wave][group]
    __shared__ float2 mr_s[64];
    const int chunks = a.c >> 3, cpg = a.c / a.groups, cq = cpg >> 3;
    const int n = blockIdx.x;
    const int q = threadIdx.x & (chunks - 1), row = threadIdx.x / chunks, rows = THREADS / chunks;
    const el16_t* xp = a.x + (size_t)n * a.hw * a.c + q * 8;
    float s = 0.0f, ss = 0.0f;
    auto add = [&](const uint4& u) {
        const uint32_t w[4] = {u.x, u.y, u.z, u.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const float lo = el16_lo(w[t]), hi = el16_hi(w[t]);
            s += lo + hi;
            ss = fmaf(lo, lo, fmaf(hi, hi, ss));
        }
    };
    uint4 v[KEEP ? MAXI : 1];
    if constexpr (KEEP) {
#pragma unroll
        for (int i = 0; i < MAXI; ++i) {
            const int p = row + i * rows;
            v[i] = p < a.hw ? *(const uint4*)(xp + (size_t)p * a.c) : make_uint4(0, 0, 0, 0);  // zeros add nothing to the sums
        }
#pragma unroll
        for (int i = 0; i < MAXI; ++i) add(v[i]);
    } else {
        int p = row;
        for (; p + 3 * rows < a.hw; p += 4 * rows) {  // four independent 16-B loads in flight
            const uint4 v0 = *(const uint4*)(xp + (size_t)p * a.c), v1 = *(const uint4*)(xp + (size_t)(p + rows) * a.c);
            const uint4 v2 = *(const uint4*)(xp + (size_t)(p + 2 * rows) * a.c), v3 = *(const uint4*)(xp + (size_t)(p + 3 * rows) * a.c);
            add(v0); add(v1); add(v2); add(v3);
        }
        for (; p < a.hw; p += rows) add(*(const uint4*)(xp + (size_t)p * a.c));
    }
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {  // lanes l and l ^ d hold the same group when d < cq or d >= chunks
        if (d < cq || d >= chunks) {
            s += __shfl_xor(s, d, 64);
            ss += __shfl_xor(ss, d, 64);
        }
    }
    const int l = threadIdx.x & 63;
    const int g = (q * 8) / cpg;
    if ((l & (cq - 1)) == 0 && (chunks >= 64 || l < chunks)) {  // exactly one lane of the wave holds group g
        part[threadIdx.x >> 6][g][0] = s;
        part[threadIdx.x >> 6][g][1] = ss;
    }
    __syncthreads();
    if (threadIdx.x < a.groups) {
        double ds = 0.0, dss = 0.0;
        // chunks > 64: a wave covers 64 of the chunk columns only and holds a slot for the groups that overlap them
        for (int w = 0; w < THREADS / 64; ++w) {
            const int q0 = (w * 64) & (chunks - 1), q1 = chunks >= 64 ? q0 + 64 : chunks;
            const int gq = threadIdx.x * cq;  // first chunk column of the group
            if (gq < q1 && gq + cq > q0) {
                ds += (double)part[w][threadIdx.x][0];
                dss += (double)part[w][threadIdx.x][1];
            }
        }
        const double inv = 1.0 / ((double)a.hw * cpg);
        const double mean = ds * inv;
        const double var = dss * inv - mean * mean;
        mr_s[threadIdx.x] = make_float2((float)mean, rsqrtf(fmaxf((float)var, 0.0f) + 1e-5f));
    }
    __syncthreads();
    const float2 ms = mr_s[g];
    const float mean = ms.x, rstd = ms.y;
    const float4 g0 = *(const float4*)(a.gamma + q * 8), g1 = *(const float4*)(a.gamma + q * 8 + 4);
    const float4 b0 = *(const float4*)(a.beta + q * 8), b1 = *(const float4*)(a.beta + q * 8 + 4);
    float A[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float C[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        A[t] *= rstd;
        C[t] = fmaf(-mean, A[t], C[t]);
    }
    if (a.film_a) {
        const size_t fi = (size_t)n * a.film_stride + q * 8;
        const float4 fa0 = *(const float4*)(a.film_a + fi), fa1 = *(const float4*)(a.film_a + fi + 4);
        const float4 fc0 = *(const float4*)(a.film_c + fi), fc1 = *(const float4*)(a.film_c + fi + 4);
        const float fa[8] = {fa0.x, fa0.y, fa0.z, fa0.w, fa1.x, fa1.y, fa1.z, fa1.w};
        const float fc[8] = {fc0.x, fc0.y, fc0.z, fc0.w, fc1.x, fc1.y, fc1.z, fc1.w};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            A[t] *= fa[t];
            C[t] = fmaf(C[t], fa[t], fc[t]);
        }
    }
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
    auto finish = [&](const uint4& xv4, const uint4& rv, int p) {  // p < hw
        const size_t e0 = ((size_t)n * a.hw + p) * a.c + q * 8;
        const uint32_t w[4] = {xv4.x, xv4.y, xv4.z, xv4.w};
        float y[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xv = (t & 1) ? el16_hi(w[t >> 1]) : el16_lo(w[t >> 1]);
            y[t] = fmaf(xv, A[t], C[t]);
        }
        act_drop<8>(y, (uint32_t)e0, row0, a.act, a.drop, key);
        if (a.residual) {
            const uint32_t rw[4] = {rv.x, rv.y, rv.z, rv.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) y[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
        }
        *(uint4*)(a.out + e0) = make_uint4(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]), pack_el16x2(y[4], y[5]), pack_el16x2(y[6], y[7]));
    };
    constexpr int RB = 4;  // loads in flight
    if constexpr (KEEP) {
#pragma unroll
        for (int i0 = 0; i0 < MAXI; i0 += RB) {
            uint4 r[RB];
            if (a.residual) {
#pragma unroll
                for (int j = 0; j < RB; ++j) {
                    const int p = min(row + (i0 + j) * rows, a.hw - 1);
                    r[j] = *(const uint4*)(a.residual + ((size_t)n * a.hw + p) * a.c + q * 8);
                }
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int p = row + (i0 + j) * rows;
                if (p < a.hw) finish(v[i0 + j], r[j], p);
            }
        }
    } else {
        // second walk over the sample: these 16-byte chunks were read by this workgroup a few microseconds ago (L2 hits)
        for (int pb = row; pb < a.hw; pb += RB * rows) {
            uint4 x4[RB], r[RB];
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int p = min(pb + j * rows, a.hw - 1);
                x4[j] = *(const uint4*)(xp + (size_t)p * a.c);
                if (a.residual) r[j] = *(const uint4*)(a.residual + ((size_t)n * a.hw + p) * a.c + q * 8);
            }
#pragma unroll
            for (int j = 0; j < RB; ++j) {
                const int p = pb + j * rows;
                if (p < a.hw) finish(x4[j], r[j], p);
            }
        }
    }
}

// Large planes (512^2: 2 048 slots per sample): the finalisation is too much to repeat in every workgroup of the apply pass --
// one workgroup per (sample, group) does it once (fixed order: 128 slot lanes x 2 sums, then a tree over the lanes) and
// gn_apply_walk_kernel runs.  (First form: one workgroup per sample, 16 slot lanes per sum: 27 us at 512^2, 6 % of that rollout.)
__global__ __launch_bounds__(256) void gn_finalize_part_kernel(const float* part, int part_slots, int c, int groups, int hw, float2* mr) {
    __shared__ double red[256];
    const int chunks = c >> 3, cpg = c / groups, opg = cpg >> 3;
    const int g = blockIdx.x, n = blockIdx.y;
    const int k = threadIdx.x & 1, sl = threadIdx.x >> 1;
    const float* p = part + (size_t)n * part_slots * chunks * 2 + (size_t)g * opg * 2 + k;
    double acc = 0.0;
    for (int s = sl; s < part_slots; s += 128)
        for (int o = 0; o < opg; ++o) acc += (double)p[((size_t)s * chunks + o) * 2];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int stride = 64; stride >= 1; stride >>= 1) {
        if (sl < stride) red[threadIdx.x] += red[threadIdx.x + 2 * stride];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double inv = 1.0 / ((double)hw * cpg);
        const double mean = red[0] * inv;
        const double var = red[1] * inv - mean * mean;
        mr[(size_t)n * groups + g] = make_float2((float)mean, rsqrtf(fmaxf((float)var, 0.0f) + 1e-5f));
    }
}

bool gn_part_supported(int c, int groups) {  // shapes gn_apply_part_kernel takes (the caller asks before requesting conv statistics)
    const int chunks = c >> 3;
    return c % 8 == 0 && groups >= 1 && c % groups == 0 && (c / groups) % 8 == 0 && groups <= 64 && (chunks & (chunks - 1)) == 0 && chunks <= 256;
}

// forward-epoch word of the fused GroupNorm convs (gn_fused.h): the granule tags of a forward are built from it, so it must change
// on every forward -- also on hipGraph replays, where kernel arguments are frozen but memory is not
__global__ void gn_epoch_bump_kernel(uint32_t* epoch) {
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        uint32_t v = (*epoch + 1u) & 0x00ffffffu;
        *epoch = v ? v : 1u;  // never 0: a zeroed granule buffer must not match
    }
}

hipError_t launch_gn_epoch_bump(uint32_t* epoch, hipStream_t s) {
    hipLaunchKernelGGL(gn_epoch_bump_kernel, dim3(1), dim3(64), 0, s, epoch);
    return hipGetLastError();
}

hipError_t launch_gn_act(const GnActArgs& a, hipStream_t s) {
    const int cpg = a.c / a.groups;
    if (a.part && a.part_slots > 0 && gn_part_supported(a.c, a.groups) && a.n <= 65535) {
        const int chunks = a.c >> 3, rows = 256 / chunks;
        const unsigned bx = (unsigned)((a.hw + rows * GP - 1) / (rows * GP));
        if ((long long)a.part_slots * chunks * 2 > 1024 && a.stats) {  // > 4 KB of partials per sample: finalise once per sample
            float2* mr = (float2*)(a.stats + (size_t)a.n * a.groups * 2 * GN_MAX_BLOCKS);
            dyf_form_note("gn_finalize_part_kernel+gn_apply", a.n);
            hipLaunchKernelGGL(gn_finalize_part_kernel, dim3(a.groups, a.n), dim3(256), 0, s, a.part, a.part_slots, a.c, a.groups, a.hw, mr);
            KernelProf kp("gn_apply_walk_kernel", s, (double)a.n * a.hw * a.c * 2.0 * (a.residual ? 3.0 : 2.0));
            hipLaunchKernelGGL(gn_apply_walk_kernel, dim3(bx, a.n), dim3(256), 0, s, a, (const float2*)mr);
            return hipGetLastError();
        }
        dyf_form_note("gn_apply_part_kernel", a.n);
        KernelProf kp("gn_apply_part_kernel", s, (double)a.n * a.hw * a.c * 2.0 * (a.residual ? 3.0 : 2.0));
        hipLaunchKernelGGL(gn_apply_part_kernel, dim3(bx, a.n), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    if (a.stats && (a.c % 8 == 0) && (cpg % 8 == 0) && a.groups <= 64) {
        {
            const int chunks = a.c >> 3, cq = cpg >> 3;
            const bool fused = dyf_form("DYF_GN_FUSED_SAMPLE") && atoi(dyf_form("DYF_GN_FUSED_SAMPLE")) != 0;  // experiment, off
            const long long per = (long long)a.hw * chunks;
            const int reread = dyf_form("DYF_GN_FUSED_REREAD") ? atoi(dyf_form("DYF_GN_FUSED_REREAD")) : 0;  // experiment: 1 = never keep
            if (fused && (chunks & (chunks - 1)) == 0 && (cq & (cq - 1)) == 0 && chunks <= 256 && per <= 1024 * 32) {
                dyf_form_note("gn_fused_sample_kernel", a.n);
                if (per <= 512 * 8 && !reread) hipLaunchKernelGGL((gn_fused_sample_kernel<8, 512, true>), dim3(a.n), dim3(512), 0, s, a);
                else if (per <= 512 * 16 && !reread) hipLaunchKernelGGL((gn_fused_sample_kernel<16, 512, true>), dim3(a.n), dim3(512), 0, s, a);
                else hipLaunchKernelGGL((gn_fused_sample_kernel<1, 1024, false>), dim3(a.n), dim3(1024), 0, s, a);
                return hipGetLastError();
            }
        }
        dyf_form_note("gn_stats_kernel+gn_apply", a.n);
        const long long per_sample = (long long)a.hw * (a.c >> 3);
        // >= 4 passes of 256 lanes per workgroup, at most GN_MAX_BLOCKS workgroups per sample (their partials are added in order)
        const unsigned bx = (unsigned)std::max<long long>(1, std::min<long long>((per_sample + 1023) / 1024, GN_MAX_BLOCKS));
        hipLaunchKernelGGL(gn_stats_kernel, dim3(bx, a.n), dim3(256), 0, s, a.x, a.hw, a.c, a.groups, a.stats);
        const long long total = (long long)a.n * per_sample;
        // the scratch holds [n][groups][2][GN_MAX_BLOCKS] doubles + [n][groups] float2 (gn_stats_doubles())
        float2* mr = (float2*)(a.stats + (size_t)a.n * a.groups * 2 * GN_MAX_BLOCKS);
        const int cnt = a.n * a.groups;
        hipLaunchKernelGGL(gn_finalize_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, (const double*)a.stats, cnt, (int)bx,
                           1.0 / ((double)a.hw * cpg), mr);
        const int chunks = a.c >> 3;
        const bool walk = !(dyf_form("DYF_GN_WALK") && atoi(dyf_form("DYF_GN_WALK")) == 0);
        if (walk && (chunks & (chunks - 1)) == 0 && chunks <= 256 && a.n <= 65535) {
            const int rows = 256 / chunks;
            KernelProf kp("gn_apply_walk_kernel", s, (double)a.n * a.hw * a.c * 2.0 * (a.residual ? 3.0 : 2.0));
            hipLaunchKernelGGL(gn_apply_walk_kernel, dim3((unsigned)((a.hw + rows * GP - 1) / (rows * GP)), a.n), dim3(256), 0, s, a, (const float2*)mr);
        } else {
            hipLaunchKernelGGL(gn_apply_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, (const float2*)mr);
        }
        return hipGetLastError();
    }
    hipLaunchKernelGGL(gn_act_kernel, dim3(a.n * a.groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ channel LayerNorm
// 8 lanes per pixel (each a strided slice of the channels), butterfly over the 8 lanes.
__global__ __launch_bounds__(256) void layernorm_c_kernel(LayerNormArgs a) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    long long pix = gid >> 3;
    const int sl = (int)(gid & 7);
    const bool live = pix < a.pixels;
    if (!live) pix = a.pixels - 1;
    const el16_t* x = a.x + (size_t)pix * a.c;
    float s = 0.0f;
    for (int c = sl; c < a.c; c += 8) s += el16_to_f32(x[c]);
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)a.c;
    float v = 0.0f;
    for (int c = sl; c < a.c; c += 8) {
        const float d = el16_to_f32(x[c]) - mean;
        v = fmaf(d, d, v);
    }
#pragma unroll
    for (int off = 4; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    const float rstd = rsqrtf(v / (float)a.c + 1e-5f);
    if (!live) return;
    const int n = (int)(pix / a.hw);
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
    for (int c = sl; c < a.c; c += 8) {
        const size_t e = (size_t)pix * a.c + c;
        float y = (el16_to_f32(x[c]) - mean) * rstd * a.g[c];
        y = drop_apply(y, (uint32_t)e, row0, a.drop, key);
        a.out[e] = f32_to_el16(y);
    }
}

// c % 8 == 0, c/8 a power of two <= 64: c/8 lanes per pixel, each ONE 16-byte chunk kept in registers for both passes
// (one 16-B load, one 16-B store per lane instead of c/8 two-byte loads three times)
__global__ __launch_bounds__(256) void layernorm_c_vec_kernel(LayerNormArgs a, int chunks) {
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const int q = (int)(gid & (chunks - 1));
    long long pix = gid / chunks;
    const bool live = pix < a.pixels;
    if (!live) pix = a.pixels - 1;
    const size_t e0 = (size_t)pix * a.c + q * 8;
    const uint4 v = *(const uint4*)(a.x + e0);
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    float x[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) x[t] = (t & 1) ? el16_hi(w[t >> 1]) : el16_lo(w[t >> 1]);
    float s = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) s += x[t];
    for (int off = 1; off < chunks; off <<= 1) s += __shfl_xor(s, off, 64);
    const float mean = s / (float)a.c;
    float var = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
        x[t] -= mean;
        var = fmaf(x[t], x[t], var);
    }
    for (int off = 1; off < chunks; off <<= 1) var += __shfl_xor(var, off, 64);
    const float rstd = rsqrtf(var / (float)a.c + 1e-5f);
    if (!live) return;
    const float4 g0 = *(const float4*)(a.g + q * 8), g1 = *(const float4*)(a.g + q * 8 + 4);
    const float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
    float y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) y[t] = x[t] * rstd * g[t];
    const int n = (int)(pix / a.hw);
    act_drop<8>(y, (uint32_t)e0, (uint32_t)((size_t)n * a.hw * a.c), ACT_NONE, a.drop, drop_row_key(a.drop, n));
    *(uint4*)(a.out + e0) = make_uint4(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]), pack_el16x2(y[4], y[5]),
                                       pack_el16x2(y[6], y[7]));
}

hipError_t launch_layernorm_c(const LayerNormArgs& a, hipStream_t s) {
    const int chunks = a.c >> 3;
    if ((a.c & 7) == 0 && chunks >= 1 && chunks <= 64 && (chunks & (chunks - 1)) == 0) {
        const long long threads = a.pixels * chunks;
        KernelProf kp("layernorm_c_vec_kernel", s, (double)a.pixels * a.c * 2.0 * 2.0);
        hipLaunchKernelGGL(layernorm_c_vec_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a, chunks);
        return hipGetLastError();
    }
    const long long threads = a.pixels * 8;
    hipLaunchKernelGGL(layernorm_c_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ LinearAttention
// One workgroup per (sample, head).  q: softmax over the 32 head channels, times 32^-1/2; k: softmax over the pixels;
// v / N; context[d][e] = sum_n k[d][n] v[e][n]; out[e][n] = sum_d context[d][e] q[d][n].
__global__ __launch_bounds__(256) void linear_attention_kernel(LinAttnArgs a) {
    __shared__ float red[8][32];      // per-wave partials
    __shared__ float kmax[32], ksum[32];
    __shared__ float ctx[32][33];
    __shared__ float tile_k[64][33], tile_v[64][33];
    const int n = blockIdx.x / a.heads, h = blockIdx.x % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32;
    const el16_t* base = a.qkv + (size_t)n * a.hw * C3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int d = tid & 31, grp = tid >> 5;  // 8 pixel-groups x 32 channels
    // ---- pass 1: max over pixels of k[d][.]
    float m = -3.0e38f;
    for (int p = grp; p < a.hw; p += 8) m = fmaxf(m, el16_to_f32(base[(size_t)p * C3 + hd + h * 32 + d]));
    red[grp][d] = m;
    __syncthreads();
    if (tid < 32) {
        float t = red[0][tid];
        for (int i = 1; i < 8; ++i) t = fmaxf(t, red[i][tid]);
        kmax[tid] = t;
    }
    __syncthreads();
    // ---- pass 2: sum over pixels of exp(k - max)
    float sacc = 0.0f;
    const float km = kmax[d];
    for (int p = grp; p < a.hw; p += 8) sacc += __expf(el16_to_f32(base[(size_t)p * C3 + hd + h * 32 + d]) - km);
    __syncthreads();
    red[grp][d] = sacc;
    __syncthreads();
    if (tid < 32) {
        float t = 0.0f;
        for (int i = 0; i < 8; ++i) t += red[i][tid];
        ksum[tid] = t;
    }
    __syncthreads();
    // ---- pass 3: context[d][e], thread (dd, e4): 4 entries each; pixels staged 64 at a time
    const int dd = tid >> 3, e0 = (tid & 7) * 4;
    float c4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int p0 = 0; p0 < a.hw; p0 += 64) {
        __syncthreads();
        for (int i = tid; i < 64 * 32; i += 256) {
            const int pp = i >> 5, ch = i & 31;
            const int p = p0 + pp;
            float kv = 0.0f, vv = 0.0f;
            if (p < a.hw) {
                kv = __expf(el16_to_f32(base[(size_t)p * C3 + hd + h * 32 + ch]) - kmax[ch]) / ksum[ch];
                vv = el16_to_f32(base[(size_t)p * C3 + 2 * hd + h * 32 + ch]);
            }
            tile_k[pp][ch] = kv;
            tile_v[pp][ch] = vv;
        }
        __syncthreads();
        for (int pp = 0; pp < 64; ++pp) {
            const float kv = tile_k[pp][dd];
#pragma unroll
            for (int t = 0; t < 4; ++t) c4[t] = fmaf(kv, tile_v[pp][e0 + t], c4[t]);
        }
    }
    const float inv_n = 1.0f / (float)a.hw;
#pragma unroll
    for (int t = 0; t < 4; ++t) ctx[dd][e0 + t] = c4[t] * inv_n;
    __syncthreads();
    // ---- pass 4: per pixel softmax_d(q) * scale, then out[e] = sum_d ctx[d][e] q[d]
    const float scale = 0.17677669529663687f;  // 32^-1/2
    for (int p = tid; p < a.hw; p += 256) {
        float q[32];
        float qm = -3.0e38f;
        const el16_t* qp = base + (size_t)p * C3 + h * 32;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            q[i] = el16_to_f32(qp[i]);
            qm = fmaxf(qm, q[i]);
        }
        float qs = 0.0f;
#pragma unroll
        for (int i = 0; i < 32; ++i) {
            q[i] = __expf(q[i] - qm);
            qs += q[i];
        }
        const float qn = scale / qs;
        el16_t* op = a.out + ((size_t)n * a.hw + p) * hd + h * 32;
        for (int e = 0; e < 32; ++e) {
            float o = 0.0f;
#pragma unroll
            for (int i = 0; i < 32; ++i) o = fmaf(ctx[i][e], q[i], o);
            op[e] = f32_to_el16(o * qn);
        }
    }
    (void)lane; (void)wave;
}

// Pixel-parallel MFMA form (any image size).  Both contractions of attention.py:28-49 are GEMMs with a 32 x 32 output
// per (sample, head); they are HBM-bound (16 flop per byte), so the work is arranged around the memory accesses:
//   (1) context: ctx[d][e] = sum_p softmax_p(k)[p][d] * v[p][e] / (h*w).  The contraction runs over PIXELS while memory is
//       channel-contiguous; MFMA wants 8 contraction values per lane.  Instead of transposing, lane (c, hi) loads its 8
//       pixels of channel c as 2-byte loads (32 lanes = 64 contiguous bytes per pixel) -- the k <-> pixel mapping of a
//       sum is free as long as both operands share it.  One pass over k: a wave keeps its 256 pixels of k in registers,
//       takes its local per-channel max, exponentiates, and feeds v_mfma_f32_32x32x16_bf16 (k' rounded to bf16: a
//       2^-9 relative error per term, averaged over the pixel sum); waves and workgroups are merged by the usual (max, sum, acc) rescaling.
//   (2) out[p][e] = sum_d ctx[d][e] * softmax_d(q)[p][d] * scale: pixels are MFMA columns, so q loads are plain 16-byte
//       loads, the softmax over d needs one exchange between lanes p and p+32, and the result is stored as 16-byte rows.
constexpr int LA_HEADS = 4;        // LinearAttention(dim) always has 4 heads of 32 channels (unet.py)
constexpr int LA_PIX = 1024;       // pixels per workgroup of the context kernel (256 per wave)
constexpr int LA_PART = 1024 + 64; // floats of one partial: acc[16][64], max[32], sum[32]
typedef __attribute__((ext_vector_type(16))) float la_f32x16;

__device__ __forceinline__ el16x8_t la_frag(const uint32_t (&w)[4]) {
    const uint4 v = make_uint4(w[0], w[1], w[2], w[3]);
    return __builtin_bit_cast(el16x8_t, v);
}
// x -> bf16 hi part (round to nearest) and bf16 lo part of the remainder
__device__ __forceinline__ void la_split(float x0, float x1, uint32_t& hi, uint32_t& lo) {
    hi = pack_el16x2(x0, x1);
    lo = pack_el16x2(x0 - el16_lo(hi), x1 - el16_hi(hi));
}

__global__ __launch_bounds__(256) void linattn_ctx_mfma_kernel(LinAttnArgs a, float* part, int nblk) {
    __shared__ float sm_m[4][32], sm_s[4][32];
    __shared__ float sm_acc[4][16][64];
    const int bh = blockIdx.y, n = bh / LA_HEADS, h = bh % LA_HEADS;
    constexpr int C3 = 3 * LA_HEADS * 32, hd = LA_HEADS * 32;  // compile-time pitch: the 256 load offsets are immediates
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), c = lane & 31, hi = lane >> 5;
    const int pw = blockIdx.x * LA_PIX + wave * 256;  // first pixel of this wave
    la_f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    float m = -3.0e38f, s = 0.0f;
    // buffer loads: voffset = this lane's constant (pixel hi*8, channel c), the pixel walk goes through the scalar offset.
    // The tail wave of a (sample, head) puts the walk into voffset instead, with the buffer ending at the last pixel of the
    // sample (the range check covers voffset only): pixels past it read 0 and are masked (k: a very negative value).
    const unsigned lane_off = (unsigned)(hi * 8 * C3 + c) * 2u;
    const el16_t* kw = a.qkv + ((size_t)n * a.hw + pw) * C3 + hd + h * 32;  // wave-uniform
    auto wave_body = [&](auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;
        const int left = a.hw - pw;  // valid pixels of this wave (tail: 1..255)
        const int k_bytes = FULL ? 256 * C3 * 2 : left * C3 * 2 - (hd + h * 32) * 2;
        const auto rs_k = __builtin_amdgcn_make_buffer_rsrc((void*)kw, 0, k_bytes, 0x00020000);
        const auto rs_v = __builtin_amdgcn_make_buffer_rsrc((void*)(kw + hd), 0, k_bytes - (FULL ? 0 : hd * 2), 0x00020000);
        auto ld = [&](const auto& rs, int rel, uint32_t fill) -> uint32_t {  // pixel pw + hi*8 + rel, channel c
            if (FULL) return (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, lane_off, rel * C3 * 2, 0);
            const uint32_t v = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs, lane_off + (unsigned)(rel * C3 * 2), 0, 0);
            return hi * 8 + rel < left ? v : fill;
        };
        constexpr uint32_t NEG = 0xC2C8u;  // bf16(-100) for pixels that do not exist; m >= the real maximum keeps exp(.) <= 1
        // pass 1: this lane's 128 values of k (channel c, pixels with (p >> 3) & 1 == hi), packed two per register
        uint32_t kp[64];
#pragma unroll
        for (int t = 0; t < 16; ++t)
#pragma unroll
            for (int jj = 0; jj < 4; ++jj)
                kp[t * 4 + jj] = ld(rs_k, t * 16 + 2 * jj, NEG) | (ld(rs_k, t * 16 + 2 * jj + 1, NEG) << 16);
        // v is prefetched PF steps ahead; the first steps fly while the max is taken
        constexpr int PF = 6;
        uint32_t vq[8][8];
        auto issue_v = [&](int t) {
#pragma unroll
            for (int j = 0; j < 8; ++j) vq[t & 7][j] = ld(rs_v, t * 16 + j, 0u);
            __builtin_amdgcn_sched_barrier(0);  // keep the loads here: sunk to their uses they would serialise the steps
        };
#pragma unroll
        for (int t = 0; t < PF; ++t) issue_v(t);
#pragma unroll
        for (int i = 0; i < 64; ++i) m = fmaxf(m, fmaxf(el16_lo(kp[i]), el16_hi(kp[i])));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        // pass 2: k' = exp(k - m) against v, 16 pixels per MFMA step
#pragma unroll
        for (int t = 0; t < 16; ++t) {
            if (t + PF < 16) issue_v(t + PF);
            uint32_t vf[4], kh[4];
#pragma unroll
            for (int jj = 0; jj < 4; ++jj) {
                vf[jj] = vq[t & 7][2 * jj] | (vq[t & 7][2 * jj + 1] << 16);
                float x0 = __expf(el16_lo(kp[t * 4 + jj]) - m);
                float x1 = __expf(el16_hi(kp[t * 4 + jj]) - m);
                if (!FULL) {  // exp(-100 - m) is not 0 when m is very negative
                    x0 = hi * 8 + t * 16 + 2 * jj < left ? x0 : 0.0f;
                    x1 = hi * 8 + t * 16 + 2 * jj + 1 < left ? x1 : 0.0f;
                }
                s += x0 + x1;
                kh[jj] = pack_el16x2(x0, x1);
            }
            // D[e][d] += v[p][e] * k'[p][d]: rows e (A = v), columns d = this lane's channel (B = k')
            acc = DYF_MFMA_32x32x16(la_frag(vf), la_frag(kh), acc, 0, 0, 0);
        }
        s += __shfl_xor(s, 32, 64);
    };
    if (pw + 256 <= a.hw) wave_body(std::true_type{});
    else if (pw < a.hw) wave_body(std::false_type{});
    // merge the four waves: column d = c of every accumulator register is rescaled by exp(m_w[d] - M[d])
    if (hi == 0) {
        sm_m[wave][c] = m;
        sm_s[wave][c] = s;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) sm_acc[wave][r][lane] = acc[r];
    __syncthreads();
    float M = sm_m[0][c];
#pragma unroll
    for (int w = 1; w < 4; ++w) M = fmaxf(M, sm_m[w][c]);
    float f[4], S = 0.0f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        f[w] = __expf(sm_m[w][c] - M);  // idle waves: m = -3e38 -> 0
        S = fmaf(f[w], sm_s[w][c], S);
    }
    float* o = part + ((size_t)bh * nblk + blockIdx.x) * LA_PART;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int r = wave * 4 + r4;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) v = fmaf(f[w], sm_acc[w][r][lane], v);
        o[r * 64 + lane] = v;
    }
    if (tid < 32) {
        o[1024 + tid] = M;
        o[1056 + tid] = S;
    }
}

// merge the workgroup partials of one (sample, head), normalise, and emit ctx as the A fragments of the output product:
// frags[bh][hi/lo part][k-step s][lane (e, hi')][8]: slot j of lane (e, hi') = ctx[d = 16 s + 8 hi' + j][e]
// dmap 0: slot j of lane (e, hi') = ctx[d = 16 s + 8 hi' + j][e] (linattn_out_mfma_kernel: q' fragments loaded from memory);
// dmap 1: d = 16 s + 8 (j >> 2) + 4 hi' + (j & 3) (linattn_fused_out_kernel: q' fragments are MFMA accumulator registers)
__global__ __launch_bounds__(256) void linattn_merge_kernel(const float* part, int nblk, float inv_n, el16_t* frags, int dmap) {
    // gridDim.y workgroups per (sample, head), each owning 16 / gridDim.y of the 16 row blocks r (64 ctx entries each) -- round 4: at
    // 512^2 with 4 rows the launch was 16 workgroups on a 256-CU chip reading 1.1 MB each (46 us, 12 launches per forward).  At
    // 512^2 there are 256 partials per head: the serial three-pass form (max, sum, weighted accumulation, each a chain of dependent
    // loads) took 225 us; here the max / sum passes run 8 partial-groups wide, the rescaling factors exp(M_b - M) are computed once
    // into LDS (by every workgroup of the head: 8 K exponentials), and the accumulation keeps 4 loads in flight.  The result does
    // not depend on gridDim.y (same per-entry summation order).
    __shared__ float wexp[256][32];   // [partial][column d]; nblk <= 256 per pass (looped otherwise)
    __shared__ float red[8][32];
    const int bh = blockIdx.x, tid = threadIdx.x, lane = tid & 63, rq = tid >> 6, d = lane & 31, hi = lane >> 5;
    const int d8 = tid & 31, bq = tid >> 5;
    const float* pb = part + (size_t)bh * nblk * LA_PART;
    float m = -3.0e38f;
    for (int b = bq; b < nblk; b += 8) m = fmaxf(m, pb[(size_t)b * LA_PART + 1024 + d8]);
    red[bq][d8] = m;
    __syncthreads();
    float M = red[0][d];
#pragma unroll
    for (int q = 1; q < 8; ++q) M = fmaxf(M, red[q][d]);
    const float M8 = fmaxf(fmaxf(fmaxf(red[0][d8], red[1][d8]), fmaxf(red[2][d8], red[3][d8])),
                           fmaxf(fmaxf(red[4][d8], red[5][d8]), fmaxf(red[6][d8], red[7][d8])));
    __syncthreads();
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float ssum = 0.0f;
    for (int b0 = 0; b0 < nblk; b0 += 256) {
        const int nb = min(256, nblk - b0);
        for (int b = bq; b < nb; b += 8) {
            const float dm = pb[(size_t)(b0 + b) * LA_PART + 1024 + d8] - M8;
            const float w = dmap ? __builtin_amdgcn_exp2f(dm) : __expf(dm);  // the fused form keeps its maxima in log2 units
            wexp[b][d8] = w;
            ssum = fmaf(pb[(size_t)(b0 + b) * LA_PART + 1056 + d8], w, ssum);
        }
        __syncthreads();
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = rq * 4 + r4;
            if ((r4 % (int)gridDim.y) != (int)blockIdx.y) continue;  // this workgroup's share of the wave's four row blocks
            const float* pr = pb + (size_t)b0 * LA_PART + r * 64 + lane;
            float v0 = 0.0f, v1 = 0.0f, v2 = 0.0f, v3 = 0.0f;
            int b = 0;
            for (; b + 3 < nb; b += 4) {
                v0 = fmaf(pr[(size_t)b * LA_PART], wexp[b][d], v0);
                v1 = fmaf(pr[(size_t)(b + 1) * LA_PART], wexp[b + 1][d], v1);
                v2 = fmaf(pr[(size_t)(b + 2) * LA_PART], wexp[b + 2][d], v2);
                v3 = fmaf(pr[(size_t)(b + 3) * LA_PART], wexp[b + 3][d], v3);
            }
            for (; b < nb; ++b) v0 = fmaf(pr[(size_t)b * LA_PART], wexp[b][d], v0);
            acc[r4] += (v0 + v1) + (v2 + v3);
        }
        __syncthreads();
    }
    red[bq][d8] = ssum;
    __syncthreads();
    float S = 0.0f;
#pragma unroll
    for (int q = 0; q < 8; ++q) S += red[q][d];
    const float norm = inv_n / S;  // v / (h*w)  (attention.py:41)
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
        const int r = rq * 4 + r4;
        if ((r4 % (int)gridDim.y) != (int)blockIdx.y) continue;
        const float v = acc[r4] * norm;
        const int e = (r >> 2) * 8 + hi * 4 + (r & 3);
        const int fh = dmap ? (d >> 2) & 1 : (d >> 3) & 1, fj = dmap ? ((d >> 3) & 1) * 4 + (d & 3) : d & 7;
        const size_t idx = (((size_t)bh * 2 * 2 + (d >> 4)) * 64 + fh * 32 + e) * 8 + fj;
        const el16_t vh = f32_to_el16(v);
        frags[idx] = vh;
        frags[idx + 2 * 64 * 8] = f32_to_el16(v - el16_to_f32(vh));
    }
}

__global__ __launch_bounds__(256) void linattn_out_mfma_kernel(LinAttnArgs a, const el16_t* frags) {
    const int bh = blockIdx.y, n = bh / a.heads, h = bh % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    el16x8_t ah[2], al[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
        ah[s] = *(const el16x8_t*)(frags + (((size_t)bh * 4 + s) * 64 + lane) * 8);
        al[s] = *(const el16x8_t*)(frags + (((size_t)bh * 4 + 2 + s) * 64 + lane) * 8);
    }
    const float scale = 0.17677669529663687f;  // 32^-1/2
#pragma unroll 1
    for (int g = 0; g < 4; ++g) {
        const int p = blockIdx.x * 512 + wave * 128 + g * 32 + l31;
        if (__builtin_amdgcn_readfirstlane(p - l31) >= a.hw) break;
        const bool valid = p < a.hw;
        const el16_t* qp = a.qkv + ((size_t)n * a.hw + (valid ? p : a.hw - 1)) * C3 + h * 32 + hi * 8;
        const uint4 x[2] = {*(const uint4*)qp, *(const uint4*)(qp + 16)};  // channels 8 hi + {0..7} and 16 + 8 hi + {0..7}
        float q[16];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const uint32_t w[4] = {x[s].x, x[s].y, x[s].z, x[s].w};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                q[s * 8 + 2 * t] = el16_lo(w[t]);
                q[s * 8 + 2 * t + 1] = el16_hi(w[t]);
            }
        }
        float qm = q[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) qm = fmaxf(qm, q[i]);
        qm = fmaxf(qm, __shfl_xor(qm, 32, 64));
        float qs = 0.0f;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            q[i] = __expf(q[i] - qm);
            qs += q[i];
        }
        qs += __shfl_xor(qs, 32, 64);
        const float qn = scale / qs;
        la_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            uint32_t qh[4], ql[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) la_split(q[s * 8 + 2 * t], q[s * 8 + 2 * t + 1], qh[t], ql[t]);
            // D[e][p] += ctx[d][e] * q'[p][d]; hi*hi + lo*hi + hi*lo keeps the product fp32-accurate
            acc = DYF_MFMA_32x32x16(ah[s], la_frag(qh), acc, 0, 0, 0);
            acc = DYF_MFMA_32x32x16(al[s], la_frag(qh), acc, 0, 0, 0);
            acc = DYF_MFMA_32x32x16(ah[s], la_frag(ql), acc, 0, 0, 0);
        }
        // lane (p, hi) holds rows e = 8 (r >> 2) + 4 hi + (r & 3); register groups 2 g2 / 2 g2 + 1 are exchanged between lanes
        // p and p + 32 so that every lane stores 8 consecutive channels
        el16_t* op = a.out + ((size_t)n * a.hw + p) * hd + h * 32 + hi * 8;
#pragma unroll
        for (int g2 = 0; g2 < 2; ++g2) {
            const uint32_t p0 = pack_el16x2(acc[g2 * 8 + 0] * qn, acc[g2 * 8 + 1] * qn), p1 = pack_el16x2(acc[g2 * 8 + 2] * qn, acc[g2 * 8 + 3] * qn);
            const uint32_t q0 = pack_el16x2(acc[g2 * 8 + 4] * qn, acc[g2 * 8 + 5] * qn), q1 = pack_el16x2(acc[g2 * 8 + 6] * qn, acc[g2 * 8 + 7] * qn);
            const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
            const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
            uint4 o;
            o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
            if (valid) *(uint4*)(op + g2 * 16) = o;
        }
    }
}

hipError_t launch_linear_attention(const LinAttnArgs& a, hipStream_t s) {
    if (a.scratch && a.heads == LA_HEADS) {
        const int nblk = (a.hw + LA_PIX - 1) / LA_PIX;
        const int BH = a.n * a.heads;
        float* part = a.scratch;                                          // [BH][nblk][LA_PART]
        el16_t* frags = (el16_t*)(a.scratch + (size_t)BH * nblk * LA_PART);  // [BH][2][2][64][8]
        hipLaunchKernelGGL(linattn_ctx_mfma_kernel, dim3(nblk, BH), dim3(256), 0, s, a, part, nblk);
        hipLaunchKernelGGL(linattn_merge_kernel, dim3(BH, (nblk >= 32 && BH < 128) ? 4 : 1), dim3(256), 0, s, (const float*)part, nblk, 1.0f / (float)a.hw, frags, 0);
        hipLaunchKernelGGL(linattn_out_mfma_kernel, dim3((a.hw + 511) / 512, BH), dim3(256), 0, s, a, (const el16_t*)frags);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(linear_attention_kernel, dim3(a.n * a.heads), dim3(256), 0, s, a);
    return hipGetLastError();
}


// ---- fused form: to_qkv, both contractions and to_out (+ bias + residual) in two passes over the normalised input ----
// The qkv tensor (3 * 128 channels per pixel: 829 MB at 300 x 60 x 60) and the attention output (128 channels) never exist:
//   pass 1 (context):  per 32-pixel group, k_h and v_h = W x on the matrix cores (x fragment = one 16-byte load per lane and
//       k-step: rows are pixels); the accumulators come out as lane (channel, hi) x 16 pixels, which IS the operand layout of
//       the pixel contraction ctx_h[e][d] += v[p][e] k'[p][d] (any pixel <-> k-slot mapping works as long as both operands
//       share it); the softmax over pixels is online (running max per column d, accumulators rescaled only when a group
//       raises it); the workgroup's (max, sum, acc) partial goes to the same merge kernel as the unfused form;
//   pass 2 (output):  q_h = W_q x with pixels as MFMA columns, so a lane (pixel, hi) holds 16 of the head's 32 channels:
//       softmax over d with one lane exchange; out_h = ctx_h^T q' (hi/lo split operands, fp32-accurate); the bf16 result
//       registers are the B operand of to_out (k = head channel), whose accumulators get bias + residual and are stored as
//       16-byte channel rows.
// HBM: xn read twice, residual read once, y written once (4 x 138 MB at OISST level 0, against 2.5 GB for the unfused chain).
__device__ __forceinline__ float lf_max32(float x) {  // max over lanes l and l ^ 32 (one VALU op instead of an LDS round trip)
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
}
__device__ __forceinline__ float lf_sum32(float x) {
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(x), __float_as_uint(x), false, false);
    return __uint_as_float(sw[0]) + __uint_as_float(sw[1]);
}
constexpr int LF_GROUPS = 32;  // 32-pixel groups per workgroup of the large-batch form (= LA_PIX pixels); the kernels take the count as an
                               // argument (`gpb`: 32 / 16 / 8, launch_linear_attention_fused)

template <int C>
struct LfCfg {
    static constexpr int KS = C / 16;
    static constexpr int CTX_W = 8 * KS * 1024;                   // k0..3, v0..3 fragments
    static constexpr int CTX_LDS = CTX_W + 4 * 16 * 64 * 4 + 2 * 4 * 32 * 4;
    static constexpr int OUT_WQ = 4 * KS * 1024, OUT_WO = (C / 32) * 8 * 1024, OUT_CF = 4 * 4 * 1024;
    static constexpr int OUT_LDS = OUT_WQ + OUT_WO + OUT_CF + C * 4;
};

template <int C>
__global__ __launch_bounds__(256, 2) void linattn_fused_ctx_kernel(const el16_t* __restrict__ xn, int hw, const el16_t* __restrict__ wfrag,
                                                                float* __restrict__ part, int nblk, int gpb) {
    constexpr int KS = LfCfg<C>::KS;
    extern __shared__ __attribute__((aligned(16))) unsigned char lf_smem[];
    uint4* wl = (uint4*)lf_smem;                                  // [8][KS][64]
    float* sm_acc = (float*)(lf_smem + LfCfg<C>::CTX_W);           // [4 waves][16][64]
    float* sm_m = sm_acc + 4 * 16 * 64;                           // [4][32]
    float* sm_s = sm_m + 4 * 32;
    const int n = blockIdx.y, blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    {
        const uint4* src = (const uint4*)wfrag + 4 * KS * 64;     // groups 4..11 of [q0..3, k0..3, v0..3]
        for (int i = tid; i < 8 * KS * 64; i += 256) wl[i] = src[i];
    }
    __syncthreads();
    const int ngroups = (hw + 31) >> 5;
    const el16_t* xs = xn + (size_t)n * hw * C + 8 * hi;
    la_f32x16 acc[4];
    float m[4], sum[4];
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        m[h] = -3.0e38f;
        sum[h] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[h][r] = 0.0f;
    }
    auto load = [&](int g, uint4 (&xf)[KS]) {
        const int p = min(g * 32 + l31, hw - 1);  // pixels past the end repeat the last one (their k' is masked below)
        const uint4* src = (const uint4*)(xs + (size_t)p * C);
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = src[2 * s];
        __builtin_amdgcn_sched_barrier(0);  // issue here, a group ahead (the scheduler would sink the loads to their uses)
    };
    int g = blk * gpb + wave;
    const int gend = min((blk + 1) * gpb, ngroups);
    uint4 xc[KS], xnx[KS];
    if (g < gend) load(g, xc);
    // W_k is pre-scaled by log2(e) (linattn_fused_pack): exp(k - max) = exp2(k2 - max2), and m[] is in log2 units
    auto group = [&](int g, auto full_c) {
        constexpr bool FULL = decltype(full_c)::value;  // a straight-line body: the masks of the tail group would split it into 60 blocks
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            int wlane = lane;
            asm volatile("" : "+v"(wlane));  // the weight fragments are loop-invariant LDS reads: hoisted, they cost 128+ registers
            la_f32x16 dk, dv;
#pragma unroll
            for (int r = 0; r < 16; ++r) dk[r] = dv[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const el16x8_t xa = __builtin_bit_cast(el16x8_t, xc[s]);
                dk = DYF_MFMA_32x32x16(xa, __builtin_bit_cast(el16x8_t, wl[(h * KS + s) * 64 + wlane]), dk, 0, 0, 0);
                dv = DYF_MFMA_32x32x16(xa, __builtin_bit_cast(el16x8_t, wl[((4 + h) * KS + s) * 64 + wlane]), dv, 0, 0, 0);
            }
            // lane (channel l31, hi): register r = pixel g*32 + 8 (r >> 2) + 4 hi + (r & 3)
            float gm = dk[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) gm = fmaxf(gm, dk[r]);
            gm = lf_max32(gm);
            if (__builtin_amdgcn_ballot_w64(gm > m[h]) != 0) {
                const float mn = fmaxf(m[h], gm), f = __builtin_amdgcn_exp2f(m[h] - mn);
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[h][r] *= f;
                sum[h] *= f;
                m[h] = mn;
            }
            uint32_t kh[8], vh[8];
#pragma unroll
            for (int r2 = 0; r2 < 8; ++r2) {
                float x0 = __builtin_amdgcn_exp2f(dk[2 * r2] - m[h]), x1 = __builtin_amdgcn_exp2f(dk[2 * r2 + 1] - m[h]);
                if (!FULL) {
                    const int p0 = g * 32 + 8 * (r2 >> 1) + 4 * hi + 2 * (r2 & 1);
                    x0 = p0 < hw ? x0 : 0.0f;
                    x1 = p0 + 1 < hw ? x1 : 0.0f;
                }
                sum[h] += x0 + x1;
                kh[r2] = pack_el16x2(x0, x1);
                vh[r2] = pack_el16x2(dv[2 * r2], dv[2 * r2 + 1]);
            }
            const uint32_t v0[4] = {vh[0], vh[1], vh[2], vh[3]}, v1[4] = {vh[4], vh[5], vh[6], vh[7]};
            const uint32_t k0[4] = {kh[0], kh[1], kh[2], kh[3]}, k1[4] = {kh[4], kh[5], kh[6], kh[7]};
            acc[h] = DYF_MFMA_32x32x16(la_frag(v0), la_frag(k0), acc[h], 0, 0, 0);
            acc[h] = DYF_MFMA_32x32x16(la_frag(v1), la_frag(k1), acc[h], 0, 0, 0);
        }
    };
    constexpr bool PREFETCH = KS <= 4;  // dim 128: a second fragment set spills; the second wave of the SIMD covers the loads
    for (; g < gend; g += 4) {
        // unconditional (past the end: re-read this group): a conditional prefetch is a branch join at which the compiler's
        // s_waitcnt model takes the NEWEST loads for the ones needed, and every MFMA of the group waits for the prefetch
        if (PREFETCH) load(g + 4 < gend ? g + 4 : g, xnx);
        if ((g + 1) * 32 <= hw) group(g, std::true_type{});  // wave-uniform
        else group(g, std::false_type{});
        if (PREFETCH) {
#pragma unroll
            for (int s = 0; s < KS; ++s) xc[s] = xnx[s];
        } else if (g + 4 < gend) {
            load(g + 4, xc);
        }
    }
    // merge the four waves per head (as linattn_ctx_mfma_kernel) and emit the partial
#pragma unroll
    for (int h = 0; h < 4; ++h) {
        const float sh = lf_sum32(sum[h]);
        if (hi == 0) {
            sm_m[wave * 32 + l31] = m[h];
            sm_s[wave * 32 + l31] = sh;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) sm_acc[(wave * 16 + r) * 64 + lane] = acc[h][r];
        __syncthreads();
        float M = sm_m[l31];
#pragma unroll
        for (int w = 1; w < 4; ++w) M = fmaxf(M, sm_m[w * 32 + l31]);
        float f[4], S = 0.0f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            f[w] = __builtin_amdgcn_exp2f(sm_m[w * 32 + l31] - M);
            S = fmaf(f[w], sm_s[w * 32 + l31], S);
        }
        float* o = part + ((size_t)(n * LA_HEADS + h) * nblk + blk) * LA_PART;
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
            const int r = wave * 4 + r4;
            float v = 0.0f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v = fmaf(f[w], sm_acc[(w * 16 + r) * 64 + lane], v);
            o[r * 64 + lane] = v;
        }
        if (tid < 32) {
            o[1024 + tid] = M;
            o[1056 + tid] = S;
        }
        __syncthreads();
    }
}

template <int C>
__global__ __launch_bounds__(256, 2) void linattn_fused_out_kernel(const el16_t* __restrict__ xn, const el16_t* __restrict__ xres, int hw,
                                                                const el16_t* __restrict__ wfrag, const el16_t* __restrict__ wofrag,
                                                                const float* __restrict__ bias, const el16_t* __restrict__ frags,
                                                                el16_t* __restrict__ y, int gpb) {
    constexpr int KS = LfCfg<C>::KS, OG = C / 32;
    extern __shared__ __attribute__((aligned(16))) unsigned char lf_smem[];
    uint4* wq = (uint4*)lf_smem;                                            // [4][KS][64]
    uint4* wo = (uint4*)(lf_smem + LfCfg<C>::OUT_WQ);                        // [OG][8][64]
    uint4* cf = (uint4*)(lf_smem + LfCfg<C>::OUT_WQ + LfCfg<C>::OUT_WO);      // [4 heads][hi s0, hi s1, lo s0, lo s1][64]
    float* bl = (float*)(lf_smem + LfCfg<C>::OUT_WQ + LfCfg<C>::OUT_WO + LfCfg<C>::OUT_CF);
    const int n = blockIdx.y, blk = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < 4 * KS * 64; i += 256) wq[i] = ((const uint4*)wfrag)[i];
    for (int i = tid; i < OG * 8 * 64; i += 256) wo[i] = ((const uint4*)wofrag)[i];
    for (int i = tid; i < 4 * 4 * 64; i += 256) cf[i] = ((const uint4*)frags)[(size_t)n * 4 * 4 * 64 + i];
    for (int i = tid; i < C; i += 256) bl[i] = bias[i];
    __syncthreads();
    const int ngroups = (hw + 31) >> 5;
    const el16_t* xs = xn + (size_t)n * hw * C + 8 * hi;
    const el16_t* rs = xres + (size_t)n * hw * C + 8 * hi;
    el16_t* ys = y + (size_t)n * hw * C + 8 * hi;
    const float scale = 0.17677669529663687f;  // 32^-1/2
    auto load = [&](int g, uint4 (&xf)[KS]) {
        const int p = min(g * 32 + l31, hw - 1);
        const uint4* src = (const uint4*)(xs + (size_t)p * C);
#pragma unroll
        for (int s = 0; s < KS; ++s) xf[s] = src[2 * s];
        __builtin_amdgcn_sched_barrier(0);
    };
    int g = blk * gpb + wave;
    const int gend = min((blk + 1) * gpb, ngroups);
    uint4 xc[KS], xnx[KS];
    if (g < gend) load(g, xc);
    for (; g < gend; g += 4) {
        load(g + 4 < gend ? g + 4 : g, xnx);  // unconditional: see linattn_fused_ctx_kernel
        const int p = g * 32 + l31;
        const bool valid = p < hw;
        const size_t poff = (size_t)(valid ? p : hw - 1) * C;
        uint4 res[OG][2];
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) res[og][g2] = *(const uint4*)(rs + poff + og * 32 + g2 * 16);
        la_f32x16 d3[OG];
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int r = 0; r < 16; ++r) d3[og][r] = 0.0f;
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            int wlane = lane;
            asm volatile("" : "+v"(wlane));  // as above: keep the LDS fragment reads inside the loop
            la_f32x16 d1;
#pragma unroll
            for (int r = 0; r < 16; ++r) d1[r] = 0.0f;
#pragma unroll
            for (int s = 0; s < KS; ++s)
                d1 = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, wq[(h * KS + s) * 64 + wlane]), __builtin_bit_cast(el16x8_t, xc[s]), d1, 0, 0, 0);
            // lane (pixel, hi): register r = q[d = 8 (r >> 2) + 4 hi + (r & 3)]
            float qm = d1[0];
#pragma unroll
            for (int r = 1; r < 16; ++r) qm = fmaxf(qm, d1[r]);
            qm = lf_max32(qm);
            float qs = 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                d1[r] = __builtin_amdgcn_exp2f(d1[r] - qm);  // W_q is pre-scaled by log2(e)
                qs += d1[r];
            }
            qs = lf_sum32(qs);
            const float qn = scale / qs;
            la_f32x16 d2;
#pragma unroll
            for (int r = 0; r < 16; ++r) d2[r] = 0.0f;
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                // q' in [0, 1] rounded to 16 bits (q itself is fp32 here; the unfused form rounds q before the exponential);
                // the context keeps its hi + lo parts: a second MFMA, no VALU work
                uint32_t qh[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) qh[t] = pack_el16x2(d1[s2 * 8 + 2 * t], d1[s2 * 8 + 2 * t + 1]);
                const el16x8_t ch = __builtin_bit_cast(el16x8_t, cf[(h * 4 + s2) * 64 + wlane]);
                const el16x8_t cl = __builtin_bit_cast(el16x8_t, cf[(h * 4 + 2 + s2) * 64 + wlane]);
                d2 = DYF_MFMA_32x32x16(ch, la_frag(qh), d2, 0, 0, 0);
                d2 = DYF_MFMA_32x32x16(cl, la_frag(qh), d2, 0, 0, 0);
            }
            // out_h[e = 8 (r >> 2) + 4 hi + (r & 3)][pixel]: registers 8 t .. 8 t + 7 are k-step 2 h + t of to_out
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                uint32_t ob[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) ob[i] = pack_el16x2(d2[t * 8 + 2 * i] * qn, d2[t * 8 + 2 * i + 1] * qn);
#pragma unroll
                for (int og = 0; og < OG; ++og)
                    d3[og] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, wo[(og * 8 + 2 * h + t) * 64 + wlane]), la_frag(ob), d3[og], 0, 0, 0);
            }
        }
        // lane (pixel, hi) holds channels 32 og + 8 (r >> 2) + 4 hi + (r & 3); after the exchange with lane pixel + 32 it
        // holds the 8 consecutive channels 32 og + 16 g2 + 8 hi + {0..7}: + bias + residual, one 16-byte store
#pragma unroll
        for (int og = 0; og < OG; ++og)
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(d3[og][g2 * 8 + i]), __float_as_uint(d3[og][g2 * 8 + 4 + i]), false, false);
                    v[i] = __uint_as_float(sw[0]);
                    v[4 + i] = __uint_as_float(sw[1]);
                }
                const int cb = og * 32 + g2 * 16 + hi * 8;
                const float4 b0 = *(const float4*)(bl + cb), b1 = *(const float4*)(bl + cb + 4);
                const uint4 rr = res[og][g2];
                uint4 o;
                o.x = pack_el16x2(v[0] + b0.x + el16_lo(rr.x), v[1] + b0.y + el16_hi(rr.x));
                o.y = pack_el16x2(v[2] + b0.z + el16_lo(rr.y), v[3] + b0.w + el16_hi(rr.y));
                o.z = pack_el16x2(v[4] + b1.x + el16_lo(rr.z), v[5] + b1.y + el16_hi(rr.z));
                o.w = pack_el16x2(v[6] + b1.z + el16_lo(rr.w), v[7] + b1.w + el16_hi(rr.w));
                // lanes past the end computed the sample's LAST pixel (clamped loads) and store it again, value for value: no
                // exec-masked branch around the store, so the s_waitcnt counts stay exact across the loop (a join would make the
                // next group's fragments wait for vmcnt(0), i.e. for these stores to be acknowledged)
                *(uint4*)(ys + poff + og * 32 + g2 * 16) = o;
            }
#pragma unroll
        for (int s = 0; s < KS; ++s) xc[s] = xnx[s];
    }
}

bool linattn_fused_supported(int c) {
    const bool on = !(dyf_form("DYF_LINATTN_FUSED") && atoi(dyf_form("DYF_LINATTN_FUSED")) == 0);
    return on && (c == 64 || c == 128);
}

hipError_t linattn_fused_init() {
    hipError_t e = hipFuncSetAttribute((const void*)linattn_fused_ctx_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LfCfg<64>::CTX_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)linattn_fused_ctx_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LfCfg<128>::CTX_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)linattn_fused_out_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, LfCfg<64>::OUT_LDS);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)linattn_fused_out_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, LfCfg<128>::OUT_LDS);
    return e;
}

// host side of the fragment orders above (w_qkv [384][c], w_out [c][128], both row-major fp32)
void linattn_fused_pack(const float* w_qkv, const float* w_out, int c, el16_t* qkv_frag, el16_t* out_frag) {
    const int ks = c / 16;
    for (int grp = 0; grp < 12; ++grp)      // q0..3, k0..3, v0..3: rows 32 grp .. 32 grp + 31 of to_qkv
        for (int s = 0; s < ks; ++s)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    qkv_frag[(((size_t)grp * ks + s) * 64 + lane) * 8 + j] =  // q and k feed exponentials: base 2 on the device
                        f32_to_el16((grp < 8 ? 1.4426950408889634f : 1.0f) * w_qkv[(size_t)(32 * grp + (lane & 31)) * c + 16 * s + 8 * (lane >> 5) + j]);
    for (int og = 0; og < c / 32; ++og)
        for (int s3 = 0; s3 < 8; ++s3)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j) {
                    const int he = 16 * s3 + 8 * (j >> 2) + 4 * (lane >> 5) + (j & 3);
                    out_frag[(((size_t)og * 8 + s3) * 64 + lane) * 8 + j] = f32_to_el16(w_out[(size_t)(32 * og + (lane & 31)) * 128 + he]);
                }
}

hipError_t launch_linear_attention_fused(const LinAttnFusedArgs& a, hipStream_t s) {
    // Pixels per workgroup.  A workgroup loads the block's weights into LDS (32 / 64 KB) and then walks its 32-pixel groups, four at
    // a time: with 1 024 pixels per workgroup a 60 x 60 plane is 4 workgroups per sample and a 30 x 30 plane ONE -- at 38 rows the two
    // passes were 152 / 38 workgroups of a 20 us serial chain each (22-28 us per launch, eight launches per forward,
    // profiles/r05f_oisst_nb38).  Below 384 workgroups the launch halves / quarters the block (more, shorter chains; the merge
    // kernel takes any number of partials).  DYF_LINATTN_GPB = 32 / 16 / 8 forces a size (read per launch).
    const int ngroups = (a.hw + 31) / 32, BH = a.n * LA_HEADS;
    int gpb = a.groups_per_block;
    if (const char* ge = dyf_form("DYF_LINATTN_GPB")) gpb = atoi(ge);
    if (gpb != 32 && gpb != 16 && gpb != 8) {
        gpb = 32;
        // (measured, OISST rollouts with blocks of 32 only / this rule at 512: 38 rows 1 791 / 1 824 fields/s, 75 rows 2 651 / 2 758, 150 rows
        // 3 676 / 3 691, 300 rows on three groups 4 090 / 4 073 -- 400-workgroup launches of a group gain nothing: the rule stops at 384)
        while (gpb > 8 && (long long)((ngroups + gpb - 1) / gpb) * a.n < 384) gpb >>= 1;
    }
    int nblk = (ngroups + gpb - 1) / gpb;
    while (gpb < 32 && (a.scratch_floats <= 0 || (long long)BH * nblk * LA_PART + (long long)BH * 1024 > a.scratch_floats)) {
        gpb <<= 1;  // the partials of the smaller blocks do not fit the scratch
        nblk = (ngroups + gpb - 1) / gpb;
    }
    float* part = a.scratch;
    el16_t* frags = (el16_t*)(a.scratch + (size_t)BH * nblk * LA_PART);
    const dim3 grid(nblk, a.n);
    dyf_form_note(gpb == 32 ? "linattn_fused_kernels<gpb=32>" : gpb == 16 ? "linattn_fused_kernels<gpb=16>" : "linattn_fused_kernels<gpb=8>", a.n);
    if (a.c == 64) {
        hipLaunchKernelGGL(linattn_fused_ctx_kernel<64>, grid, dim3(256), LfCfg<64>::CTX_LDS, s, a.xn, a.hw, a.wqkv_frag, part, nblk, gpb);
    } else {
        hipLaunchKernelGGL(linattn_fused_ctx_kernel<128>, grid, dim3(256), LfCfg<128>::CTX_LDS, s, a.xn, a.hw, a.wqkv_frag, part, nblk, gpb);
    }
    hipLaunchKernelGGL(linattn_merge_kernel, dim3(BH, (nblk >= 32 && BH < 128) ? 4 : 1), dim3(256), 0, s, (const float*)part, nblk, 1.0f / (float)a.hw, frags, 1);
    if (a.c == 64) {
        hipLaunchKernelGGL(linattn_fused_out_kernel<64>, grid, dim3(256), LfCfg<64>::OUT_LDS, s, a.xn, a.xres, a.hw, a.wqkv_frag, a.wout_frag, a.bout, (const el16_t*)frags, a.y, gpb);
    } else {
        hipLaunchKernelGGL(linattn_fused_out_kernel<128>, grid, dim3(256), LfCfg<128>::OUT_LDS, s, a.xn, a.xres, a.hw, a.wqkv_frag, a.wout_frag, a.bout, (const el16_t*)frags, a.y, gpb);
    }
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ Attention
// stream of (batch row n, head h): the row's stream key with the head folded in
__device__ __forceinline__ RngKey attn_drop_key(const DropSpec& d, int n, uint32_t h) {
    RngKey k = drop_row_key(d, n);
    if (d.mode == 1) k.k0 = fmix32(k.k0 + (h + 1u) * 0x9E3779B9u);
    return k;
}
// element (query i, key j) of head-sample bh: RNG stream keyed per bh (i*N + j stays below 2^32 for N <= 65535);
// injected masks are indexed as the (b, h, i, j) tensor the reference's nn.Dropout sees
__device__ __forceinline__ float attn_drop(float p, const DropSpec& d, RngKey key, uint32_t bh, uint32_t i, uint32_t j,
                                           uint32_t n) {
    if (d.mode == 0) return p;
    if (d.mode == 1) return rng_keep8(i * n + j, key, d.thresh8) ? p * d.scale8 : 0.0f;  // quad form (common.h)
    return d.mask[((size_t)bh * n + i) * n + j] != 0 ? p * d.scale : 0.0f;
}
// scale of the survivors: engine streams use the quad form's 256 / k, injected masks the reference's 1 / (1 - p)
__device__ __forceinline__ float attn_drop_scale(const DropSpec& d) { return d.mode == 1 ? d.scale8 : d.scale; }

// One thread per query, keys/values of the head streamed through LDS in tiles of 64, online softmax in fp32.
__global__ __launch_bounds__(64) void attention_kernel(AttnArgs a) {
    __shared__ float ks[64][33], vs[64][33];
    const int qtiles = (a.hw + 63) / 64;
    const int bh = blockIdx.x / qtiles, qt = blockIdx.x % qtiles;
    const int n = bh / a.heads, h = bh % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32;
    const el16_t* base = a.qkv + (size_t)n * a.hw * C3;
    const int i = qt * 64 + threadIdx.x;
    const bool live = i < a.hw;
    const float scale = 0.17677669529663687f;
    float q[32], o[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        q[c] = live ? el16_to_f32(base[(size_t)i * C3 + h * 32 + c]) * scale : 0.0f;
        o[c] = 0.0f;
    }
    float m = -3.0e38f, l = 0.0f;
    const RngKey key = attn_drop_key(a.drop, n, (uint32_t)h);
    for (int j0 = 0; j0 < a.hw; j0 += 64) {
        __syncthreads();
        for (int t = threadIdx.x; t < 64 * 32; t += 64) {
            const int jj = t >> 5, c = t & 31;
            const int j = j0 + jj;
            ks[jj][c] = j < a.hw ? el16_to_f32(base[(size_t)j * C3 + hd + h * 32 + c]) : 0.0f;
            vs[jj][c] = j < a.hw ? el16_to_f32(base[(size_t)j * C3 + 2 * hd + h * 32 + c]) : 0.0f;
        }
        __syncthreads();
        const int jn = min(64, a.hw - j0);
        for (int jj = 0; jj < jn; ++jj) {
            float sc = 0.0f;
#pragma unroll
            for (int c = 0; c < 32; ++c) sc = fmaf(q[c], ks[jj][c], sc);
            if (sc > m) {  // rescale the running sums to the new maximum
                const float f = __expf(m - sc);
                l *= f;
#pragma unroll
                for (int c = 0; c < 32; ++c) o[c] *= f;
                m = sc;
            }
            const float pr = __expf(sc - m);
            l += pr;  // the softmax normaliser is computed BEFORE dropout (attention.py:69-70)
            const float pd = live ? attn_drop(pr, a.drop, key, (uint32_t)bh, (uint32_t)i, (uint32_t)(j0 + jj), (uint32_t)a.hw) : 0.0f;
#pragma unroll
            for (int c = 0; c < 32; ++c) o[c] = fmaf(pd, vs[jj][c], o[c]);
        }
    }
    if (!live) return;
    el16_t* op = a.out + ((size_t)n * a.hw + i) * hd + h * 32;  // "b h (x y) d -> b (h d) x y"
    const float inv = 1.0f / l;
#pragma unroll
    for (int c = 0; c < 32; ++c) op[c] = f32_to_el16(o[c] * inv);
}

// ------------------------------------------------------------------------------------------------ flash attention
// K8 on MFMA (attention.py:62-72, heads of 32 channels): one wave owns 32 queries, a workgroup 128 queries of one
// (sample, head); keys/values stream through LDS in tiles of 64.  Both contractions are computed TRANSPOSED so that the
// softmax row of a query lives in ONE lane pair (lane q and q+32) and never crosses lanes otherwise:
//   S^T[key][q] = K . Q^T        A = K rows (keys),  B = Q rows (queries)   (2 x v_mfma_f32_32x32x16_bf16, k = channel)
//   O^T[d][q]  += V^T . P^T      A = V^T rows (d),   B = P^T = the exponentiated S^T registers, packed to bf16
// The MFMA k-slot <-> key mapping of the second product is chosen to be the C layout of the first (key = 16s + (j&3) +
// 8(j>>2) + 4*hi), so P feeds the second MFMA straight from registers; V is staged transposed ([d][key], rows padded to
// 136 B: conflict-free ds_read_b64).  Online softmax in fp32; the normaliser is accumulated BEFORE dropout.
typedef __attribute__((ext_vector_type(16))) float fa_f32x16;

__global__ __launch_bounds__(256) void flash_attention_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) el16_t Ks[64 * 32];   // [key][32 ch], 16-B chunk ^= (key >> 2) & 3
    __shared__ __attribute__((aligned(16))) el16_t Vt[32 * 68];   // [ch][64 keys + 4 pad]
    const int qblocks = (a.hw + 127) / 128;
    const int bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    const int n = bh / a.heads, h = bh % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32, N = a.hw;
    const el16_t* base = a.qkv + (size_t)n * N * C3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q = qb * 128 + wave * 32 + l31;
    const float scale = 0.17677669529663687f;  // 32^-1/2
    // Q fragments (B operand of S^T): lane (q, hi) holds channels ks*16 + hi*8 .. +8
    el16x8_t qf[2];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q < N) v = *(const uint4*)(base + (size_t)q * C3 + h * 32 + ks * 16 + hi * 8);
        qf[ks] = *(el16x8_t*)&v;
    }
    fa_f32x16 o;
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = 0.0f;
    float m = -1.0e30f, l = 0.0f;   // running maximum in the log2 domain (scores are pre-multiplied by scale * log2 e)
    const RngKey dkey = attn_drop_key(a.drop, n, (uint32_t)h);
    const float c2 = scale * 1.4426950408889634f;

    // K / V of the NEXT 64-key tile travel through registers: their global loads are issued right after the current tile has
    // been written to LDS and land while it is being consumed (the un-prefetched form exposed one HBM/L2 round trip per tile:
    // 4.2 ms for 4 x 4 heads x 16 384 tokens)
    const int skey = tid >> 2, sch = tid & 3;   // staging role: thread -> (key, 16-B chunk of 8 channels)
    uint4 kv_n = make_uint4(0, 0, 0, 0), vv_n = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int j0) {
        const int j = j0 + skey;
        kv_n = make_uint4(0, 0, 0, 0);
        vv_n = kv_n;
        if (j < N) {
            kv_n = *(const uint4*)(base + (size_t)j * C3 + hd + h * 32 + sch * 8);
            vv_n = *(const uint4*)(base + (size_t)j * C3 + 2 * hd + h * 32 + sch * 8);
        }
    };
    fetch(0);

    // one 32-key sub-tile; variant 1 (FAST): all 32 keys exist and no dropout on the probabilities; variant 2: all 32 keys and
    // the query exist, dropout from the engine's generator with an even token count -- keys j, j + 1 of a register pair share
    // one keep word (rng_keep hashes element pair (q*N + j) >> 1), so 8 hashes serve the lane's 16 probabilities and there are
    // no per-element bounds or mode tests (the general form, variant 0, ran the 16 384-token attention of the 512^2
    // configuration at 4.3 ms per NB = 4 call against 1.6 ms without dropout)
    auto subtile = [&](int jb, int st, auto variant_c) {
        constexpr int VARIANT = decltype(variant_c)::value;
        constexpr bool FAST = VARIANT != 0;
        fa_f32x16 sc;
#pragma unroll
        for (int r = 0; r < 16; ++r) sc[r] = 0.0f;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int key = st * 32 + l31;
            const el16x8_t kf = *(const el16x8_t*)(Ks + key * 32 + (((ks * 2 + hi) ^ ((key >> 2) & 3)) << 3));
            sc = DYF_MFMA_32x32x16(kf, qf[ks], sc, 0, 0, 0);
        }
        // lane (q, hi) now holds keys jb + (r&3) + 8(r>>2) + 4hi of query q
        float tmax = -1.0e30f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (FAST) sc[r] *= c2;
            else sc[r] = (jb + (r & 3) + 8 * (r >> 2) + 4 * hi) < N ? sc[r] * c2 : -1.0e30f;
            tmax = fmaxf(tmax, sc[r]);
        }
        tmax = fmaxf(tmax, __shfl_xor(tmax, 32, 64));
        const float mn = fmaxf(m, tmax);
        const float alpha = __builtin_amdgcn_exp2f(m - mn);
        m = mn;
        float psum = 0.0f;
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float e = __builtin_amdgcn_exp2f(sc[r] - mn);
            psum += e;
            if (VARIANT == 1) p[r] = e;
            else if (VARIANT == 2) p[r] = e;  // masked below, pair by pair
            else {
                const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                p[r] = (j < N && q < N) ? attn_drop(e, a.drop, dkey, (uint32_t)bh, (uint32_t)q, (uint32_t)j, (uint32_t)N) : 0.0f;
            }
        }
        if (VARIANT == 2) {
            const uint32_t th = a.drop.thresh8;
            const float dsc = a.drop.scale8;
            const uint32_t e0 = (uint32_t)q * (uint32_t)N + (uint32_t)(jb + 4 * hi);  // a multiple of 4: N % 4 == 0, jb and 4*hi are
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g + 3: keys e0 + 8g + {0..3} = ONE quad word (common.h rng_keep8)
                const uint32_t w = rng_quad_word((e0 + 8u * (uint32_t)g) >> 2, dkey);
                p[4 * g] = (w & 0xffu) < th ? p[4 * g] * dsc : 0.0f;
                p[4 * g + 1] = ((w >> 8) & 0xffu) < th ? p[4 * g + 1] * dsc : 0.0f;
                p[4 * g + 2] = ((w >> 16) & 0xffu) < th ? p[4 * g + 2] * dsc : 0.0f;
                p[4 * g + 3] = (w >> 24) < th ? p[4 * g + 3] * dsc : 0.0f;
            }
        }
        psum += __shfl_xor(psum, 32, 64);
        l = l * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) o[r] *= alpha;
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            uint32_t pk[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) pk[t] = pack_el16x2(p[s2 * 8 + 2 * t], p[s2 * 8 + 2 * t + 1]);
            const el16x8_t pf = *(el16x8_t*)pk;
            // V^T fragment: lane (d = l31, hi): keys st*32 + 16*s2 + 4*hi + {0..3} and + 8
            const el16_t* vr = Vt + l31 * 68 + st * 32 + s2 * 16 + 4 * hi;
            uint2 v0 = *(const uint2*)vr, v1 = *(const uint2*)(vr + 8);
            uint32_t vw[4] = {v0.x, v0.y, v1.x, v1.y};
            const el16x8_t vf = *(el16x8_t*)vw;
            o = DYF_MFMA_32x32x16(vf, pf, o, 0, 0, 0);
        }
    };

    for (int j0 = 0; j0 < N; j0 += 64) {
        __syncthreads();  // every wave is done reading the previous tile
        {   // stage K [64][32] and V^T [32][64] from the prefetched registers
            *(uint4*)(Ks + skey * 32 + ((sch ^ ((skey >> 2) & 3)) << 3)) = kv_n;
            const el16_t* ve = (const el16_t*)&vv_n;
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(sch * 8 + i) * 68 + skey] = ve[i];
        }
        if (j0 + 64 < N) fetch(j0 + 64);
        __syncthreads();
        const bool whole = j0 + 64 <= N;  // block-uniform
        if (whole && a.drop.mode == 0) {
            subtile(j0, 0, std::integral_constant<int, 1>{});
            subtile(j0 + 32, 1, std::integral_constant<int, 1>{});
        } else if (whole && a.drop.mode == 1 && (N & 3) == 0 && (qb + 1) * 128 <= N) {
            subtile(j0, 0, std::integral_constant<int, 2>{});
            subtile(j0 + 32, 1, std::integral_constant<int, 2>{});
        } else {
            subtile(j0, 0, std::integral_constant<int, 0>{});
            if (j0 + 32 < N) subtile(j0 + 32, 1, std::integral_constant<int, 0>{});
        }
    }
    if (q >= N) return;
    // O^T[d][q]: lane (q, hi) holds d = (r&3) + 8(r>>2) + 4hi  -> four 8-byte stores of 4 consecutive channels
    const float inv = 1.0f / l;
    el16_t* op = a.out + ((size_t)n * N + q) * hd + h * 32;  // "b h (x y) d -> b (h d) x y"
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack_el16x2(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        w.y = pack_el16x2(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(uint2*)(op + 8 * g + 4 * hi) = w;
    }
}

// ---- second form (round 3): the same data flow with the softmax's VALU work cut to what the exponentials need.
// The first form is VALU-bound: per 32-key sub-tile a lane spends 16 scale multiplies, 16 max, 16 subtractions, 16 exp2, 16 adds,
// 16 rescales of O, 8 packs and two cross-lane exchanges (~370 issue cycles) next to 4 MFMAs (128 cycles).  Here:
//   * Q is multiplied by scale * log2(e) ONCE, when its fragments are loaded;
//   * the running maximum m enters through the MFMA's C operand: the score accumulators START at -m, so the MFMA delivers s - m
//     (16 register moves instead of 16 subtractions behind the MFMA's latency);
//   * m is LAZY: it moves only when some score of the sub-tile exceeds it by more than 2^6 (then, and on the first sub-tile, a
//     wave-uniform slow path rescales O and l) -- exp2 of a value <= 6 is at most 64, harmless in fp32 sums and in the 16-bit P
//     operand -- so the common path has no rescale of O, no alpha, and no cross-lane exchange at all;
//   * the normaliser is summed per LANE (both lanes of a query share m) and the two halves meet once, at the end;
//   * the dropout scale 1/(1-p) is applied once to the output instead of to every kept probability.
// Common path per sub-tile and lane: 8 v_max3, one compare, 16 v_exp_f32, 16 adds, 8 packs (+ 8 keep words and 16 selects with
// dropout on the probabilities).
// VARIANT 1: whole tile, no dropout; 2: whole tile, engine dropout with paired keep words; 0: general (partial tiles; dropout per
// element when DROP).  DROP is the kernel's compile-time dropout switch: the no-dropout kernel carries no generator code at all
// (register budget: 128 per lane for 4 waves per SIMD).
// QB: 32-query blocks per wave.  QB = 2 (sequences of >= 512 tokens): a wave owns 64 queries -- two independent softmax chains
// for the scheduler to interleave, and every K / V fragment read from LDS feeds two MFMAs.
typedef float fa_f32x2 __attribute__((ext_vector_type(2)));
// kf0 / kf1 / vf0: this lane's LDS addresses of the K fragments (ks = 0, 1) and of the V^T fragment of sub-tile 0 (sub-tile st adds
// a wave-uniform offset: the swizzle key (key >> 2) & 3 does not depend on st).  NEGM: the tuple -m lives in registers across the
// loop and is the MFMA's C operand (kernels with the registers to spare); otherwise 16 moves per sub-tile rebuild it.
#ifndef FA2_BIAS
#define FA2_BIAS 0  // -DFA2_BIAS=1: -m from a third (bf16, k-slot 0) matrix instruction instead of 16 v_mov per sub-tile: measured
                   // 1.162-1.186 vs 1.143 ms without dropout, 1.870 vs 1.903 ms with -- the kernel is not issue-bound; not kept
#endif
typedef __attribute__((ext_vector_type(4))) short fa_bf16x4;
#ifndef FA2_PKMOV
#define FA2_PKMOV 0  // measured: 1.184 ms with v_pk_mov_b32 vs 1.159 ms with the compiler's 16 v_mov_b32 (same box): not kept
#endif
template <int VARIANT, bool DROP, int QB, bool NEGM>
__device__ __forceinline__ void fa2_subtile(const AttnArgs& a, const el16_t* kf0, const el16_t* kf1, const el16_t* vf0, const el16x8_t (&qf)[QB][2],
                                            fa_f32x16 (&o)[QB], fa_f32x16 (&negm)[QB], float (&m)[QB], fa_f32x2 (&l2)[QB], bool& first, int jb,
                                            int st, int q0, int N, int hi, RngKey dkey, uint32_t bh, fa_bf16x4 aone, fa_bf16x4 (&bm)[QB]) {
    fa_f32x16 sc[QB];
    {
        const el16x8_t k0 = *(const el16x8_t*)(kf0 + st * 1024), k1 = *(const el16x8_t*)(kf1 + st * 1024);
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            if (NEGM) {
                sc[b] = DYF_MFMA_32x32x16(k0, qf[b][0], negm[b], 0, 0, 0);
#if FA2_BIAS
            } else if (true) {
                // the -m of every score comes from a THIRD matrix instruction (k-slot 0: 1 on the key side, -m on the query side, bf16)
                // on the idle matrix pipe instead of 16 v_mov per sub-tile on the saturated vector pipe; m is kept bf16-exact
                const fa_f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                sc[b] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(aone, bm[b], zero, 0, 0, 0);
                sc[b] = DYF_MFMA_32x32x16(k0, qf[b][0], sc[b], 0, 0, 0);
#endif
            } else {
                // accumulator initialised with -m; -DFA2_PKMOV=1: two registers per instruction (v_pk_mov_b32) -- an experiment
#if FA2_PKMOV
                const fa_f32x2 nm2 = {-m[b], -m[b]};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {
                    fa_f32x2 t;
                    asm volatile("v_pk_mov_b32 %0, %1, %1" : "=v"(t) : "v"(nm2));
                    sc[b][r] = t[0];
                    sc[b][r + 1] = t[1];
                }
#else
                const float nm = -m[b];
#pragma unroll
                for (int r = 0; r < 16; ++r) sc[b][r] = nm;
#endif
                sc[b] = DYF_MFMA_32x32x16(k0, qf[b][0], sc[b], 0, 0, 0);
            }
            sc[b] = DYF_MFMA_32x32x16(k1, qf[b][1], sc[b], 0, 0, 0);
        }
    }
    // lane (q, hi) holds (score - m) of keys jb + (r&3) + 8(r>>2) + 4hi of query q, in the log2 domain
    float tmax[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        if (VARIANT == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (jb + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) sc[b][r] = -1.0e30f;
        }
        float t = fmaxf(fmaxf(sc[b][0], sc[b][1]), sc[b][2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) t = fmaxf(fmaxf(t, sc[b][r]), sc[b][r + 1]);
        tmax[b] = fmaxf(t, sc[b][15]);
    }
    const float tany = QB == 2 ? fmaxf(tmax[0], tmax[QB - 1]) : tmax[0];
    if (first || __builtin_amdgcn_ballot_w64(tany > 6.0f) != 0ull) {  // wave-uniform; rare after the first sub-tiles
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            const float tm = fmaxf(tmax[b], __shfl_xor(tmax[b], 32, 64));  // both lanes of a query agree on the new maximum
            float delta = first ? tm : fmaxf(tm, 0.0f);                     // first sub-tile: m = the exact maximum (m was 0)
#if FA2_BIAS
            {   // m stays exactly representable in bf16 (round to nearest even): it travels as a bf16 MFMA operand
                uint32_t u = __builtin_bit_cast(uint32_t, m[b] + delta);
                u = (u + 0x7FFFu + ((u >> 16) & 1u)) & 0xFFFF0000u;
                const float mn = __builtin_bit_cast(float, u);
                delta = mn - m[b];
                bm[b][0] = (short)(hi == 0 ? ((__builtin_bit_cast(uint32_t, -mn)) >> 16) : 0u);
            }
#endif
            // (first: O and l are still zero -- and a first maximum below -128 would make exp2(-delta) infinite: 0 * inf)
            const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);
            m[b] += delta;
            l2[b] *= alpha;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                o[b][r] *= alpha;
                sc[b][r] -= delta;
                if (NEGM) negm[b][r] = -m[b];
            }
        }
        first = false;
    }
    uint32_t pk[QB][8];
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        float p[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
#ifdef FA_EXP_NO_EXP
            p[r] = sc[b][r] + 1.0f;  // timing experiment (wrong results): no exponentials
#else
            p[r] = __builtin_amdgcn_exp2f(sc[b][r]);
#endif
        }
        // the normaliser is accumulated BEFORE dropout (attention.py:69-70), two elements per instruction (v_pk_add_f32: a plain
        // wave64 VALU instruction occupies the SIMD for 4 cycles -- PMC: 4.6 cycles per VALU instruction, VALU busy 82 % of the
        // kernel against 23 % for the MFMA pipe)
#pragma unroll
        for (int r = 0; r < 16; r += 2) l2[b] += fa_f32x2{p[r], p[r + 1]};
        const int q = q0 + 32 * b;
        if (VARIANT == 2) {
            const uint32_t th = a.drop.thresh8;
            const uint32_t e0 = (uint32_t)q * (uint32_t)N + (uint32_t)(jb + 4 * hi);  // a multiple of 4: N % 4 == 0, jb and 4*hi are
#pragma unroll
            for (int g = 0; g < 4; ++g) {  // registers 4g .. 4g + 3: keys e0 + 8g + {0..3} = ONE quad word
                const uint32_t w = rng_quad_word((e0 + 8u * (uint32_t)g) >> 2, dkey);
                p[4 * g] = (w & 0xffu) < th ? p[4 * g] : 0.0f;
                p[4 * g + 1] = ((w >> 8) & 0xffu) < th ? p[4 * g + 1] : 0.0f;
                p[4 * g + 2] = ((w >> 16) & 0xffu) < th ? p[4 * g + 2] : 0.0f;
                p[4 * g + 3] = (w >> 24) < th ? p[4 * g + 3] : 0.0f;
            }
        } else if (VARIANT == 0 && DROP) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int j = jb + (r & 3) + 8 * (r >> 2) + 4 * hi;
                const bool keep = j < N && q < N &&
                                  (a.drop.mode == 1 ? rng_keep8((uint32_t)q * (uint32_t)N + (uint32_t)j, dkey, a.drop.thresh8)
                                                    : (a.drop.mask[((size_t)bh * N + q) * N + j] != 0));
                p[r] = keep ? p[r] : 0.0f;
            }
        }
#pragma unroll
        for (int t = 0; t < 8; ++t) pk[b][t] = pack_el16x2(p[2 * t], p[2 * t + 1]);
    }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
        const el16_t* vr = vf0 + st * 32 + s2 * 16;  // V^T fragment: lane (d = l31, hi): keys st*32 + 16*s2 + 4*hi + {0..3} and + 8
        uint2 v0 = *(const uint2*)vr, v1 = *(const uint2*)(vr + 8);
        uint32_t vw[4] = {v0.x, v0.y, v1.x, v1.y};
        const el16x8_t vf = *(el16x8_t*)vw;
#pragma unroll
        for (int b = 0; b < QB; ++b) {
            uint32_t pw[4] = {pk[b][4 * s2], pk[b][4 * s2 + 1], pk[b][4 * s2 + 2], pk[b][4 * s2 + 3]};
            o[b] = DYF_MFMA_32x32x16(vf, *(el16x8_t*)pw, o[b], 0, 0, 0);
        }
    }
}

// Occupancy decides this kernel (measured, 16 384 tokens x 4 heads x NB = 4, no dropout): the chain K-fragment read -> 2 MFMA ->
// max -> 16 exp -> pack -> V-fragment read -> 2 MFMA of a sub-tile is latency, not issue, bound -- removing the exponentials AND the
// V^T staging altogether moved 1.72 ms to 1.49 ms, while 3 instead of 2 resident waves per SIMD moved it to 1.24 ms.  Hence the
// register diet (no second accumulator tuple, rolled sub-tile loop) and 4 waves per SIMD for the 32-query form.
#ifndef FA2_MINW
#define FA2_MINW 4
#endif
#ifndef FA2_MINW2
#define FA2_MINW2 3
#endif
template <bool DROP, int QB>
__global__ __launch_bounds__(256, QB == 2 ? FA2_MINW2 : FA2_MINW) void flash_attention2_kernel(AttnArgs a) {
    // NEGM (the -m tuple resident in registers) measured SLOWER for the 32-query no-dropout kernel (1.24 vs 1.17 ms: 128 registers
    // leave the scheduler no slack at 4 waves per SIMD): off everywhere; -DFA2_NEGM=1 re-enables the experiment
#ifndef FA2_NEGM
#define FA2_NEGM 0
#endif
    constexpr bool NEGM = FA2_NEGM && !DROP && QB == 1;
    __shared__ __attribute__((aligned(16))) el16_t Ks[64 * 32];   // [key][32 ch], 16-B chunk ^= (key >> 2) & 3
    __shared__ __attribute__((aligned(16))) el16_t Vt[32 * 68];   // [ch][64 keys + 4 pad]
    constexpr int QW = 32 * QB, QG = 4 * QW;   // queries per wave / per workgroup
    const int qblocks = (a.hw + QG - 1) / QG;
    // XCD-aware block map: workgroup b runs on XCD b % 8 and every XCD has its own 4 MB L2.  All query blocks of one (sample,
    // head) stream the same K / V (2 MB at 16 384 tokens): dealt round-robin over the XCDs every L2 sees every (sample, head)
    // -- 33 MB at NB = 4 -- and the kernel runs at the fabric's rate; with (sample, head) pinned to XCD (bh % 8) each L2
    // holds the one or two K / V sets its workgroups are walking.
    int bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    if ((a.n * a.heads) % 8 == 0) {
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        bh = (k / qblocks) * 8 + xcd;
        qb = k % qblocks;
    }
    const int n = bh / a.heads, h = bh % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32, N = a.hw;
    const el16_t* base = a.qkv + (size_t)n * N * C3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = qb * QG + wave * QW + l31;   // query of block 0; block b: + 32 b
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-1/2 * log2(e): scores live in the log2 domain
    el16x8_t qf[QB][2];  // Q fragments (B operand of S^T), pre-multiplied by c2: lane (q, hi) holds channels ks*16 + hi*8 .. +8
    fa_f32x16 o[QB], negm[QB];
    float m[QB];
    fa_f32x2 l2[QB];
    const el16_t* kf0 = Ks + l31 * 32 + (((0 + hi) ^ ((l31 >> 2) & 3)) << 3);
    const el16_t* kf1 = Ks + l31 * 32 + (((2 + hi) ^ ((l31 >> 2) & 3)) << 3);
    const el16_t* vf0 = Vt + l31 * 68 + 4 * hi;
#pragma unroll
    for (int b = 0; b < QB; ++b) {
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            uint4 v = make_uint4(0, 0, 0, 0);
            if (q0 + 32 * b < N) v = *(const uint4*)(base + (size_t)(q0 + 32 * b) * C3 + h * 32 + ks * 16 + hi * 8);
            uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int t = 0; t < 4; ++t) w[t] = pack_el16x2(el16_lo(w[t]) * c2, el16_hi(w[t]) * c2);
            qf[b][ks] = *(el16x8_t*)w;
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) o[b][r] = negm[b][r] = 0.0f;
        m[b] = 0.0f;
        l2[b] = fa_f32x2{0.0f, 0.0f};
    }
    bool first = true;
    const RngKey dkey = DROP ? attn_drop_key(a.drop, n, (uint32_t)h) : RngKey{0u, 0u};
    // bias operands of the third matrix instruction (32x32x8, lane (row / column, hi) holds k = 4 hi .. 4 hi + 3): k-slot 0 carries
    // 1 on the key side and -m on the query side
    const fa_bf16x4 aone = {(short)(hi == 0 ? 0x3F80 : 0), 0, 0, 0};
    fa_bf16x4 bm[QB];
#pragma unroll
    for (int b = 0; b < QB; ++b) bm[b] = fa_bf16x4{0, 0, 0, 0};

    const int skey = tid >> 2, sch = tid & 3;   // staging role: thread -> (key, 16-B chunk of 8 channels)
    uint4 kv_n = make_uint4(0, 0, 0, 0), vv_n = make_uint4(0, 0, 0, 0);
    auto fetch = [&](int j0) {  // K / V of the NEXT 64-key tile travel through registers while the current one is consumed
        const int j = j0 + skey;
        kv_n = make_uint4(0, 0, 0, 0);
        vv_n = kv_n;
        if (j < N) {
            kv_n = *(const uint4*)(base + (size_t)j * C3 + hd + h * 32 + sch * 8);
            vv_n = *(const uint4*)(base + (size_t)j * C3 + 2 * hd + h * 32 + sch * 8);
        }
    };
    fetch(0);
    for (int j0 = 0; j0 < N; j0 += 64) {
#ifndef FA_EXP_NOSYNC
        __syncthreads();  // every wave is done reading the previous tile
#endif
        {
            *(uint4*)(Ks + skey * 32 + ((sch ^ ((skey >> 2) & 3)) << 3)) = kv_n;
            const el16_t* ve = (const el16_t*)&vv_n;
#ifdef FA_EXP_NO_VT
            if (j0 == 0)  // timing experiment (wrong results): V^T staged once
#endif
#pragma unroll
            for (int i = 0; i < 8; ++i) Vt[(sch * 8 + i) * 68 + skey] = ve[i];
        }
        if (j0 + 64 < N) fetch(j0 + 64);
#ifndef FA_EXP_NOSYNC
        __syncthreads();
#endif
        const bool whole = j0 + 64 <= N;  // block-uniform
        // (the two sub-tiles of a tile run as a rolled loop: unrolled, the compiler interleaves them and needs > 128 registers)
        if (whole && !DROP) {
#pragma nounroll
            for (int st = 0; st < 2; ++st)
                fa2_subtile<1, false, QB, NEGM>(a, kf0, kf1, vf0, qf, o, negm, m, l2, first, j0 + 32 * st, st, q0, N, hi, dkey, (uint32_t)bh, aone, bm);
        } else if (DROP && whole && a.drop.mode == 1 && (N & 3) == 0 && (qb + 1) * QG <= N) {
#pragma nounroll
            for (int st = 0; st < 2; ++st)
                fa2_subtile<2, DROP, QB, NEGM>(a, kf0, kf1, vf0, qf, o, negm, m, l2, first, j0 + 32 * st, st, q0, N, hi, dkey, (uint32_t)bh, aone, bm);
        } else {
            fa2_subtile<0, DROP, QB, NEGM>(a, kf0, kf1, vf0, qf, o, negm, m, l2, first, j0, 0, q0, N, hi, dkey, (uint32_t)bh, aone, bm);
            if (j0 + 32 < N) fa2_subtile<0, DROP, QB, NEGM>(a, kf0, kf1, vf0, qf, o, negm, m, l2, first, j0 + 32, 1, q0, N, hi, dkey, (uint32_t)bh, aone, bm);
        }
    }
#pragma unroll
    for (int b = 0; b < QB; ++b) {
        const float ll = l2[b].x + l2[b].y;
        const float l = ll + __shfl_xor(ll, 32, 64);
        const int q = q0 + 32 * b;
        if (q >= N) continue;
        // O^T[d][q]: lane (q, hi) holds d = (r&3) + 8(r>>2) + 4hi  -> four 8-byte stores of 4 consecutive channels
        const float inv = (DROP ? attn_drop_scale(a.drop) : 1.0f) / l;
        el16_t* op = a.out + ((size_t)n * N + q) * hd + h * 32;  // "b h (x y) d -> b (h d) x y"
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            uint2 w;
            w.x = pack_el16x2(o[b][g * 4 + 0] * inv, o[b][g * 4 + 1] * inv);
            w.y = pack_el16x2(o[b][g * 4 + 2] * inv, o[b][g * 4 + 3] * inv);
            *(uint2*)(op + 8 * g + 4 * hi) = w;
        }
    }
}

// ---- third form (round 4; its kernel, flash_attention3_kernel, was retired in round 5 -- the arithmetic described here lives on in
// the fourth form below, which is why the description stays): the second form's data flow with the vector-ALU work of the common path cut to what the softmax needs
// -- 16 exponentials, 8 packs and the normaliser sums per 16 scores of a lane.  Counters and the instruction stream of the second
// form (profiles/r03_flash_attention_pmc.txt; 95 vector instructions per 32 x 32 sub-tile at 86 % vector-issue occupancy) say
// where the rest went:
//   * 16 v_mov per sub-tile rebuilt the -m accumulator tuple          -> a THIRD matrix instruction delivers -m: k-slot 0 of a
//     32x32x16 product carries 1 on the key side and -m on the query side (m is kept exactly representable in the 16-bit type);
//   * 11 instructions per sub-tile found the maximum that is almost never needed (m is lazy) -> the check is made on the SUM of the
//     sub-tile's exponentials, which the normaliser needs anyway: a lane's 16 probabilities exceed 2^12 in sum only if one score
//     is more than 8 above m, and while the sum stays below it every probability is < 2^12 -- harmless in fp32 sums and in the
//     16-bit P operand.  The rare slow path (wave-uniform; always taken on the first sub-tile) recomputes the raw scores with two
//     matrix instructions, takes the exact maximum, rescales O and l and redoes the exponentials;
//   * 32 v_mov per 64-key tile copied the O accumulators between the register assignments of the three sub-tile variants that
//     shared one loop -> the main loop holds ONLY the whole-tile variant; a partial last tile runs after it through the general
//     sub-tile of the second form, and launches whose dropout layout the paired keep words cannot serve stay on the second form.

// ---- fourth form (round 4): the third form's arithmetic as a SOFTWARE PIPELINE over 32-key sub-tiles.  Timing experiments on the
// third form (tools/variants, wrong results: no bias product -4 %, no S^T products -17 %, no exponentials -12 %, no P V products
// -8 %, no staging / barriers -15 %, the last two together -27 %) say that matrix, vector and staging time ADD in that kernel: a
// wave issues in order, its softmax waits for its own S^T products and its P V products wait for its softmax.  Here
//   * the scores of sub-tile i + 1 are computed WHILE the exponentials of sub-tile i are issued -- the matrix instructions of a
//     wave sit between its own independent vector instructions (two score tuples alternate: the step is written once and
//     instantiated for A -> B and B -> A);
//   * K / V tiles live in a ring of THREE LDS buffers, so one barrier per 64-key tile is enough: tile t + 2 is written after the
//     barrier at the top of tile t, when every wave has left tile t - 1;
//   * V is staged ROW-major with one 16-byte write per thread, as two [64 keys][16 channels] images, and its transposed MFMA
//     fragments come from ds_read_b64_tr_b16 (a 16-lane group hands in the sixteen 8-byte pieces of a [4 keys][16 channels] block
//     and lane c receives column c): no 2-byte transposing writes, no V^T image;
//   * m stays EXACTLY 0 while the first sub-tile's maximum lies within +-8 (log2 domain) -- then no bias product is issued at all
//     (a wave-uniform flag selects the instantiation); the sum check still moves m when a later score exceeds it by more than 8;
//   * a workgroup of eight waves (256 queries) stages every tile once for twice the queries: waves 0-3 carry K, waves 4-7 V.
// When the lazy maximum moves (wave-uniform slow path, first sub-tile included) the scores already computed for sub-tile i + 1
// are shifted by the same amount.  One 32-query block per wave.
#ifndef FA4_MINW
#define FA4_MINW 4
#endif
constexpr int FA4_VH = 64 * 16 + 64;        // el16 elements between the two channel halves of the V image (128 B of padding: the
                                           // two 16-lane groups of a half-wave then read different bank halves)
constexpr int FA4_BUF = 64 * 32 + FA4_VH + 64 * 16;  // one K / V tile buffer: K [key][32 ch] (16-B chunk ^= (key >> 2) & 3), V images
typedef short fa_s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ el16x8_t fa4_vfrag(const el16_t* v) {  // keys {0..3} + {8..11} (relative) of this lane's channel
    typedef __attribute__((address_space(3))) fa_s16x4* lds_p;
    const fa_s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(v));
    const fa_s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_p)(v + 8 * 16));
    uint2 w[2] = {__builtin_bit_cast(uint2, lo), __builtin_bit_cast(uint2, hi)};
    return *(el16x8_t*)w;
}
#ifndef FA4_MFMA_SUM
#define FA4_MFMA_SUM 1
#endif
// sum of a lane's sixteen packed 16-bit values on the matrix pipe (see fa4_step)
__device__ __forceinline__ float fa4_rowsum(const uint32_t (&pk)[8]) {
    typedef float fa_f32x4 __attribute__((ext_vector_type(4)));
    fa_f32x4 acc = {0.0f, 0.0f, 0.0f, 0.0f};
#if DYF_F16
    typedef _Float16 fa_h4 __attribute__((ext_vector_type(4)));
    const fa_h4 ones = {(_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f, (_Float16)1.0f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint2 w = make_uint2(pk[2 * t], pk[2 * t + 1]);
        acc = __builtin_amdgcn_mfma_f32_4x4x4f16(ones, __builtin_bit_cast(fa_h4, w), acc, 0, 0, 0);
    }
#else
    const fa_bf16x4 ones = {0x3F80, 0x3F80, 0x3F80, 0x3F80};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const uint2 w = make_uint2(pk[2 * t], pk[2 * t + 1]);
        acc = __builtin_amdgcn_mfma_f32_4x4x4bf16_1k(ones, __builtin_bit_cast(fa_bf16x4, w), acc, 0, 0, 0);
    }
#endif
    return acc[0];
}
template <bool DROP, bool BIASED, bool TAIL>
__device__ __forceinline__ void fa4_step(const AttnArgs& a, fa_f32x16& sc_cur, fa_f32x16& sc_next, const el16_t* kc0, const el16_t* kc1,
                                         const el16_t* kn0, const el16_t* kn1, const el16_t* vc, const el16x8_t (&qf)[2], fa_f32x16& o, float& m,
                                         fa_f32x2& l2, el16x8_t& bm, const el16x8_t aone, bool& first, bool& biased, int jb, int q, int N, int hi,
                                         RngKey dkey) {  // q: DROP kernels pass the lane's Weyl value wq instead of the query index
    const fa_f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    const el16x8_t vf0 = fa4_vfrag(vc), vf1 = fa4_vfrag(vc + 16 * 16);
    if (TAIL) {  // partial last tile: the scores of THIS sub-tile are computed here, keys beyond the sequence masked
        sc_cur = DYF_MFMA_32x32x16(aone, bm, zero, 0, 0, 0);
        sc_cur = DYF_MFMA_32x32x16(*(const el16x8_t*)kc0, qf[0], sc_cur, 0, 0, 0);
        sc_cur = DYF_MFMA_32x32x16(*(const el16x8_t*)kc1, qf[1], sc_cur, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (jb + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) sc_cur[r] = -1.0e30f;
    }
    float p[16];
#pragma unroll
#if defined(FA4_X_NOEXP)      // timing experiments (wrong results)
    for (int r = 0; r < 16; ++r) p[r] = sc_cur[r] + 1.0f;
#elif defined(FA4_X_MFMAONLY)
    for (int r = 0; r < 16; ++r) p[r] = sc_cur[r];
#else
    for (int r = 0; r < 16; ++r) p[r] = __builtin_amdgcn_exp2f(sc_cur[r]);
#endif
    if (!TAIL) {
        const el16x8_t k0 = *(const el16x8_t*)kn0, k1 = *(const el16x8_t*)kn1;
#if defined(FA4_X_NOQK)
#pragma unroll
        for (int r = 0; r < 16; ++r) sc_next[r] = -m + (float)k0[0] + (float)k1[0];
#else
        if (BIASED) {
            sc_next = DYF_MFMA_32x32x16(aone, bm, zero, 0, 0, 0);  // -m[q] in every (key, q) entry
            sc_next = DYF_MFMA_32x32x16(k0, qf[0], sc_next, 0, 0, 0);
        } else {
            sc_next = DYF_MFMA_32x32x16(k0, qf[0], zero, 0, 0, 0);
        }
        sc_next = DYF_MFMA_32x32x16(k1, qf[1], sc_next, 0, 0, 0);
#endif
    }
#if FA4_MFMA_SUM
    // no dropout: the sub-tile's sum comes from the MATRIX pipe -- four 4x4x4 products of the packed probabilities with a tile of
    // ones give every lane the sum of its own sixteen values (block b = lane / 4, column j = lane % 4: D[b][i][j] = sum_k B[b][k][j]
    // for every row i), eight packed vector adds fewer per sub-tile; the sum is that of the ROUNDED probabilities, which is what
    // the P V products use
    uint32_t pk[8];
    fa_f32x2 ts;
    if (!DROP) {
#pragma unroll
        for (int t = 0; t < 8; ++t) pk[t] = pack_el16x2(p[2 * t], p[2 * t + 1]);
        ts = fa_f32x2{fa4_rowsum(pk), 0.0f};
    } else {
        ts = fa_f32x2{p[0], p[1]};
#pragma unroll
        for (int r = 2; r < 16; r += 2) ts += fa_f32x2{p[r], p[r + 1]};
    }
#elif FA4_SCALAR_SUM  // experiment: four scalar chains instead of packed adds
    float s0 = p[0] + p[4], s1 = p[1] + p[5], s2 = p[2] + p[6], s3 = p[3] + p[7];
    s0 += p[8]; s1 += p[9]; s2 += p[10]; s3 += p[11];
    s0 += p[12]; s1 += p[13]; s2 += p[14]; s3 += p[15];
    fa_f32x2 ts = fa_f32x2{s0 + s2, s1 + s3};
#else
    fa_f32x2 ts = fa_f32x2{p[0], p[1]};
#if !defined(FA4_X_MFMAONLY)
#pragma unroll
    for (int r = 2; r < 16; r += 2) ts += fa_f32x2{p[r], p[r + 1]};
#endif
#endif
    const float tsum = ts.x + ts.y;
    // wave-uniform slow path; rare after the first sub-tiles.  (An un-biased instantiation entered with m != 0 -- the step right
    // after the one that moved m away from 0 -- also comes here: its next scores were computed without the bias product.)
    if (first || (!BIASED && biased) || __builtin_amdgcn_ballot_w64(!(tsum <= 4096.0f)) != 0ull) {
        fa_f32x16 raw = DYF_MFMA_32x32x16(*(const el16x8_t*)kc0, qf[0], zero, 0, 0, 0);  // raw scores of THIS sub-tile
        raw = DYF_MFMA_32x32x16(*(const el16x8_t*)kc1, qf[1], raw, 0, 0, 0);
        if (TAIL) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (jb + (r & 3) + 8 * (r >> 2) + 4 * hi >= N) raw[r] = -1.0e30f;
        }
        float t = fmaxf(fmaxf(raw[0], raw[1]), raw[2]);
#pragma unroll
        for (int r = 3; r < 15; r += 2) t = fmaxf(fmaxf(t, raw[r]), raw[r + 1]);
        t = fmaxf(t, raw[15]);
        t = fmaxf(t, __shfl_xor(t, 32, 64));  // both lanes of a query agree on the new maximum
        if (!first) t = fmaxf(t, m);
        // m is exactly 0 while the maximum lies within +-8; otherwise it travels as a 16-bit matrix operand: clamped into the
        // type's finite range and rounded to nearest even
        const float mn = fabsf(t) <= 8.0f ? 0.0f : el16_to_f32(f32_to_el16(fminf(fmaxf(t, -60000.0f), 60000.0f)));
        const float delta = mn - m;
        const float alpha = first ? 1.0f : __builtin_amdgcn_exp2f(-delta);  // (first: O and l are still zero)
        m = mn;
        bm[0] = __builtin_bit_cast(el16_native_t, (el16_t)(hi == 0 ? f32_to_el16(-mn) : (el16_t)0));
        biased = __builtin_amdgcn_ballot_w64(mn != 0.0f) != 0ull;
        l2 *= alpha;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            o[r] *= alpha;
            p[r] = __builtin_amdgcn_exp2f(raw[r] - mn);
            if (!TAIL) sc_next[r] -= BIASED ? delta : mn;  // computed with the old -m / without a bias product
        }
#if FA4_MFMA_SUM
        if (!DROP) {
#pragma unroll
            for (int t = 0; t < 8; ++t) pk[t] = pack_el16x2(p[2 * t], p[2 * t + 1]);
            ts = fa_f32x2{fa4_rowsum(pk), 0.0f};
        } else
#endif
        {
            ts = fa_f32x2{p[0], p[1]};
#pragma unroll
            for (int r = 2; r < 16; r += 2) ts += fa_f32x2{p[r], p[r + 1]};
        }
        first = false;
    }
    l2 += ts;  // the normaliser is accumulated BEFORE dropout (attention.py:69-70)
    if (DROP) {
        const uint32_t th = a.drop.thresh8;
        // QUAD form (common.h rng_keep8): element q * N + jb + 4 hi is a multiple of 4 (N % 4 == 0) -> quad index / 4; what arrives as
        // `q` is the Weyl value of the lane's quad (q * N + 4 hi) / 4, the sub-tile adds (jb / 4) * RNG_WEYL (scalar) and register group g
        // (keys jb + 4 hi + 8 g + {0..3}) the constant 2 g * RNG_WEYL: FOUR hashes per 16 probabilities (eight in the pair form)
        const uint32_t w0 = (uint32_t)q + (uint32_t)(jb >> 2) * RNG_WEYL;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t w = rng_pair_mix(w0 + (2u * (uint32_t)g) * RNG_WEYL, dkey);
            p[4 * g] = (w & 0xffu) < th ? p[4 * g] : 0.0f;
            p[4 * g + 1] = ((w >> 8) & 0xffu) < th ? p[4 * g + 1] : 0.0f;
            p[4 * g + 2] = ((w >> 16) & 0xffu) < th ? p[4 * g + 2] : 0.0f;
            p[4 * g + 3] = (w >> 24) < th ? p[4 * g + 3] : 0.0f;
        }
    }
#if FA4_MFMA_SUM
    if (DROP) {
#pragma unroll
        for (int t = 0; t < 8; ++t) pk[t] = pack_el16x2(p[2 * t], p[2 * t + 1]);
    }
#else
    uint32_t pk[8];
#pragma unroll
#if defined(FA4_X_MFMAONLY)
    for (int t = 0; t < 8; ++t) pk[t] = __builtin_bit_cast(uint32_t, p[2 * t]);
#else
    for (int t = 0; t < 8; ++t) pk[t] = pack_el16x2(p[2 * t], p[2 * t + 1]);
#endif
#endif
#if defined(FA4_X_NOPV)
    o[0] += __builtin_bit_cast(float, pk[0] ^ pk[1] ^ pk[2] ^ pk[3]) + (float)vf0[0];
    o[1] += __builtin_bit_cast(float, pk[4] ^ pk[5] ^ pk[6] ^ pk[7]) + (float)vf1[0];
#else
    {
        uint32_t pw[4] = {pk[0], pk[1], pk[2], pk[3]};
        o = DYF_MFMA_32x32x16(vf0, *(el16x8_t*)pw, o, 0, 0, 0);
    }
    {
        uint32_t pw[4] = {pk[4], pk[5], pk[6], pk[7]};
        o = DYF_MFMA_32x32x16(vf1, *(el16x8_t*)pw, o, 0, 0, 0);
    }
#endif
}

// Launch contract: DROP kernels need a.drop.mode == 1, an even token count and whole query blocks (launch_attention checks).
// NW: waves per workgroup (4 or 8).
template <bool DROP, int NW>
__global__ __launch_bounds__(64 * NW, FA4_MINW) void flash_attention4_kernel(AttnArgs a) {
    __shared__ __attribute__((aligned(16))) el16_t KV[3 * FA4_BUF];
    constexpr int QG = 32 * NW;
    const int qblocks = (a.hw + QG - 1) / QG;
    int bh = blockIdx.x / qblocks, qb = blockIdx.x % qblocks;
    if ((a.n * a.heads) % 8 == 0) {  // (sample, head) bh pinned to XCD bh % 8 (see the second form)
        const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
        bh = (k / qblocks) * 8 + xcd;
        qb = k % qblocks;
    }
    const int n = bh / a.heads, h = bh % a.heads;
    const int C3 = 3 * a.heads * 32, hd = a.heads * 32, N = a.hw;
    const el16_t* base = a.qkv + (size_t)n * N * C3;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
    const int q0 = qb * QG + wave * 32 + l31;
    const float c2 = 0.17677669529663687f * 1.4426950408889634f;  // 32^-1/2 * log2(e): scores live in the log2 domain
    el16x8_t qf[2];  // Q fragments (B operand of S^T), pre-multiplied by c2: lane (q, hi) holds channels ks*16 + hi*8 .. +8
    fa_f32x16 o, scA, scB;
    float m = 0.0f;
    fa_f32x2 l2 = fa_f32x2{0.0f, 0.0f};
    el16x8_t bm, aone;   // operands of the bias product: k-slot 0 = -m of the lane's query / 1 for every key
    const int kf0 = l31 * 32 + (((0 + hi) ^ ((l31 >> 2) & 3)) << 3);   // this lane's fragment offsets inside a buffer
    const int kf1 = l31 * 32 + (((2 + hi) ^ ((l31 >> 2) & 3)) << 3);
    // V: lane (d = l31, hi) belongs to the 16-lane group (channel half d >> 4, hi) and hands in piece i = d & 15 of the group's
    // [4 keys][16 channels] block: key row 4 hi + (i >> 2), channels 4 (i & 3) ..
    const int vf0 = 64 * 32 + (l31 >> 4) * FA4_VH + (4 * hi + ((l31 & 15) >> 2)) * 16 + (l31 & 3) * 4;
#pragma unroll
    for (int t = 0; t < 8; ++t) aone[t] = bm[t] = __builtin_bit_cast(el16_native_t, (el16_t)0);
    aone[0] = __builtin_bit_cast(el16_native_t, (el16_t)(hi == 0 ? f32_to_el16(1.0f) : (el16_t)0));
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
        uint4 v = make_uint4(0, 0, 0, 0);
        if (q0 < N) v = *(const uint4*)(base + (size_t)q0 * C3 + h * 32 + ks * 16 + hi * 8);
        uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int t = 0; t < 4; ++t) w[t] = pack_el16x2(el16_lo(w[t]) * c2, el16_hi(w[t]) * c2);
        qf[ks] = *(el16x8_t*)w;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) o[r] = scA[r] = scB[r] = 0.0f;
    bool first = true, biased = false;
    const RngKey dkey = DROP ? attn_drop_key(a.drop, n, (uint32_t)h) : RngKey{0u, 0u};
    // DROP: what the steps receive as "q" is the Weyl value of the lane's first keep-word pair, (q0 * N + 4 hi) / 2
    const int qarg = DROP ? (int)rng_weyl(((uint32_t)q0 * (uint32_t)N + 4u * (uint32_t)hi) >> 2, dkey) : q0;

    // staging role: thread -> (key, 16-B chunk of 8 channels); NW = 4: every thread carries its K and its V piece, NW = 8: the
    // threads of waves 0-3 carry K, those of waves 4-7 V.  Rows beyond the sequence are CLAMPED to the last row (finite values;
    // their scores are masked in the tail step and their tiles never read in the main loop).
    const int st_id = tid & 255, skey = st_id >> 2, sch = st_id & 3;
    const bool vrole = NW >= 8 && (tid & 511) >= 256;
    const int kdst = skey * 32 + ((sch ^ ((skey >> 2) & 3)) << 3);
    const int vdst = 64 * 32 + (sch >> 1) * FA4_VH + skey * 16 + (sch & 1) * 8;
    const int sdst = vrole ? vdst : kdst;
    const el16_t* gk = base + hd + h * 32 + sch * 8 + (vrole ? hd : 0);
    uint4 kv_n = make_uint4(0, 0, 0, 0), vv_n = make_uint4(0, 0, 0, 0);
    const bool stager = NW <= 8 || tid < 512;  // NW = 16: waves 0-7 stage
    auto fetch = [&](int j0) {  // K / V of a later 64-key tile travel through registers while the current one is consumed
        const uint32_t row = (uint32_t)min(j0 + skey, N - 1);
        if (stager) kv_n = *(const uint4*)(gk + row * (uint32_t)C3);
        if (NW == 4) vv_n = *(const uint4*)(gk + hd + row * (uint32_t)C3);
    };
    auto stage = [&](int buf) {
        if (NW == 4) {
            *(uint4*)(KV + buf + kdst) = kv_n;
            *(uint4*)(KV + buf + vdst) = vv_n;
        } else if (stager) {
            *(uint4*)(KV + buf + sdst) = kv_n;
        }
    };
    int c0 = 0, c1 = FA4_BUF, c2b = 2 * FA4_BUF;  // ring: tile t, t + 1, t + 2
    fetch(0);
    stage(c0);
    fetch(64);
    stage(c1);
    fetch(128);
    __syncthreads();
    const int T = N / 64;  // whole tiles
    if (T > 0) {           // scores of (tile 0, sub-tile 0), raw (m = 0): the first step's slow path takes them from there
        const fa_f32x16 zero = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        scA = DYF_MFMA_32x32x16(*(const el16x8_t*)(KV + c0 + kf0), qf[0], zero, 0, 0, 0);
        scA = DYF_MFMA_32x32x16(*(const el16x8_t*)(KV + c0 + kf1), qf[1], scA, 0, 0, 0);
    }
    // one 64-key tile; B: instantiation with / without bias products (a macro, not a lambda: nested closures kept their
    // captures in scratch memory)
#if !defined(FA4_X_NOSTAGE)
#define FA4_TILE_STAGE()                                                                                          \
    if (t > 0) __syncthreads(); /* every wave has left tile t - 1; tile t + 1 (written one tile ago) is visible */ \
    stage(c2b);                 /* tile t + 2 */                                                                   \
    fetch(64 * (t + 3));
#else
#define FA4_TILE_STAGE()
#endif
    // sub-tile 0: next = (t, 1) in the same buffer; sub-tile 1: next = (t + 1, 0) (its scores are dropped after the last tile)
#define FA4_TILE(B)                                                                                                                       \
    {                                                                                                                                     \
        FA4_TILE_STAGE()                                                                                                                  \
        const int j0 = 64 * t;                                                                                                            \
        const el16_t* B0 = KV + c0;                                                                                                       \
        const el16_t* B1 = KV + c1;                                                                                                       \
        fa4_step<DROP, B, false>(a, scA, scB, B0 + kf0, B0 + kf1, B0 + 1024 + kf0, B0 + 1024 + kf1, B0 + vf0, qf, o, m, l2, bm, aone,    \
                                 first, biased, j0, qarg, N, hi, dkey);                                                                   \
        fa4_step<DROP, B, false>(a, scB, scA, B0 + 1024 + kf0, B0 + 1024 + kf1, B1 + kf0, B1 + kf1, B0 + vf0 + 32 * 16, qf, o, m, l2, bm, \
                                 aone, first, biased, j0 + 32, qarg, N, hi, dkey);                                                        \
        const int tmp = c0;                                                                                                               \
        c0 = c1;                                                                                                                          \
        c1 = c2b;                                                                                                                         \
        c2b = tmp;                                                                                                                        \
    }
    int t = 0;
    for (; t < T && !biased; ++t) FA4_TILE(false)  // every m of the wave is 0: no bias products
    for (; t < T; ++t) FA4_TILE(true)
#undef FA4_TILE
#undef FA4_TILE_STAGE
    if (!DROP && 64 * T < N) {  // partial last tile (ring slot c0)
        if (T > 0) __syncthreads();
        const int j0 = 64 * T;
        const el16_t* B0 = KV + c0;
        fa4_step<false, true, true>(a, scA, scB, B0 + kf0, B0 + kf1, B0 + kf0, B0 + kf1, B0 + vf0, qf, o, m, l2, bm, aone, first, biased, j0, q0, N, hi, dkey);
        if (j0 + 32 < N)
            fa4_step<false, true, true>(a, scA, scB, B0 + 1024 + kf0, B0 + 1024 + kf1, B0 + kf0, B0 + kf1, B0 + vf0 + 32 * 16, qf, o, m, l2, bm, aone, first, biased, j0 + 32, q0, N, hi, dkey);
    }
    const float ll = l2.x + l2.y;
    const float l = ll + __shfl_xor(ll, 32, 64);
    if (q0 >= N) return;
    // O^T[d][q]: lane (q, hi) holds d = (r&3) + 8(r>>2) + 4hi  -> four 8-byte stores of 4 consecutive channels
    const float inv = (DROP ? attn_drop_scale(a.drop) : 1.0f) / l;
    el16_t* op = a.out + ((size_t)n * N + q0) * hd + h * 32;  // "b h (x y) d -> b (h d) x y"
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        uint2 w;
        w.x = pack_el16x2(o[g * 4 + 0] * inv, o[g * 4 + 1] * inv);
        w.y = pack_el16x2(o[g * 4 + 2] * inv, o[g * 4 + 3] * inv);
        *(uint2*)(op + 8 * g + 4 * hi) = w;
    }
}

hipError_t launch_attention(const AttnArgs& a, hipStream_t s) {
    // DYF_FLASH_ATTN: unset / 4 (/ 3: that kernel was retired in round 5) = flash_attention4_kernel (the pipelined form; falls back to
    // flash_attention2_kernel for the dropout layouts it does not take), 2 = flash_attention2_kernel, 1 = the first flash form,
    // 0 = the plain per-query kernel
    const int flash = dyf_form("DYF_FLASH_ATTN") ? atoi(dyf_form("DYF_FLASH_ATTN")) : 4;
    if (flash != 0 && a.hw <= 65535) {
        const int qblocks = (a.hw + 127) / 128;
        // 64 queries per wave from 512 tokens on (DYF_FLASH_QB=1 keeps 32): shorter sequences would leave CUs without a workgroup
        const int qb_env = dyf_form("DYF_FLASH_QB") ? atoi(dyf_form("DYF_FLASH_QB")) : 2;
        // with dropout on the probabilities: DYF_FLASH_QB_DROP=2 selects the 64-query form (the second form spilled there)
        const int qb_drop = dyf_form("DYF_FLASH_QB_DROP") ? atoi(dyf_form("DYF_FLASH_QB_DROP")) : 1;
        const bool drop = a.drop.mode != 0;
        const bool qb2 = flash != 1 && a.hw >= 512 && (drop ? qb_drop == 2 && flash >= 3 : qb_env == 2);
        const int qblocks2 = (a.hw + 255) / 256;
        // the fourth form: one 32-query block per wave, DYF_FLASH_NW = 8 (default) / 4 waves per workgroup; a sequence shorter than
        // 512 tokens stays on four waves (more workgroups)
        const int nw_env = dyf_form("DYF_FLASH_NW") ? atoi(dyf_form("DYF_FLASH_NW")) : 8;
        const int nw = nw_env == 16 && a.hw >= 2048 ? 16 : nw_env >= 8 && a.hw >= 512 ? 8 : 4;
        const bool v4 = flash >= 3 && (!drop || (a.drop.mode == 1 && (a.hw & 3) == 0 && a.hw % (32 * nw) == 0));
        if (v4) {
            dyf_form_note(nw == 16 ? "flash_attention4_kernel<NW=16>" : nw == 8 ? "flash_attention4_kernel<NW=8>" : "flash_attention4_kernel<NW=4>", a.n);
            const int qb4 = (a.hw + 32 * nw - 1) / (32 * nw);
            if (nw == 16 && drop) hipLaunchKernelGGL((flash_attention4_kernel<true, 16>), dim3(a.n * a.heads * qb4), dim3(1024), 0, s, a);
            else if (nw == 16) hipLaunchKernelGGL((flash_attention4_kernel<false, 16>), dim3(a.n * a.heads * qb4), dim3(1024), 0, s, a);
            else if (nw == 8 && drop) hipLaunchKernelGGL((flash_attention4_kernel<true, 8>), dim3(a.n * a.heads * qb4), dim3(512), 0, s, a);
            else if (nw == 8) hipLaunchKernelGGL((flash_attention4_kernel<false, 8>), dim3(a.n * a.heads * qb4), dim3(512), 0, s, a);
            else if (drop) hipLaunchKernelGGL((flash_attention4_kernel<true, 4>), dim3(a.n * a.heads * qb4), dim3(256), 0, s, a);
            else hipLaunchKernelGGL((flash_attention4_kernel<false, 4>), dim3(a.n * a.heads * qb4), dim3(256), 0, s, a);
            return hipGetLastError();
        }
        const bool qb2v2 = qb2 && !drop;
        dyf_form_note(flash == 1 ? "flash_attention_kernel" : qb2v2 ? "flash_attention2_kernel<QB=2>" : "flash_attention2_kernel<QB=1>", a.n);
        if (flash == 1) hipLaunchKernelGGL(flash_attention_kernel, dim3(a.n * a.heads * qblocks), dim3(256), 0, s, a);
        else if (qb2v2) hipLaunchKernelGGL((flash_attention2_kernel<false, 2>), dim3(a.n * a.heads * qblocks2), dim3(256), 0, s, a);
        else if (drop) hipLaunchKernelGGL((flash_attention2_kernel<true, 1>), dim3(a.n * a.heads * qblocks), dim3(256), 0, s, a);
        else hipLaunchKernelGGL((flash_attention2_kernel<false, 1>), dim3(a.n * a.heads * qblocks), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    const int qtiles = (a.hw + 63) / 64;
    hipLaunchKernelGGL(attention_kernel, dim3(a.n * a.heads * qtiles), dim3(64), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ head / upsample
__global__ void head_kernel(HeadArgs a) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)a.n * a.hw;
    if (idx >= total) return;
    const int n = (int)(idx / a.hw), p = (int)(idx % a.hw);
    const el16_t* x = a.x + (size_t)idx * a.c;
    if ((a.c & 7) == 0 && a.cout <= 4) {  // 16-byte loads of the pixel's channels, all outputs accumulated in one pass
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int c0 = 0; c0 < a.c; c0 += 8) {
            const uint4 q = *(const uint4*)(x + c0);
            const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const float v = (t & 1) ? el16_hi(qw[t >> 1]) : el16_lo(qw[t >> 1]);
#pragma unroll
                for (int co = 0; co < 4; ++co)
                    if (co < a.cout) acc[co] = fmaf(v, a.wgt[(size_t)co * a.c + c0 + t], acc[co]);
            }
        }
#pragma unroll
        for (int co = 0; co < 4; ++co)
            if (co < a.cout) a.out[((size_t)n * a.cout + co) * a.hw + p] = acc[co] + a.bias[co];
        return;
    }
    for (int co = 0; co < a.cout; ++co) {
        float acc = a.bias[co];
        const float* w = a.wgt + (size_t)co * a.c;
        for (int c = 0; c < a.c; ++c) acc = fmaf(el16_to_f32(x[c]), w[c], acc);
        a.out[((size_t)n * a.cout + co) * a.hw + p] = acc;
    }
}

// c % 8 == 0, c/8 a power of two <= 64, cout <= 4: every lane owns ONE 16-byte channel chunk (its weights stay in registers)
// and walks pixels; the c/8 partial dot products of a pixel are summed across neighbouring lanes.  Loads are fully coalesced
// (the per-pixel form above strides lanes by a whole pixel: 64 cache lines per load instruction).
__global__ __launch_bounds__(256) void head_vec_kernel(HeadArgs a) {
    const int chunks = a.c >> 3, q = threadIdx.x & (chunks - 1), row = threadIdx.x / chunks, rows = 256 / chunks;
    float w[4][8];
#pragma unroll
    for (int co = 0; co < 4; ++co)
#pragma unroll
        for (int t = 0; t < 8; ++t) w[co][t] = co < a.cout ? a.wgt[(size_t)co * a.c + q * 8 + t] : 0.0f;
    const long long total = (long long)a.n * a.hw;
    for (long long p = (long long)blockIdx.x * rows + row; p < total; p += (long long)gridDim.x * rows) {
        const uint4 v = *(const uint4*)(a.x + (size_t)p * a.c + q * 8);
        const uint32_t qw[4] = {v.x, v.y, v.z, v.w};
        float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float xv = (t & 1) ? el16_hi(qw[t >> 1]) : el16_lo(qw[t >> 1]);
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] = fmaf(xv, w[co][t], acc[co]);
        }
        for (int d = 1; d < chunks; d <<= 1)
#pragma unroll
            for (int co = 0; co < 4; ++co) acc[co] += __shfl_xor(acc[co], d, 64);
        if (q == 0) {
            const int n = (int)(p / a.hw), pp = (int)(p - (long long)n * a.hw);
#pragma unroll
            for (int co = 0; co < 4; ++co)
                if (co < a.cout) a.out[((size_t)n * a.cout + co) * a.hw + pp] = acc[co] + a.bias[co];
        }
    }
}

hipError_t launch_head(const HeadArgs& a, hipStream_t s) {
    const long long total = (long long)a.n * a.hw;
    const int chunks = a.c >> 3;
    if ((a.c & 7) == 0 && chunks >= 1 && chunks <= 64 && (chunks & (chunks - 1)) == 0 && a.cout <= 4) {
        const int rows = 256 / chunks;
        const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((total + rows * 4 - 1) / (rows * 4), 4096));
        hipLaunchKernelGGL(head_vec_kernel, dim3(grid), dim3(256), 0, s, a);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

__global__ void up2x_nearest_kernel(const el16_t* src, int n, int h, int w, int c, el16_t* out, long long total) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int ch = (int)(idx % c);
    const long long pix = idx / c;
    const int ow = 2 * w, oh = 2 * h;
    const int ni = (int)(pix / ((long long)oh * ow));
    const int rem = (int)(pix % ((long long)oh * ow));
    const int y = rem / ow, x = rem % ow;
    out[idx] = src[(((size_t)ni * h + (y >> 1)) * w + (x >> 1)) * c + ch];
}

// c % 8 == 0: one lane per 16-byte chunk of one SOURCE pixel, written to its four output pixels (one read, four writes)
__global__ __launch_bounds__(256) void up2x_nearest_vec_kernel(const uint4* src, int h, int w, int c8, uint4* out, unsigned total) {
    const unsigned idx = blockIdx.x * 256u + threadIdx.x;
    if (idx >= total) return;
    const unsigned ch = idx % (unsigned)c8, pix = idx / (unsigned)c8;
    const unsigned x = pix % (unsigned)w, row = pix / (unsigned)w;  // row = n*h + y
    const uint4 v = src[idx];
    const size_t o = ((size_t)row * 2 * (2 * w) + 2 * x) * c8 + ch;
    const size_t rs = (size_t)(2 * w) * c8;
    out[o] = v;
    out[o + c8] = v;
    out[o + rs] = v;
    out[o + rs + c8] = v;
}

// Dropout on a whole NHWC 16-bit tensor (unet.Unet's dropout_input / dropout_input_for_residual on init_conv's output, unet.py:276-277;
// only with input_dropout > 0, which no shipped config sets): two elements per thread, one keep word per pair.
__global__ __launch_bounds__(256) void drop16_kernel(const el16_t* x, el16_t* y, long long pairs_total, uint32_t pairs_per_row, DropSpec d) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= pairs_total) return;
    const int n = (int)(i / pairs_per_row);
    const uint32_t e0 = (uint32_t)(i - (long long)n * pairs_per_row) * 2u;
    const RngKey key = drop_row_key(d, n);
    const uint32_t row0 = (uint32_t)n * pairs_per_row * 2u;
    float v[2] = {el16_to_f32(x[2 * i]), el16_to_f32(x[2 * i + 1])};
    act_drop_mode<2, ACT_NONE>(v, row0 + e0, row0, d, key);
    *(uint32_t*)(y + 2 * i) = pack_el16x2(v[0], v[1]);
}

hipError_t launch_drop16(const el16_t* x, el16_t* y, int n, long long per_row, const DropSpec& d, hipStream_t s) {
    if (per_row % 2 != 0 || per_row / 2 > 0xFFFFFFFFll) return hipErrorInvalidValue;
    const long long pairs = (long long)n * per_row / 2;
    hipLaunchKernelGGL(drop16_kernel, dim3((unsigned)((pairs + 255) / 256)), dim3(256), 0, s, x, y, pairs, (uint32_t)(per_row / 2), d);
    return hipGetLastError();
}

hipError_t launch_up2x_nearest(const el16_t* src, int n, int h, int w, int c, el16_t* out, hipStream_t s) {
    const long long total = (long long)n * 4 * h * w * c;
    if (c % 8 == 0 && total / 32 < 0xFFFFFFFFll) {
        const unsigned tv = (unsigned)((long long)n * h * w * (c / 8));
        KernelProf kp("up2x_nearest_vec_kernel", s, (double)n * h * w * c * 2.0 * 5.0);  // read once, write 4x
        hipLaunchKernelGGL(up2x_nearest_vec_kernel, dim3((tv + 255) / 256), dim3(256), 0, s, (const uint4*)src, h, w, c / 8,
                           (uint4*)out, tv);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(up2x_nearest_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, c, out,
                       total);
    return hipGetLastError();
}
