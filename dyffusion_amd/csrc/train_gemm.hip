// fp32 implicit-GEMM convolutions of the training step on the matrix cores (gfx950 `v_mfma_f32_32x32x2_f32`): forward,
// data gradient and weight gradient of `nn.Conv2d` on fp32 NHWC tensors (csrc/train.hip: conv_fwd / conv_dgrad / conv_wgrad;
// the reference gets them from torch.autograd over src/models/unet_simple.py:29-56).  Operands stay fp32 end to end, so the
// gradient parity with autograd over the oracle stays at the 1e-6 level of the plain VALU kernels these replace (only the
// summation order differs); fp32 MFMA peaks at 157 TFLOP/s on MI355X, the VALU kernels ran at 4-6.
//
// One kernel, three gathers.  C[M][N] = sum_k A(m, k) * B(k, n), workgroup tile 128 x 64, K stage 16, 4 waves of 64 x 32
// (two 32 x 32 accumulators, one B fragment feeds both):
//   forward   M = output pixels, N = cout, K = (tap, ci):  A = x at the tap's shifted pixel (zero outside), B = wt[tap][ci][co]
//   dgrad     M = input pixels,  N = cin,  K = (tap, co):  A = dz at ((iy + p - ky) / s, ..) where divisible, B = w[co][tap][ci]
//   wgrad     M = cout, N = cin, K = output pixels (one tap per workgroup, pixel range split over workgroups, merged with
//             atomics like the kernel it replaces):  A = dz[pixel][co], B = x at the tap's shifted pixel
// Both operand tiles sit in LDS as [row][16 k] with k permuted to [k even | k odd] (a lane of the 32x32x2 MFMA needs
// k = 2j + (lane >> 5) for j = 0..7: two ds_read_b128), rows 20 floats apart (conflict-free for the b128 lane groups).
// Global loads are 16-byte vectors along the contiguous axis of each operand (channels), issued for stage s + 1 before the
// MFMAs of stage s.
#include "train_internal.h"

#include <cstdlib>
#include <cstring>

typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {

constexpr int GM = 128, GN = 64, GK = 16, LDR = 20;
enum { TG_FWD = 0, TG_DGRAD = 1, TG_WGRAD = 2 };

template <int MODE>
__global__ __launch_bounds__(256) void t_gemm_mfma(dyf::TConv g, const float* __restrict__ Ap, const float* __restrict__ Bp,
                                                   const float* __restrict__ bias, float* __restrict__ Cp, int split_len) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) float As[2][GM * LDR];
    __shared__ __attribute__((aligned(16))) float Bs[2][GN * LDR];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x, tn = blockIdx.y;
    const int taps = g.k * g.k;
    const long long opix = (long long)g.n * g.ho * g.wo, ipix = (long long)g.n * g.h * g.w;
    const long long M = MODE == TG_FWD ? opix : MODE == TG_DGRAD ? ipix : g.cout;
    const int CK = MODE == TG_FWD ? g.cin : g.cout;  // channels per tap on the K axis (forward / dgrad)
    int tap = 0;
    long long kbeg = 0, kend = 0;  // wgrad: pixel range of this workgroup
    int nstage, st0 = 0;  // stages [st0, st0 + nstage) of the K axis
    if (MODE == TG_WGRAD) {
        tap = blockIdx.z % taps;
        kbeg = (long long)(blockIdx.z / taps) * split_len;
        kend = kbeg + split_len < opix ? kbeg + split_len : opix;
        nstage = (int)((kend - kbeg + GK - 1) / GK);
    } else {
        // split_len > 0: split-K over gridDim.z workgroups of split_len stages each (layers with few output tiles and a deep K:
        // the 4 x 4 ... 16 x 16 planes at small batches); the partial sums go to a workspace and are merged in a fixed order
        // (merging with atomics made the ACTIVATIONS differ in the last bit from run to run, which a (Leaky)ReLU near zero
        // turns into a different derivative now and then: gradients moved by up to 5e-4 of their norm between identical runs)
        const int total = taps * CK / GK;
        st0 = split_len > 0 ? (int)blockIdx.z * split_len : 0;
        nstage = split_len > 0 ? min(split_len, total - st0) : total;
    }

    // ---- per-thread load slots
    // A, forward / dgrad: 2 slots (row = s >> 2, k quad = s & 3), float4 along k; wgrad: 2 slots (k = s >> 5, m quad = s & 31)
    int a_b[2], a_y[2], a_x[2];
    bool a_ok[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int s = tid + 256 * i;
        a_b[i] = a_y[i] = a_x[i] = 0;
        a_ok[i] = false;
        if (MODE != TG_WGRAD) {
            const long long m = (long long)tm * GM + (s >> 2);
            a_ok[i] = m < M;
            const int pw = MODE == TG_FWD ? g.wo : g.w, ph = MODE == TG_FWD ? g.ho : g.h;
            const long long mm = a_ok[i] ? m : 0;
            a_x[i] = (int)(mm % pw);
            a_y[i] = (int)((mm / pw) % ph);
            a_b[i] = (int)(mm / ((long long)pw * ph));
        }
    }
    const int b_k = tid >> 4, b_nq = tid & 15;  // B: k row, n quad
    // dgrad: stride a power of two (every shipped layer) -> shifts instead of two divisions by a run-time stride per slot and stage
    const int sh = (g.s & (g.s - 1)) == 0 ? __builtin_ctz((unsigned)g.s) : -1;
    // wgrad: (ox, oy, b) of this thread's B pixel, carried from stage to stage (the stages of a launch are loaded in order) instead
    // of three 64-bit divisions per stage
    int w_ox = 0, w_oy = 0, w_b = 0;
    if (MODE == TG_WGRAD) {
        const unsigned pix0 = (unsigned)(kbeg + b_k);
        w_ox = (int)(pix0 % (unsigned)g.wo);
        w_oy = (int)((pix0 / (unsigned)g.wo) % (unsigned)g.ho);
        w_b = (int)(pix0 / ((unsigned)g.wo * (unsigned)g.ho));
    }

    float4 ra[2], rb;
    auto load = [&](int stage) {
        if (MODE != TG_WGRAD) {
            const int k0 = stage * GK;
            const int tp = k0 / CK, c0 = k0 - tp * CK;
            const int ky = tp / g.k, kx = tp - ky * g.k;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int kq = (tid + 256 * i) & 3;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == TG_FWD) {
                    const int iy = a_y[i] * g.s - g.p + ky, ix = a_x[i] * g.s - g.p + kx;
                    if (a_ok[i] && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w)
                        ra[i] = *(const float4*)(Ap + (((size_t)a_b[i] * g.h + iy) * g.w + ix) * g.cin + c0 + kq * 4);
                } else {
                    const int ty = a_y[i] + g.p - ky, tx = a_x[i] + g.p - kx;
                    const int oy = sh >= 0 ? ty >> sh : ty / g.s, ox = sh >= 0 ? tx >> sh : tx / g.s;
                    if (a_ok[i] && ty >= 0 && tx >= 0 && oy * g.s == ty && ox * g.s == tx && oy < g.ho && ox < g.wo)
                        ra[i] = *(const float4*)(Ap + (((size_t)a_b[i] * g.ho + oy) * g.wo + ox) * g.cout + c0 + kq * 4);
                }
            }
            if (MODE == TG_FWD)
                rb = *(const float4*)(Bp + ((size_t)tp * g.cin + c0 + b_k) * g.cout + tn * GN + b_nq * 4);
            else
                rb = *(const float4*)(Bp + ((size_t)(c0 + b_k) * taps + tp) * g.cin + tn * GN + b_nq * 4);
        } else {
            const long long p0 = kbeg + (long long)stage * GK;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = tid + 256 * i;
                const long long pix = p0 + (s >> 5);
                const int m = tm * GM + (s & 31) * 4;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < kend && m < g.cout) ra[i] = *(const float4*)(Ap + (size_t)pix * g.cout + m);
            }
            const long long pix = p0 + b_k;
            rb = make_float4(0.f, 0.f, 0.f, 0.f);
            if (pix < kend) {
                const int ox = w_ox, oy = w_oy, b = w_b;
                const int ky = tap / g.k, kx = tap - ky * g.k;
                const int iy = oy * g.s - g.p + ky, ix = ox * g.s - g.p + kx;
                if ((unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w)
                    rb = *(const float4*)(Bp + (((size_t)b * g.h + iy) * g.w + ix) * g.cin + tn * GN + b_nq * 4);
            }
            w_ox += GK;  // the next stage's pixel
            while (w_ox >= g.wo) {
                w_ox -= g.wo;
                if (++w_oy >= g.ho) { w_oy = 0; ++w_b; }
            }
        }
    };
    // element k of a row lives at (k & 1) * 8 + (k >> 1)
    auto store = [&](int buf) {
        if (MODE != TG_WGRAD) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = tid + 256 * i;
                float* d = &As[buf][(s >> 2) * LDR + 2 * (s & 3)];  // k = 4 kq + e: e = 0, 2 -> even half; e = 1, 3 -> odd half
                *(float2*)d = make_float2(ra[i].x, ra[i].z);
                *(float2*)(d + 8) = make_float2(ra[i].y, ra[i].w);
            }
        } else {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = tid + 256 * i;
                const int k = s >> 5, kp = (k & 1) * 8 + (k >> 1);
                float* d = &As[buf][((s & 31) * 4) * LDR + kp];
                d[0] = ra[i].x; d[LDR] = ra[i].y; d[2 * LDR] = ra[i].z; d[3 * LDR] = ra[i].w;
            }
        }
        const int kp = (b_k & 1) * 8 + (b_k >> 1);
        float* d = &Bs[buf][(b_nq * 4) * LDR + kp];
        d[0] = rb.x; d[LDR] = rb.y; d[2 * LDR] = rb.z; d[3 * LDR] = rb.w;
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

    if (nstage > 0) {
        load(st0);
        store(0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) load(st0 + st + 1);
        const float* ar0 = &As[buf][(wm * 64 + l31) * LDR + hi * 8];
        const float* ar1 = ar0 + 32 * LDR;
        const float* br = &Bs[buf][(wn * 32 + l31) * LDR + hi * 8];
        const float4 a00 = *(const float4*)ar0, a01 = *(const float4*)(ar0 + 4);
        const float4 a10 = *(const float4*)ar1, a11 = *(const float4*)(ar1 + 4);
        const float4 b0 = *(const float4*)br, b1 = *(const float4*)(br + 4);
        const float af0[8] = {a00.x, a00.y, a00.z, a00.w, a01.x, a01.y, a01.z, a01.w};
        const float af1[8] = {a10.x, a10.y, a10.z, a10.w, a11.x, a11.y, a11.z, a11.w};
        const float bf[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(af0[j], bf[j], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(af1[j], bf[j], acc[1], 0, 0, 0);
        }
        if (st + 1 < nstage) store(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: lane (l31, hi), register r of accumulator i: row wm*64 + i*32 + 8*(r/4) + 4*hi + r%4, column wn*32 + l31
    const int n = tn * GN + wn * 32 + l31;
    const float bv = (MODE != TG_WGRAD && bias) ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = (long long)tm * GM + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (m >= M) continue;
            if (MODE == TG_WGRAD) {
                atomicAdd(Cp + ((size_t)m * taps + tap) * g.cin + n, acc[i][r]);
            } else {
                const int NC = MODE == TG_FWD ? g.cout : g.cin;
                // split-K: raw partial sums to Cp[split][m][n] (the workspace); t_splitk_finish adds them in split order
                if (split_len > 0) Cp[((size_t)blockIdx.z * M + m) * NC + n] = acc[i][r];
                else Cp[(size_t)m * NC + n] = acc[i][r] + bv;
            }
        }
#endif
}

// ---- 16-bit operands (round 4, opt-in: DYF_TRAIN_OPERANDS=bf16 | fp16).  The same three gathers with the operands rounded to the
// bf16 (always: train_internal.h) WHILE they are staged into LDS (activations, gradients and weights stay fp32 in HBM: "fp32 master weights"),
// fp32 accumulation on v_mfma_f32_32x32x16: 16x the matrix rate of the fp32 instruction, so the kernel is bound by its operand
// loads (24 KB of fp32 per 32-deep K stage and workgroup) instead of by the matrix cores.  K stage 32 = two k16 sub-steps; LDS tiles
// [row][32 k] 16-bit, rows 80 bytes apart; a lane's fragment is one ds_read_b128.  What changes numerically is the products' input
// precision (8 / 11 mantissa bits): gradients agree with the fp32 path to ~1e-2 of their norm (tests/test_gpu_training.py), the
// usual mixed-precision trade; the default stays fp32 (1e-6 parity with autograd over the oracle).
constexpr int GK16 = 32, LDR16 = 40;  // K stage, LDS row pitch in 16-bit elements

template <int MODE>
__global__ __launch_bounds__(256) void t_gemm_mfma16(dyf::TConv g, const float* __restrict__ Ap, const float* __restrict__ Bp,
                                                     const float* __restrict__ bias, float* __restrict__ Cp, int split_len, int pmode) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) t16_t As[2][GM * LDR16];
    __shared__ __attribute__((aligned(16))) t16_t Bs[2][GN * LDR16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x, tn = blockIdx.y;
    const int taps = g.k * g.k;
    const long long opix = (long long)g.n * g.ho * g.wo, ipix = (long long)g.n * g.h * g.w;
    // pmode (data gradient of a 4 x 4 / stride 2 / pad 1 conv, h and w even): blockIdx.z is the parity class (iy & 1, ix & 1) of the input
    // pixels of this launch slice; a class is reached by 2 x 2 of the 16 taps only (ky = (iy + 1) mod 2 (+ 2), kx likewise), so its rows
    // run a dense K = 4 cout instead of 16 cout with three quarters of the gathers predicated off
    const int ppy = (MODE == TG_DGRAD && pmode) ? (int)(blockIdx.z >> 1) : 0, ppx = (MODE == TG_DGRAD && pmode) ? (int)(blockIdx.z & 1) : 0;
    const long long M = MODE == TG_FWD ? opix : MODE == TG_DGRAD ? (pmode ? ipix / 4 : ipix) : g.cout;
    const int CK = MODE == TG_FWD ? g.cin : g.cout;
    int tap = 0;
    long long kbeg = 0, kend = 0;
    int nstage, st0 = 0;
    if (MODE == TG_WGRAD) {
        tap = blockIdx.z % taps;
        kbeg = (long long)(blockIdx.z / taps) * split_len;
        kend = kbeg + split_len < opix ? kbeg + split_len : opix;
        nstage = (int)((kend - kbeg + GK16 - 1) / GK16);
    } else {
        const int total = (MODE == TG_DGRAD && pmode ? 4 : taps) * CK / GK16;
        st0 = split_len > 0 ? (int)blockIdx.z * split_len : 0;
        nstage = split_len > 0 ? min(split_len, total - st0) : total;
    }
    // A, forward / dgrad: 4 slots (row = s >> 3, k quad = s & 7); wgrad: 4 slots (k = s >> 5, m quad = s & 31)
    int a_b[4], a_y[4], a_x[4];
    bool a_ok[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = tid + 256 * i;
        a_b[i] = a_y[i] = a_x[i] = 0;
        a_ok[i] = false;
        if (MODE != TG_WGRAD) {
            const long long m = (long long)tm * GM + (s >> 3);
            a_ok[i] = m < M;
            const bool pm = MODE == TG_DGRAD && pmode;
            const int pw = MODE == TG_FWD ? g.wo : (pm ? g.w / 2 : g.w), ph = MODE == TG_FWD ? g.ho : (pm ? g.h / 2 : g.h);
            const long long mm = a_ok[i] ? m : 0;
            a_x[i] = (int)(mm % pw);
            a_y[i] = (int)((mm / pw) % ph);
            a_b[i] = (int)(mm / ((long long)pw * ph));
            if (pm) { a_x[i] = 2 * a_x[i] + ppx; a_y[i] = 2 * a_y[i] + ppy; }
        }
    }
    const int sh = (g.s & (g.s - 1)) == 0 ? __builtin_ctz((unsigned)g.s) : -1;  // (as in t_gemm_mfma)
    int w_ox[2] = {0, 0}, w_oy[2] = {0, 0}, w_b[2] = {0, 0};
    if (MODE == TG_WGRAD) {
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned pix0 = (unsigned)(kbeg + ((tid + 256 * i) >> 4));
            w_ox[i] = (int)(pix0 % (unsigned)g.wo);
            w_oy[i] = (int)((pix0 / (unsigned)g.wo) % (unsigned)g.ho);
            w_b[i] = (int)(pix0 / ((unsigned)g.wo * (unsigned)g.ho));
        }
    }
    float4 ra[4], rb[2];
    auto load = [&](int stage) {
        if (MODE != TG_WGRAD) {
            const int k0 = stage * GK16;
            const int tq = k0 / CK, c0 = k0 - tq * CK;
            const bool pm = MODE == TG_DGRAD && pmode;
            const int ky = pm ? ((ppy + 1) & 1) + 2 * (tq >> 1) : tq / g.k, kx = pm ? ((ppx + 1) & 1) + 2 * (tq & 1) : tq - (tq / g.k) * g.k;
            const int tp = ky * g.k + kx;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kq = (tid + 256 * i) & 7;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (MODE == TG_FWD) {
                    const int iy = a_y[i] * g.s - g.p + ky, ix = a_x[i] * g.s - g.p + kx;
                    if (a_ok[i] && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w)
                        ra[i] = *(const float4*)(Ap + (((size_t)a_b[i] * g.h + iy) * g.w + ix) * g.cin + c0 + kq * 4);
                } else {
                    const int ty = a_y[i] + g.p - ky, tx = a_x[i] + g.p - kx;
                    const int oy = sh >= 0 ? ty >> sh : ty / g.s, ox = sh >= 0 ? tx >> sh : tx / g.s;
                    if (a_ok[i] && ty >= 0 && tx >= 0 && oy * g.s == ty && ox * g.s == tx && oy < g.ho && ox < g.wo)
                        ra[i] = *(const float4*)(Ap + (((size_t)a_b[i] * g.ho + oy) * g.wo + ox) * g.cout + c0 + kq * 4);
                }
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = tid + 256 * i, bk = s >> 4, bnq = s & 15;
                if (MODE == TG_FWD)
                    rb[i] = *(const float4*)(Bp + ((size_t)tp * g.cin + c0 + bk) * g.cout + tn * GN + bnq * 4);
                else
                    rb[i] = *(const float4*)(Bp + ((size_t)(c0 + bk) * taps + tp) * g.cin + tn * GN + bnq * 4);
            }
        } else {
            const long long p0 = kbeg + (long long)stage * GK16;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int s = tid + 256 * i;
                const long long pix = p0 + (s >> 5);
                const int m = tm * GM + (s & 31) * 4;
                ra[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < kend && m < g.cout) ra[i] = *(const float4*)(Ap + (size_t)pix * g.cout + m);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int s = tid + 256 * i, bk = s >> 4, bnq = s & 15;
                const long long pix = p0 + bk;
                rb[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (pix < kend) {
                    const int ox = w_ox[i], oy = w_oy[i], b = w_b[i];
                    const int ky = tap / g.k, kx = tap - ky * g.k;
                    const int iy = oy * g.s - g.p + ky, ix = ox * g.s - g.p + kx;
                    if ((unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w)
                        rb[i] = *(const float4*)(Bp + (((size_t)b * g.h + iy) * g.w + ix) * g.cin + tn * GN + bnq * 4);
                }
                w_ox[i] += GK16;  // the next stage's pixel
                while (w_ox[i] >= g.wo) {
                    w_ox[i] -= g.wo;
                    if (++w_oy[i] >= g.ho) { w_oy[i] = 0; ++w_b[i]; }
                }
            }
        }
    };
    auto store = [&](int buf) {
        if (MODE != TG_WGRAD) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // 4 consecutive k of one row: one 8-byte store
                const int s = tid + 256 * i;
                *(uint2*)&As[buf][(s >> 3) * LDR16 + 4 * (s & 7)] = make_uint2(pack_t16x2(ra[i].x, ra[i].y), pack_t16x2(ra[i].z, ra[i].w));
            }
        } else {
#pragma unroll
            for (int i = 0; i < 4; ++i) {  // 4 consecutive rows (cout) of one k (pixel)
                const int s = tid + 256 * i, k = s >> 5;
                t16_t* d = &As[buf][((s & 31) * 4) * LDR16 + k];
                d[0] = f32_to_t16(ra[i].x); d[LDR16] = f32_to_t16(ra[i].y); d[2 * LDR16] = f32_to_t16(ra[i].z); d[3 * LDR16] = f32_to_t16(ra[i].w);
            }
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {  // 4 consecutive columns n of one k
            const int s = tid + 256 * i, bk = s >> 4, bnq = s & 15;
            t16_t* d = &Bs[buf][(bnq * 4) * LDR16 + bk];
            d[0] = f32_to_t16(rb[i].x); d[LDR16] = f32_to_t16(rb[i].y); d[2 * LDR16] = f32_to_t16(rb[i].z); d[3 * LDR16] = f32_to_t16(rb[i].w);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    if (nstage > 0) {
        load(st0);
        store(0);
    }
    __syncthreads();
    for (int st = 0; st < nstage; ++st) {
        const int buf = st & 1;
        if (st + 1 < nstage) load(st0 + st + 1);
        const t16_t* ar0 = &As[buf][(wm * 64 + l31) * LDR16 + hi * 8];
        const t16_t* ar1 = ar0 + 32 * LDR16;
        const t16_t* br = &Bs[buf][(wn * 32 + l31) * LDR16 + hi * 8];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const t16x8_t a0 = *(const t16x8_t*)(ar0 + ks * 16), a1 = *(const t16x8_t*)(ar1 + ks * 16);
            const t16x8_t b = *(const t16x8_t*)(br + ks * 16);
            acc[0] = T16_MFMA_32x32x16(a0, b, acc[0], 0, 0, 0);
            acc[1] = T16_MFMA_32x32x16(a1, b, acc[1], 0, 0, 0);
        }
        if (st + 1 < nstage) store(buf ^ 1);
        __syncthreads();
    }
    const int n = tn * GN + wn * 32 + l31;
    const float bv = (MODE != TG_WGRAD && bias) ? bias[n] : 0.0f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const long long m = (long long)tm * GM + wm * 64 + i * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            if (m >= M) continue;
            if (MODE == TG_WGRAD) {
                atomicAdd(Cp + ((size_t)m * taps + tap) * g.cin + n, acc[i][r]);
            } else {
                const int NC = MODE == TG_FWD ? g.cout : g.cin;
                if (MODE == TG_DGRAD && pmode) {
                    const int pw = g.w / 2, ph = g.h / 2;
                    const int cx = (int)(m % pw), cy = (int)((m / pw) % ph), cb = (int)(m / ((long long)pw * ph));
                    Cp[(((size_t)cb * g.h + 2 * cy + ppy) * g.w + 2 * cx + ppx) * NC + n] = acc[i][r] + bv;
                } else if (split_len > 0) Cp[((size_t)blockIdx.z * M + m) * NC + n] = acc[i][r];
                else Cp[(size_t)m * NC + n] = acc[i][r] + bv;
            }
        }
#endif
}

// y[i] = bias[i % N] + sum over the splits, in split order
__global__ void t_splitk_finish(const float* ws, int splits, long long MN, int N, const float* bias, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= MN) return;
    float s = bias ? bias[i % N] : 0.0f;
    for (int z = 0; z < splits; ++z) s += ws[(size_t)z * MN + i];
    y[i] = s;
}

}  // namespace

namespace dyf {

// 16-bit operands: the training convs round their operands to bf16 (both builds: train_internal.h) while staging them (t_gemm_mfma16,
// train_halo16.hip) -- chosen per engine by dyf_train_set_precision(16) (the reference's `trainer.precision=16`), or, for engines
// that did not say (0), by the tests' kernel-form switch DYF_TRAIN_OPERANDS=bf16 | 16 (dyf_debug_set_form; read per call).  Otherwise fp32.
thread_local int g_train_precision = 0;
bool train_operands16() {
    if (g_train_precision == 16) return true;
    if (g_train_precision == 32) return false;
    const char* v = dyf_form("DYF_TRAIN_OPERANDS");
    return v && (!strcmp(v, "bf16") || !strcmp(v, "16"));
}

// split-K plan of a forward / dgrad launch: (splits, stages per split); splits == 1 -> no split
static void plan_splitk(long long tiles, int stages, int& splits, int& len) {
    splits = 1;
    len = 0;
    const bool enabled = !(dyf_form("DYF_TRAIN_SPLITK") && atoi(dyf_form("DYF_TRAIN_SPLITK")) == 0);
    if (!enabled || tiles >= 128 || stages < 16) return;
    long long want = std::min<long long>((512 + tiles - 1) / tiles, stages / 4);  // >= 4 stages per split
    if (want < 2) return;
    len = (int)((stages + want - 1) / want);
    splits = (stages + len - 1) / len;
    if (splits < 2) { splits = 1; len = 0; }
}

bool tgemm_conv_fwd(const TConv& g, const float* x, const float* wt, const float* bias, float* y, float* ws, size_t ws_floats,
                    hipStream_t st) {
    if (g.cin % GK != 0 || g.cout % GN != 0) return false;
    const long long M = (long long)g.n * g.ho * g.wo, mt = (M + GM - 1) / GM;
    const bool h16 = train_operands16() && g.cin % GK16 == 0;
    if (h16 && thalo_conv3x3(g, 0, x, wt, bias, y, ws, ws_floats, st)) return true;
    int splits, len;
    plan_splitk(mt * (g.cout / GN), g.k * g.k * g.cin / (h16 ? GK16 : GK), splits, len);
    if (splits > 1 && (ws == nullptr || (size_t)splits * M * g.cout > ws_floats)) { splits = 1; len = 0; }
    if (h16)
        hipLaunchKernelGGL(t_gemm_mfma16<TG_FWD>, dim3((unsigned)mt, g.cout / GN, splits), dim3(256), 0, st, g, x, wt, bias, splits > 1 ? ws : y, len, 0);
    else
        hipLaunchKernelGGL(t_gemm_mfma<TG_FWD>, dim3((unsigned)mt, g.cout / GN, splits), dim3(256), 0, st, g, x, wt, bias, splits > 1 ? ws : y, len);
    if (splits > 1)
        hipLaunchKernelGGL(t_splitk_finish, dim3((unsigned)((M * g.cout + 255) / 256)), dim3(256), 0, st, ws, splits, M * g.cout, g.cout, bias, y);
    return true;
}

bool tgemm_conv_dgrad(const TConv& g, const float* dz, const float* w, const float* bias, float* dx, float* ws, size_t ws_floats,
                      hipStream_t st) {
    if (g.cout % GK != 0 || g.cin % GN != 0) return false;
    const long long M = (long long)g.n * g.h * g.w, mt = (M + GM - 1) / GM;
    const bool h16 = train_operands16() && g.cout % GK16 == 0;
    if (h16 && thalo_conv3x3(g, 1, dz, w, bias, dx, ws, ws_floats, st)) return true;
    {   // 4 x 4 / stride 2 / pad 1: one launch slice per parity class of the input pixels (see the kernel); small planes keep split-K
        const bool pclass = !(dyf_form("DYF_TRAIN_DGRAD_PARITY") && atoi(dyf_form("DYF_TRAIN_DGRAD_PARITY")) == 0);  // per call: tests flip it
        const long long mq = ((long long)g.n * g.h * g.w / 4 + GM - 1) / GM;
        if (h16 && pclass && g.k == 4 && g.s == 2 && g.p == 1 && g.h % 2 == 0 && g.w % 2 == 0 && g.ho == g.h / 2 && g.wo == g.w / 2 &&
            mq * (g.cin / GN) >= 64 && mq <= 0x7fffffffll) {
            dyf_form_note("t_gemm_mfma16:dgrad_parity", g.n);
            hipLaunchKernelGGL(t_gemm_mfma16<TG_DGRAD>, dim3((unsigned)mq, g.cin / GN, 4), dim3(256), 0, st, g, dz, w, bias, dx, 0, 1);
            return true;
        }
    }
    int splits, len;
    plan_splitk(mt * (g.cin / GN), g.k * g.k * g.cout / (h16 ? GK16 : GK), splits, len);
    if (splits > 1 && (ws == nullptr || (size_t)splits * M * g.cin > ws_floats)) { splits = 1; len = 0; }
    if (h16)
        hipLaunchKernelGGL(t_gemm_mfma16<TG_DGRAD>, dim3((unsigned)mt, g.cin / GN, splits), dim3(256), 0, st, g, dz, w, bias, splits > 1 ? ws : dx, len, 0);
    else
        hipLaunchKernelGGL(t_gemm_mfma<TG_DGRAD>, dim3((unsigned)mt, g.cin / GN, splits), dim3(256), 0, st, g, dz, w, bias, splits > 1 ? ws : dx, len);
    if (splits > 1)
        hipLaunchKernelGGL(t_splitk_finish, dim3((unsigned)((M * g.cin + 255) / 256)), dim3(256), 0, st, ws, splits, M * g.cin, g.cin, bias, dx);
    return true;
}

// dw += ... (the caller accumulates the bias gradient separately)
bool tgemm_conv_wgrad(const TConv& g, const float* dz, const float* x, float* dw, hipStream_t st) {
    if (g.cin % GN != 0 || g.cout % 4 != 0) return false;
    const long long pix = (long long)g.n * g.ho * g.wo;
    const int taps = g.k * g.k, mt = (g.cout + GM - 1) / GM, nt = g.cin / GN;
    // enough workgroups to fill the chip, at least 256 pixels per split
    long long splits = std::max<long long>(1, std::min<long long>((pix + 255) / 256, (2048 + (long long)mt * nt * taps - 1) / ((long long)mt * nt * taps)));
    const bool h16 = train_operands16();
    if (h16 && thalo_wgrad3x3(g, dz, x, dw, st)) return true;
    const int gk = h16 ? GK16 : GK;
    int len = (int)(((pix + splits - 1) / splits + gk - 1) / gk * gk);
    splits = (pix + len - 1) / len;
    if ((long long)taps * splits > 65535) return false;
    if (h16)
        hipLaunchKernelGGL(t_gemm_mfma16<TG_WGRAD>, dim3(mt, nt, (unsigned)(taps * splits)), dim3(256), 0, st, g, dz, x, nullptr, dw, len, 0);
    else
        hipLaunchKernelGGL(t_gemm_mfma<TG_WGRAD>, dim3(mt, nt, (unsigned)(taps * splits)), dim3(256), 0, st, g, dz, x, nullptr, dw, len);
    return true;
}

}  // namespace dyf
