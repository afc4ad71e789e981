// Training step of the forecaster objective on the GPU: forward WITH batch-statistics BatchNorm / dropout and the backward
// pass of arch unet_simple.
//
// Replaces, for `DYffusion.p_losses` in training mode (src/diffusion/dyffusion.py:496-567, entered from
// BaseDiffusion.forward, src/diffusion/_base_diffusion.py:81-106), what the reference gets from torch.autograd over
// src/models/unet_simple.py:13-82,164-197: conv / transposed-conv dgrad + wgrad, BatchNorm2d in training mode (batch mean /
// biased variance, running-statistics update), GroupNorm, FiLM `x*(scale+1)+shift` with its time-MLP, (Leaky)ReLU, Dropout,
// bilinear resampling (align_corners=False) and its adjoint, skip concatenation.  The frozen interpolator takes part with
// running-statistics BatchNorm and input gradients only (the second loss term differentiates THROUGH it, :534-557).
//
// Everything here is fp32 (NHWC activations, fp32 weights in [cout][tap][cin] and [tap][cin][cout] order, fp64 statistics):
// gradient parity with autograd is ~1e-6, not bf16-limited.  The convolutions with >= 64 channels run on the fp32 matrix cores
// (train_gemm.hip: one implicit-GEMM kernel with forward / dgrad / wgrad gathers); the plain VALU kernels below serve the 3- /
// 5- / 8-channel ends of the network and DYF_TRAIN_MFMA=0.  Tapes and temporaries come from a caching allocator (TrainState).
#include "engine_internal.h"
#include "train_internal.h"

#include "../../include/dyffusion_hip.h"

using namespace dyf;

namespace dyf {

struct TBlockW {            // fp32 parameters (and gradients) of one UNetBlock
    float *w = nullptr, *wt = nullptr;   // conv weight [cout][tap][cin] and its [tap][cin][cout] transpose (forward)
    float *b = nullptr, *gamma = nullptr, *beta = nullptr, *rmean = nullptr, *rvar = nullptr;
    float *fw = nullptr, *fb = nullptr;  // FiLM head Linear(tdim -> 2 cout)
    float *g_w = nullptr, *g_b = nullptr, *g_gamma = nullptr, *g_beta = nullptr, *g_fw = nullptr, *g_fb = nullptr;
};

struct TNet {
    TBlockW blk[12];
    float *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr, *g_t_w1 = nullptr, *g_t_b1 = nullptr,
          *g_t_w2 = nullptr, *g_t_b2 = nullptr;
    float *stem_w = nullptr, *stem_wt = nullptr, *stem_b = nullptr, *g_stem_w = nullptr, *g_stem_b = nullptr;  // [dim][cin]
    float *ro_w = nullptr, *ro_wt = nullptr, *ro_b = nullptr, *g_ro_w = nullptr, *g_ro_b = nullptr;  // conv C: [dim][16][C]
    std::vector<std::pair<float*, size_t>> grads;  // every gradient buffer (zeroing)
    std::vector<void*> owned;
    bool ready = false;
};

struct TTape {              // what one recorded forward leaves for its backward
    int net = -1, nb = 0, flags = 0;
    std::vector<void*> owned;
    float *x_in = nullptr, *x_up = nullptr, *s0 = nullptr;
    float *cin_ptr[12] = {}, *z[12] = {}, *y[12] = {}, *ss[12] = {}, *mean[12] = {}, *rstd[12] = {};
    float *xlast = nullptr;      // input of the readout
    float *e0 = nullptr, *l1 = nullptr, *gl = nullptr, *temb = nullptr, *silu = nullptr;  // time-MLP chain
    uint32_t* row_keys = nullptr;
};

struct RTNet;   // ResNet-UNet training copy / tapes (train_resnet.inc)
struct RTape;

struct TrainState {
    TNet net[2];
    TTape tape[4];
    RTNet* rnet[2] = {nullptr, nullptr};
    RTape* rtape[4] = {nullptr, nullptr, nullptr, nullptr};
    // caching allocator of the tapes / temporaries: blocks go back to the pool instead of hipFree (which synchronises the
    // device) and are handed out again by exact size -- after the first step a training step allocates nothing.  Everything
    // runs on one stream, so reuse is ordered behind the previous use.
    std::multimap<size_t, void*> pool;
    size_t pool_bytes = 0;  // bytes idle in the pool
    float* splitk_ws = nullptr;  // partial sums of the split-K convolutions (train_gemm.hip), TRAIN_SPLITK_FLOATS floats
    std::vector<void*> ws_owned;
    hipStream_t stream = nullptr;  // stream of the running train_forward / train_backward (zero-fills are queued on it)
};

}  // namespace dyf

namespace {

#define TK(expr)                                                        \
    do {                                                                \
        hipError_t _e = (expr);                                         \
        if (_e != hipSuccess) return fail(e, DYF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

inline unsigned nblk(long long total, int bs = 256) { return (unsigned)((total + bs - 1) / bs); }

// ------------------------------------------------------------------------------------------------ layout / resampling
__global__ void t_nchw_cat_to_nhwc(const float* s0, int c0, const float* s1, int c1, const float* s2, int c2, int n, int hw, float* out) {
    const int C = c0 + c1 + c2;
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * hw * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int b = (int)(p / hw), px = (int)(p % hw);
    float v;
    if (c < c0) v = s0[((size_t)b * c0 + c) * hw + px];
    else if (c < c0 + c1) v = s1[((size_t)b * c1 + (c - c0)) * hw + px];
    else v = s2[((size_t)b * c2 + (c - c0 - c1)) * hw + px];
    out[i] = v;
}

// NHWC channel range [c_lo, c_lo + cn) -> NCHW (cn channels)
__global__ void t_nhwc_to_nchw(const float* in, int n, int hw, int C, int c_lo, int cn, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * cn * hw) return;
    const int px = (int)(i % hw), c = (int)((i / hw) % cn), b = (int)(i / ((long long)hw * cn));
    out[i] = in[((size_t)b * hw + px) * C + c_lo + c];
}

__global__ void t_nchw_to_nhwc(const float* in, int n, int hw, int C, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * C * hw) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int b = (int)(p / hw), px = (int)(p % hw);
    out[i] = in[((size_t)b * C + c) * hw + px];
}

// F.interpolate(mode="bilinear", align_corners=False), NHWC fp32
__global__ void t_resize_fwd(const float* in, int n, int ih, int iw, int C, int oh, int ow, int nearest, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * oh * ow * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int ox = (int)(p % ow), oy = (int)((p / ow) % oh), b = (int)(p / ((long long)ow * oh));
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_coord(oy, (float)ih / (float)oh, ih, y0, y1, ly, nearest != 0);
    bilinear_coord(ox, (float)iw / (float)ow, iw, x0, x1, lx, nearest != 0);
    const float* base = in + (size_t)b * ih * iw * C + c;
    const float v00 = base[((size_t)y0 * iw + x0) * C], v01 = base[((size_t)y0 * iw + x1) * C];
    const float v10 = base[((size_t)y1 * iw + x0) * C], v11 = base[((size_t)y1 * iw + x1) * C];
    const float top = v00 * (1.0f - lx) + v01 * lx, bot = v10 * (1.0f - lx) + v11 * lx;
    out[i] = top * (1.0f - ly) + bot * ly;
}

// adjoint of t_resize_fwd: din (zero-initialised) += scatter of dout
// The x2 bilinear upsample (align_corners = False) of the decoder blocks and its adjoint as GATHERS over 4 channels per thread
// (C % 4 == 0).  Output row 2i + p reads input rows (i - 1 + p, i + p) with weights (0.25, 0.75) / (0.75, 0.25), clamped at the
// borders; so input row i collects 0.25 dout[2i-1] + 0.75 dout[2i] + 0.75 dout[2i+1] + 0.25 dout[2i+2], rows outside dropped, the
// clamped taps of the first / last output row added (their whole weight lands on row 0 / h-1).  No atomics, every store 16 B.
__device__ __forceinline__ void up2x_adj_taps(int i, int h, int o[4], float w[4]) {
    o[0] = 2 * i - 1; o[1] = 2 * i; o[2] = 2 * i + 1; o[3] = 2 * i + 2;
    w[0] = i >= 1 ? 0.25f : 0.0f;
    w[1] = i >= 1 ? 0.75f : 1.0f;
    w[2] = i <= h - 2 ? 0.75f : 1.0f;
    w[3] = i <= h - 2 ? 0.25f : 0.0f;
    o[0] = max(o[0], 0);
    o[3] = min(o[3], 2 * h - 1);
}
// (din / Ca4, din2 / Cb4: the gradient of a CONCATENATED low-resolution tensor leaves as its two parts -- channel quads [0, Ca4) to
// din, the rest to din2 -- so that no split pass follows; din2 = null, Ca4 = C4: one tensor)
__global__ __launch_bounds__(256) void t_up2x_bwd(const float* dout, int n, int h, int w, int C4, float* din, int Ca4, float* din2) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * h * w * C4) return;
    const int c4 = (int)(idx % C4);
    const long long p = idx / C4;
    const int j = (int)(p % w), i = (int)((p / w) % h), b = (int)(p / ((long long)w * h));
    int oy[4], ox[4];
    float wy[4], wx[4];
    up2x_adj_taps(i, h, oy, wy);
    up2x_adj_taps(j, w, ox, wx);
    const float4* src = (const float4*)dout + (size_t)b * (4 * (size_t)h * w) * C4 + c4;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float f = wy[a] * wx[q];
            const float4 v = src[((size_t)oy[a] * (2 * w) + ox[q]) * C4];
            acc.x = fmaf(f, v.x, acc.x); acc.y = fmaf(f, v.y, acc.y); acc.z = fmaf(f, v.z, acc.z); acc.w = fmaf(f, v.w, acc.w);
        }
    if (c4 < Ca4) ((float4*)din)[p * Ca4 + c4] = acc;
    else ((float4*)din2)[p * (C4 - Ca4) + (c4 - Ca4)] = acc;
}
// (in / Ca4, in2: x2 upsample of the concatenation [in | in2] without materialising it; in2 = null, Ca4 = C4: one tensor)
__global__ __launch_bounds__(256) void t_up2x_fwd(const float* in, int n, int h, int w, int C4, float* out, int Ca4, const float* in2) {
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * 4 * h * w * C4) return;
    const int c4 = (int)(idx % C4);
    const long long p = idx / C4;
    const int ox = (int)(p % (2 * w)), oy = (int)((p / (2 * w)) % (2 * h)), b = (int)(p / (4LL * w * h));
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_coord(oy, 0.5f, h, y0, y1, ly);
    bilinear_coord(ox, 0.5f, w, x0, x1, lx);
    const int Cs = c4 < Ca4 ? Ca4 : C4 - Ca4;  // channel quads of the source this quad comes from
    const float4* base = (c4 < Ca4 ? (const float4*)in + c4 : (const float4*)in2 + (c4 - Ca4)) + (size_t)b * h * w * Cs;
    const float4 v00 = base[((size_t)y0 * w + x0) * Cs], v01 = base[((size_t)y0 * w + x1) * Cs];
    const float4 v10 = base[((size_t)y1 * w + x0) * Cs], v11 = base[((size_t)y1 * w + x1) * Cs];
    float4 r;
#define UP2X_MIX(m) { const float top = v00.m * (1.0f - lx) + v01.m * lx, bot = v10.m * (1.0f - lx) + v11.m * lx; r.m = top * (1.0f - ly) + bot * ly; }
    UP2X_MIX(x) UP2X_MIX(y) UP2X_MIX(z) UP2X_MIX(w)
#undef UP2X_MIX
    typedef float nt_f32x4 __attribute__((ext_vector_type(4)));
    __builtin_nontemporal_store((nt_f32x4){r.x, r.y, r.z, r.w}, (nt_f32x4*)out + idx);  // a 2 GB stream nobody re-reads from L2
}
__global__ void t_resize_bwd(const float* dout, int n, int ih, int iw, int C, int oh, int ow, int nearest, float* din) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)n * oh * ow * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int ox = (int)(p % ow), oy = (int)((p / ow) % oh), b = (int)(p / ((long long)ow * oh));
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_coord(oy, (float)ih / (float)oh, ih, y0, y1, ly, nearest != 0);
    bilinear_coord(ox, (float)iw / (float)ow, iw, x0, x1, lx, nearest != 0);
    const float g = dout[i];
    float* base = din + (size_t)b * ih * iw * C + c;
    atomicAdd(base + ((size_t)y0 * iw + x0) * C, g * (1.0f - ly) * (1.0f - lx));
    atomicAdd(base + ((size_t)y0 * iw + x1) * C, g * (1.0f - ly) * lx);
    atomicAdd(base + ((size_t)y1 * iw + x0) * C, g * ly * (1.0f - lx));
    atomicAdd(base + ((size_t)y1 * iw + x1) * C, g * ly * lx);
}

// ------------------------------------------------------------------------------------------------ convolution (fp32, NHWC)

// y[n,oy,ox,co] = b[co] + sum_{tap,ci} x[n, oy*s-p+ky, ox*s-p+kx, ci] * wt[tap][ci][co]
__global__ void t_conv_fwd(TConv g, const float* x, const float* wt, const float* bias, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)g.n * g.ho * g.wo * g.cout) return;
    const int co = (int)(i % g.cout);
    const long long pix = i / g.cout;
    const int ox = (int)(pix % g.wo), oy = (int)((pix / g.wo) % g.ho), b = (int)(pix / ((long long)g.wo * g.ho));
    float acc = bias ? bias[co] : 0.0f;
    for (int ky = 0; ky < g.k; ++ky) {
        const int iy = oy * g.s - g.p + ky;
        if ((unsigned)iy >= (unsigned)g.h) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int ix = ox * g.s - g.p + kx;
            if ((unsigned)ix >= (unsigned)g.w) continue;
            const float* xp = x + (((size_t)b * g.h + iy) * g.w + ix) * g.cin;
            const float* wp = wt + (size_t)(ky * g.k + kx) * g.cin * g.cout + co;
            for (int ci = 0; ci < g.cin; ++ci) acc = fmaf(xp[ci], wp[(size_t)ci * g.cout], acc);
        }
    }
    y[i] = acc;
}

// the same sums in the same order, four output channels per thread (cout % 4 == 0): 16-byte weight loads and stores.  The
// small-channel convs of unet_simple (the 1x1 stem 5 -> 64 and the readout's input gradient 3 -> 64, 4x4 / stride 2) are
// latency-bound in the one-channel form: 1.4 ms per launch at 32 rows x 256^2.
__global__ __launch_bounds__(256) void t_conv_fwd4(TConv g, const float* x, const float* wt, const float* bias, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c4 = g.cout >> 2;
    if (i >= (long long)g.n * g.ho * g.wo * c4) return;
    const int co = (int)(i % c4) * 4;
    const long long pix = i / c4;
    const int ox = (int)(pix % g.wo), oy = (int)((pix / g.wo) % g.ho), b = (int)(pix / ((long long)g.wo * g.ho));
    float4 acc = bias ? *(const float4*)(bias + co) : make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    for (int ky = 0; ky < g.k; ++ky) {
        const int iy = oy * g.s - g.p + ky;
        if ((unsigned)iy >= (unsigned)g.h) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int ix = ox * g.s - g.p + kx;
            if ((unsigned)ix >= (unsigned)g.w) continue;
            const float* xp = x + (((size_t)b * g.h + iy) * g.w + ix) * g.cin;
            const float* wp = wt + (size_t)(ky * g.k + kx) * g.cin * g.cout + co;
            for (int ci = 0; ci < g.cin; ++ci) {
                const float xv = xp[ci];
                const float4 w4 = *(const float4*)(wp + (size_t)ci * g.cout);
                acc.x = fmaf(xv, w4.x, acc.x);
                acc.y = fmaf(xv, w4.y, acc.y);
                acc.z = fmaf(xv, w4.z, acc.z);
                acc.w = fmaf(xv, w4.w, acc.w);
            }
        }
    }
    *(float4*)(y + (size_t)pix * g.cout + co) = acc;
}

// The same convs on the fp32 matrix cores (wo a multiple of 32: a wave's 32 pixels are one row segment): y[p][co] = bias[co] +
// sum_j patch[p][j] W[j][co] with j = (tap, ci) walked two per v_mfma_f32_32x32x2_f32 -- lane (pixel, j parity) gathers its patch
// value through a small LDS table j -> (ky, kx, ci) (taps outside the image predicated to zero), the W row of a step is one
// coalesced 256-byte read; 64 output channels per wave pass, results leave as 128-byte rows.  Summation order differs from the
// kernels above in the last bits (fp32 throughout).  550 us per launch in the four-channel form at 32 rows x 256^2.
typedef __attribute__((ext_vector_type(16))) float tf_f32x16;
__global__ __launch_bounds__(256) void t_conv_fwd_smallc_mfma(TConv g, const float* __restrict__ x, const float* __restrict__ wt,
                                                              const float* __restrict__ bias, float* __restrict__ y, int tiles_per_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ int tbl[128];
    const int nacc = g.k * g.k * g.cin, ksteps = (nacc + 1) >> 1;
    if (threadIdx.x < 128) {
        const int j = threadIdx.x, tap = j / g.cin, ky = tap / g.k;
        tbl[j] = j < nacc ? (ky | ((tap - ky * g.k) << 8) | ((j - tap * g.cin) << 16)) : -1;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, hi = lane >> 5;
    const int cob = blockIdx.y * 64;
    const long long tiles = (long long)g.n * g.ho * g.wo / 32;
    const long long t0 = ((long long)blockIdx.x * 4 + wave) * tiles_per_wave, t1 = t0 + tiles_per_wave < tiles ? t0 + tiles_per_wave : tiles;
    const float bv0 = bias ? bias[cob + l31] : 0.0f, bv1 = bias ? bias[cob + 32 + l31] : 0.0f;
    for (long long tile = t0; tile < t1; ++tile) {
        const long long p0 = tile * 32;
        const int ox0 = (int)(p0 % g.wo), oy = (int)((p0 / g.wo) % g.ho), b = (int)(p0 / ((long long)g.wo * g.ho));
        const int iy0 = oy * g.s - g.p, ix0 = (ox0 + l31) * g.s - g.p;
        tf_f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.0f;
#pragma unroll 4
        for (int kk = 0; kk < ksteps; ++kk) {
            const int j = 2 * kk + hi, e = tbl[j];
            float a = 0.0f, w0 = 0.0f, w1 = 0.0f;
            if (e >= 0) {
                const int iy = iy0 + (e & 255), ix = ix0 + ((e >> 8) & 255);
                if ((unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w) a = x[(((size_t)b * g.h + iy) * g.w + ix) * g.cin + (e >> 16)];
                w0 = wt[(size_t)j * g.cout + cob + l31];
                w1 = wt[(size_t)j * g.cout + cob + 32 + l31];
            }
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, w1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {  // D[i][j]: lane j = l31 -> channel, register r -> pixel i = 8 (r >> 2) + 4 hi + (r & 3)
            const size_t row = (size_t)(p0 + 8 * (r >> 2) + 4 * hi + (r & 3)) * g.cout + cob;
            y[row + l31] = acc0[r] + bv0;
            y[row + 32 + l31] = acc1[r] + bv1;
        }
    }
#endif
}

// dx[n,iy,ix,ci] = sum over (ky,kx) with (iy+p-ky) % s == 0 and co of dz[n,(iy+p-ky)/s,(ix+p-kx)/s,co] * w[co][tap][ci]
// (+ bias[ci] when used as the FORWARD of a transposed convolution)
__global__ void t_conv_dgrad(TConv g, const float* dz, const float* w, const float* bias, float* dx) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)g.n * g.h * g.w * g.cin) return;
    const int ci = (int)(i % g.cin);
    const long long pix = i / g.cin;
    const int ix = (int)(pix % g.w), iy = (int)((pix / g.w) % g.h), b = (int)(pix / ((long long)g.w * g.h));
    float acc = bias ? bias[ci] : 0.0f;
    const int taps = g.k * g.k;
    for (int ky = 0; ky < g.k; ++ky) {
        const int ty = iy + g.p - ky;
        if (ty < 0 || ty % g.s) continue;
        const int oy = ty / g.s;
        if (oy >= g.ho) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int tx = ix + g.p - kx;
            if (tx < 0 || tx % g.s) continue;
            const int ox = tx / g.s;
            if (ox >= g.wo) continue;
            const float* zp = dz + (((size_t)b * g.ho + oy) * g.wo + ox) * g.cout;
            const float* wp = w + (size_t)(ky * g.k + kx) * g.cin + ci;
            for (int co = 0; co < g.cout; ++co) acc = fmaf(zp[co], wp[(size_t)co * taps * g.cin], acc);
        }
    }
    dx[i] = acc;
}

// t_conv_dgrad for a handful of result channels (cin <= 4: the readout's transposed conv 64 -> C, run as the dgrad form): 16
// lanes per pixel, lane q owns channels co = 4q + {0..3} (+ 64, ...) of dz -- a pixel's 256-byte row is one coalesced read per
// tap --, the weights sit in LDS as [tap][co][4], the cin partial sums are reduced over the 16 lanes with shuffles.
__global__ __launch_bounds__(256) void t_conv_dgrad_smalln(TConv g, const float* dz, const float* w, const float* bias, float* dx) {
    extern __shared__ float4 sn_w[];  // [taps][cout]
    const int taps = g.k * g.k;
    for (int i = threadIdx.x; i < taps * g.cout; i += 256) {
        const int tap = i / g.cout, co = i - tap * g.cout;
        const float* wp = w + ((size_t)co * taps + tap) * g.cin;
        sn_w[i] = make_float4(wp[0], g.cin > 1 ? wp[1] : 0.0f, g.cin > 2 ? wp[2] : 0.0f, g.cin > 3 ? wp[3] : 0.0f);
    }
    __syncthreads();
    const long long pix = ((long long)blockIdx.x * 256 + threadIdx.x) >> 4;
    const int q = threadIdx.x & 15;
    const long long total = (long long)g.n * g.h * g.w;
    const bool live = pix < total;
    const long long pp = live ? pix : total - 1;
    const int ix = (int)(pp % g.w), iy = (int)((pp / g.w) % g.h), b = (int)(pp / ((long long)g.w * g.h));
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    // stride a power of two (every shipped layer): shifts -- two divisions by a run-time stride per tap were most of this kernel's
    // instructions (16 taps x ~70 of them against four taps of useful work: 3.5 ms per readout launch at 32 rows x 512^2)
    const int sh = (g.s & (g.s - 1)) == 0 ? __builtin_ctz((unsigned)g.s) : -1;
    for (int ky = 0; ky < g.k; ++ky) {
        const int ty = iy + g.p - ky, oy = sh >= 0 ? ty >> sh : ty / g.s;
        if (ty < 0 || oy * g.s != ty || oy >= g.ho) continue;
        for (int kx = 0; kx < g.k; ++kx) {
            const int tx = ix + g.p - kx, ox = sh >= 0 ? tx >> sh : tx / g.s;
            if (tx < 0 || ox * g.s != tx || ox >= g.wo) continue;
            const float* zp = dz + (((size_t)b * g.ho + oy) * g.wo + ox) * g.cout;
            const float4* wt = sn_w + (size_t)(ky * g.k + kx) * g.cout;
            for (int c0 = 4 * q; c0 < g.cout; c0 += 64) {
                const float4 z = *(const float4*)(zp + c0);
                const float4 w0 = wt[c0], w1 = wt[c0 + 1], w2 = wt[c0 + 2], w3 = wt[c0 + 3];
                a0 = fmaf(z.x, w0.x, fmaf(z.y, w1.x, fmaf(z.z, w2.x, fmaf(z.w, w3.x, a0))));
                a1 = fmaf(z.x, w0.y, fmaf(z.y, w1.y, fmaf(z.z, w2.y, fmaf(z.w, w3.y, a1))));
                a2 = fmaf(z.x, w0.z, fmaf(z.y, w1.z, fmaf(z.z, w2.z, fmaf(z.w, w3.z, a2))));
                a3 = fmaf(z.x, w0.w, fmaf(z.y, w1.w, fmaf(z.z, w2.w, fmaf(z.w, w3.w, a3))));
            }
        }
    }
    for (int off = 8; off > 0; off >>= 1) {
        a0 += __shfl_xor(a0, off); a1 += __shfl_xor(a1, off); a2 += __shfl_xor(a2, off); a3 += __shfl_xor(a3, off);
    }
    if (live && q == 0) {
        float* o = dx + (size_t)pix * g.cin;
        o[0] = a0 + (bias ? bias[0] : 0.0f);
        if (g.cin > 1) o[1] = a1 + (bias ? bias[1] : 0.0f);
        if (g.cin > 2) o[2] = a2 + (bias ? bias[2] : 0.0f);
        if (g.cin > 3) o[3] = a3 + (bias ? bias[3] : 0.0f);
    }
}

// dw[co][tap][ci] += sum_{n,oy,ox} dz[n,oy,ox,co] * x[n,oy*s-p+ky,ox*s-p+kx,ci].  One workgroup = a 16 x 16 (co, ci) tile of
// one tap over a slice of the output pixels; slices are merged with atomics.  db[co] += sum dz (tap 0 / ci-tile 0 only).
__global__ __launch_bounds__(256) void t_conv_wgrad(TConv g, const float* dz, const float* x, float* dw, float* db, int pix_per_block) {
    const int taps = g.k * g.k;
    const int co_tiles = (g.cout + 15) / 16, ci_tiles = (g.cin + 15) / 16;
    int bid = blockIdx.x;
    const int cit = bid % ci_tiles; bid /= ci_tiles;
    const int cot = bid % co_tiles; bid /= co_tiles;
    const int tap = bid % taps;
    const int slice = bid / taps;
    const int ky = tap / g.k, kx = tap % g.k;
    const int tco = threadIdx.x >> 4, tci = threadIdx.x & 15;
    const int co = cot * 16 + tco, ci = cit * 16 + tci;
    const long long M = (long long)g.n * g.ho * g.wo;
    const long long m0 = (long long)slice * pix_per_block, m1 = m0 + pix_per_block < M ? m0 + pix_per_block : M;
    __shared__ float sz[16][17], sx[16][17];
    float acc = 0.0f, accb = 0.0f;
    for (long long mb = m0; mb < m1; mb += 16) {
        {   // stage 16 pixels x 16 channels of dz and of the shifted x
            const int pp = threadIdx.x >> 4, cc = threadIdx.x & 15;
            const long long m = mb + pp;
            float vz = 0.0f, vx = 0.0f;
            if (m < m1) {
                const int ox = (int)(m % g.wo), oy = (int)((m / g.wo) % g.ho), b = (int)(m / ((long long)g.wo * g.ho));
                if (cot * 16 + cc < g.cout) vz = dz[(size_t)m * g.cout + cot * 16 + cc];
                const int iy = oy * g.s - g.p + ky, ix = ox * g.s - g.p + kx;
                if ((unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w && cit * 16 + cc < g.cin)
                    vx = x[(((size_t)b * g.h + iy) * g.w + ix) * g.cin + cit * 16 + cc];
            }
            sz[pp][cc] = vz;
            sx[pp][cc] = vx;
        }
        __syncthreads();
#pragma unroll
        for (int pp = 0; pp < 16; ++pp) {
            acc = fmaf(sz[pp][tco], sx[pp][tci], acc);
            accb += sz[pp][tco];
        }
        __syncthreads();
    }
    if (co < g.cout && ci < g.cin) atomicAdd(dw + ((size_t)co * taps + tap) * g.cin + ci, acc);
    if (db && tap == 0 && cit == 0 && tci == 0 && co < g.cout) atomicAdd(db + co, accb);
}

// The same sums for STRIDE 2 with one thread per result pixel: wave w of a workgroup owns the pixels of parity class
// (py, px) = (w >> 1, w & 1) of one row pair -- 64 of them, every other column -- so the taps that reach them, ky = (iy + p) mod 2
// (+ 2, ...) and the same in x, are the same for the whole wave: no per-lane tap test, the weights of a (tap, channel quad) are one
// broadcast LDS read, a lane walks the 64 channels of its four source pixels with 16-byte loads and keeps its cin sums in
// registers (the 16-lanes-per-pixel form above re-reads every source pixel through shuffles and reductions: 3.3 ms per readout
// launch at 32 rows x 512^2, cache-traffic bound).  h and w even.
__global__ __launch_bounds__(256) void t_conv_dgrad_smalln_s2(TConv g, const float* dz, const float* w, const float* bias, float* dx) {
    extern __shared__ float4 sn_w[];  // [taps][cout]
    const int taps = g.k * g.k;
    for (int i = threadIdx.x; i < taps * g.cout; i += 256) {
        const int tap = i / g.cout, co = i - tap * g.cout;
        const float* wp = w + ((size_t)co * taps + tap) * g.cin;
        sn_w[i] = make_float4(wp[0], g.cin > 1 ? wp[1] : 0.0f, g.cin > 2 ? wp[2] : 0.0f, g.cin > 3 ? wp[3] : 0.0f);
    }
    __syncthreads();
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int py = wave >> 1, px = wave & 1;
    const int wc = (g.w / 2 + 63) / 64, hh = g.h / 2;
    int bid = blockIdx.x;
    const int cx = bid % wc;
    bid /= wc;
    const int ry = bid % hh, b = bid / hh;
    const int iy = 2 * ry + py, ix = 2 * (cx * 64 + lane) + px;
    const bool live = ix < g.w;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    for (int ky = (iy + g.p) & 1; ky < g.k; ky += 2) {
        const int ty = iy + g.p - ky, oy = ty >> 1;
        if (ty < 0 || oy >= g.ho) continue;  // wave-uniform
        for (int kx = (px + g.p) & 1; kx < g.k; kx += 2) {
            const int tx = ix + g.p - kx, ox = tx >> 1;
            if (!(live && tx >= 0 && ox < g.wo)) continue;  // lanes at the left / right border only
            const float* zp = dz + (((size_t)b * g.ho + oy) * g.wo + ox) * g.cout;
            const float4* wt = sn_w + (size_t)(ky * g.k + kx) * g.cout;
            for (int c0 = 0; c0 < g.cout; c0 += 4) {
                const float4 z = *(const float4*)(zp + c0);
                const float4 w0 = wt[c0], w1 = wt[c0 + 1], w2 = wt[c0 + 2], w3 = wt[c0 + 3];
                a0 = fmaf(z.x, w0.x, fmaf(z.y, w1.x, fmaf(z.z, w2.x, fmaf(z.w, w3.x, a0))));
                a1 = fmaf(z.x, w0.y, fmaf(z.y, w1.y, fmaf(z.z, w2.y, fmaf(z.w, w3.y, a1))));
                a2 = fmaf(z.x, w0.z, fmaf(z.y, w1.z, fmaf(z.z, w2.z, fmaf(z.w, w3.z, a2))));
                a3 = fmaf(z.x, w0.w, fmaf(z.y, w1.w, fmaf(z.z, w2.w, fmaf(z.w, w3.w, a3))));
            }
        }
    }
    if (!live) return;
    float* o = dx + (((size_t)b * g.h + iy) * g.w + ix) * g.cin;
    o[0] = a0 + (bias ? bias[0] : 0.0f);
    if (g.cin > 1) o[1] = a1 + (bias ? bias[1] : 0.0f);
    if (g.cin > 2) o[2] = a2 + (bias ? bias[2] : 0.0f);
    if (g.cin > 3) o[3] = a3 + (bias ? bias[3] : 0.0f);
}

// The 4 x 4 / stride 2 / pad 1 case with 64 source channels (the readout's transposed conv) with the source rows in LDS: the form
// above has every lane walk the 256-byte rows of its four source pixels alone (16 bytes of a line per load instruction, each
// line fetched again by the three other parity classes: 2.0 ms per launch at 32 rows x 512^2, 8 ms of a 108 ms step).  Here the
// workgroup's three source rows x 66 columns are staged once with coalesced loads (pixel pitch 68 floats: the lanes' 16-byte
// reads fall on distinct banks) and serve all four parity classes.
constexpr int CT_COLS = 66, CT_PITCH = 68;
__global__ __launch_bounds__(256) void t_conv_dgrad_smalln_s2_rows(TConv g, const float* dz, const float* w, const float* bias, float* dx, int pairs_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float ct_smem[];
    float4* sn_w = (float4*)ct_smem;              // [16 taps][64 co] x (cin <= 4 weights)
    float* zs = ct_smem + 16 * 64 * 4;            // [3 rows][66 columns][68]
    // (the weight table costs 3 072 strided 4-byte loads: a workgroup builds it once and walks pairs_per_wg output row pairs with it)
    for (int i = threadIdx.x; i < 16 * 64; i += 256) {
        const int tap = i >> 6, co = i & 63;
        const float* wp = w + ((size_t)co * 16 + tap) * g.cin;
        sn_w[i] = make_float4(wp[0], g.cin > 1 ? wp[1] : 0.0f, g.cin > 2 ? wp[2] : 0.0f, g.cin > 3 ? wp[3] : 0.0f);
    }
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int py = wave >> 1, px = wave & 1;
    const int wc = (g.w / 2 + 63) / 64, hh = g.h / 2, groups = (hh + pairs_per_wg - 1) / pairs_per_wg;
    int bid = blockIdx.x;
    const int cx = bid % wc;
    bid /= wc;
    const int rg = bid % groups, b = bid / groups;
    const int ix = 2 * (cx * 64 + lane) + px;
    const float b0 = bias ? bias[0] : 0.0f, b1 = bias && g.cin > 1 ? bias[1] : 0.0f, b2 = bias && g.cin > 2 ? bias[2] : 0.0f,
                b3 = bias && g.cin > 3 ? bias[3] : 0.0f;
    for (int ry = rg * pairs_per_wg; ry < min(hh, (rg + 1) * pairs_per_wg); ++ry) {
        __syncthreads();  // the weight table is complete / every wave is done with the previous rows
        for (int idx = threadIdx.x; idx < 3 * CT_COLS * 16; idx += 256) {
            const int r = idx / (CT_COLS * 16), rem = idx - r * (CT_COLS * 16), pc = rem >> 4, q = rem & 15;
            const int oy = ry - 1 + r, ox = cx * 64 - 1 + pc;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)oy < (unsigned)g.ho && (unsigned)ox < (unsigned)g.wo) v = *(const float4*)(dz + (((size_t)b * g.ho + oy) * g.wo + ox) * 64 + q * 4);
            *(float4*)(zs + (r * CT_COLS + pc) * CT_PITCH + q * 4) = v;
        }
        __syncthreads();
        const int iy = 2 * ry + py;
        float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
#pragma unroll
        for (int jy = 0; jy < 2; ++jy) {
            const int ky = ((py + 1) & 1) + 2 * jy, oy = (iy + 1 - ky) >> 1;   // rows outside the image were staged as zeros
            const int r = oy - (ry - 1);
#pragma unroll
            for (int jx = 0; jx < 2; ++jx) {
                const int kx = ((px + 1) & 1) + 2 * jx, ox = (ix + 1 - kx) >> 1;
                const float* zp = zs + (r * CT_COLS + (ox - (cx * 64 - 1))) * CT_PITCH;
                const float4* wt = sn_w + (ky * 4 + kx) * 64;
#pragma unroll 4
                for (int c0 = 0; c0 < 64; c0 += 4) {
                    const float4 z = *(const float4*)(zp + c0);
                    const float4 w0 = wt[c0], w1 = wt[c0 + 1], w2 = wt[c0 + 2], w3 = wt[c0 + 3];
                    a0 = fmaf(z.x, w0.x, fmaf(z.y, w1.x, fmaf(z.z, w2.x, fmaf(z.w, w3.x, a0))));
                    a1 = fmaf(z.x, w0.y, fmaf(z.y, w1.y, fmaf(z.z, w2.y, fmaf(z.w, w3.y, a1))));
                    a2 = fmaf(z.x, w0.z, fmaf(z.y, w1.z, fmaf(z.z, w2.z, fmaf(z.w, w3.z, a2))));
                    a3 = fmaf(z.x, w0.w, fmaf(z.y, w1.w, fmaf(z.z, w2.w, fmaf(z.w, w3.w, a3))));
                }
            }
        }
        if (ix < g.w) {
            float* o = dx + (((size_t)b * g.h + iy) * g.w + ix) * g.cin;
            o[0] = a0 + b0;
            if (g.cin > 1) o[1] = a1 + b1;
            if (g.cin > 2) o[2] = a2 + b2;
            if (g.cin > 3) o[3] = a3 + b3;
        }
    }
}

// Weight gradient of a conv with a handful of INPUT channels (the 1x1 stem, cin = 5; the readout's transposed conv, cin = 3,
// 16 taps), cout a multiple of 64: the tiled kernel above spends two barriers per 16 pixels on a 16 x 16 tile of which 3 - 5
// columns exist (3.3 ms per launch at 32 rows).  Here a wave owns 64 output channels (lane = co) and walks its share of the
// workgroup's pixels: dz[m][co] is one coalesced load, the shifted x pixel of a tap is the same for the whole wave (scalar
// loads), and every lane keeps all taps x cin partial sums in registers (NACC = taps * cin <= 64).  The four waves of a workgroup
// take every fourth pixel; their sums meet in LDS and leave with one atomic per (co, tap, ci) and workgroup.
template <int K, int CIN>
__global__ __launch_bounds__(256) void t_conv_wgrad_smallc(TConv g, const float* dz, const float* x, float* dw, float* db, int pix_per_block) {
    constexpr int NACC = K * K * CIN;
    __shared__ float red[3][NACC + 1][64];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cob = blockIdx.y * 64;
    const long long M = (long long)g.n * g.ho * g.wo;
    const long long m0 = (long long)blockIdx.x * pix_per_block, m1 = m0 + pix_per_block < M ? m0 + pix_per_block : M;
    float acc[NACC], accb = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = 0.0f;
    for (long long m = m0 + wave; m < m1; m += 4) {
        const float z = dz[(size_t)m * g.cout + cob + lane];
        accb += z;
        const int ox = (int)(m % g.wo), oy = (int)((m / g.wo) % g.ho), b = (int)(m / ((long long)g.wo * g.ho));
#pragma unroll
        for (int ky = 0; ky < K; ++ky) {
            const int iy = oy * g.s - g.p + ky;
#pragma unroll
            for (int kx = 0; kx < K; ++kx) {
                const int ix = ox * g.s - g.p + kx;
                const bool in = (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w;  // wave-uniform
                const float* xp = x + (((size_t)b * g.h + (in ? iy : 0)) * g.w + (in ? ix : 0)) * CIN;
#pragma unroll
                for (int ci = 0; ci < CIN; ++ci) acc[(ky * K + kx) * CIN + ci] = fmaf(z, in ? xp[ci] : 0.0f, acc[(ky * K + kx) * CIN + ci]);
            }
        }
    }
    if (wave > 0) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) red[wave - 1][i][lane] = acc[i];
        red[wave - 1][NACC][lane] = accb;
    }
    __syncthreads();
    if (wave > 0) return;
    const int co = cob + lane;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
        atomicAdd(dw + (size_t)co * NACC + i, acc[i] + red[0][i][lane] + red[1][i][lane] + red[2][i][lane]);  // dw[co][tap][ci]
    if (db) atomicAdd(db + co, accb + red[0][NACC][lane] + red[1][NACC][lane] + red[2][NACC][lane]);
}

// The same weight gradient on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: fp32 operands, so it serves both operand modes):
// dW[co][j] = sum_p dz[p][co] * patch[p][j], j = (tap, ci) padded to 32 NT columns, one extra column of ones collects the bias gradient.
// A wave walks its pixel range two pixels per MFMA (lanes hi = 0 / 1): the A fragment is the pixels' 128-byte row of dz, the B fragment
// is gathered by per-lane CONSTANT offsets (lane j's tap and channel never change) from the pixel pair's base, taps outside the image
// predicated to zero.  (The form above keeps taps x cin sums per lane and feeds them from scalar loads, one pixel at a time: 2.5 ms per
// launch for the readout's 4 x 4 x 3 at 32 rows, 1.0 ms for the 7 x 7 x 2 init conv -- its traffic time is ~0.1 ms.)  wo even.
typedef __attribute__((ext_vector_type(16))) float tw_f32x16;
template <int NT>
__global__ __launch_bounds__(256) void t_conv_wgrad_smallc_mfma(TConv g, const float* __restrict__ dz, const float* __restrict__ x,
                                                                float* __restrict__ dw, float* __restrict__ db, int pairs_per_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), l31 = lane & 31, hi = lane >> 5;
    const int cob = blockIdx.y * 64, nacc = g.k * g.k * g.cin;
    const long long total_pairs = (long long)g.n * g.ho * g.wo / 2;
    const long long pair0 = ((long long)blockIdx.x * 4 + wave) * pairs_per_wave;
    const long long pair1 = pair0 + pairs_per_wave < total_pairs ? pair0 + pairs_per_wave : total_pairs;
    int jy[NT], jx[NT], jc[NT], jkind[NT];  // column j = 32 t + l31: tap row / column, channel; kind 0 = padding, 1 = weight, 2 = ones
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = t * 32 + l31, tap = j / g.cin;
        jkind[t] = j < nacc ? 1 : j == nacc ? 2 : 0;
        jy[t] = tap / g.k;
        jx[t] = tap - jy[t] * g.k;
        jc[t] = j - tap * g.cin;
    }
    tw_f32x16 acc[2][NT];
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.0f;
    if (pair0 < pair1) {
        const long long p0 = 2 * pair0;
        int ox0 = (int)(p0 % g.wo), oy = (int)((p0 / g.wo) % g.ho), b = (int)(p0 / ((long long)g.wo * g.ho));
        constexpr int UN = 4;  // pixel pairs whose operands are requested before the first MFMA (a wave has one or two neighbours on its SIMD)
        for (long long pair = pair0; pair < pair1; pair += UN) {
            float a0[UN], a1[UN], bv[UN][NT];
#pragma unroll
            for (int u = 0; u < UN; ++u) {
                const bool live = pair + u < pair1;  // wave-uniform
                const size_t p = (size_t)(2 * (live ? pair + u : pair) + hi);
                a0[u] = live ? dz[p * g.cout + cob + l31] : 0.0f;
                a1[u] = live ? dz[p * g.cout + cob + 32 + l31] : 0.0f;
                const int iy0 = oy * g.s - g.p, ix0 = (ox0 + hi) * g.s - g.p;
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int iy = iy0 + jy[t], ix = ix0 + jx[t];
                    bv[u][t] = jkind[t] == 2 ? 1.0f : 0.0f;
                    if (live && jkind[t] == 1 && (unsigned)iy < (unsigned)g.h && (unsigned)ix < (unsigned)g.w)
                        bv[u][t] = x[(((size_t)b * g.h + iy) * g.w + ix) * g.cin + jc[t]];
                }
                ox0 += 2;
                if (ox0 >= g.wo) {
                    ox0 = 0;
                    if (++oy >= g.ho) { oy = 0; ++b; }
                }
            }
#pragma unroll
            for (int u = 0; u < UN; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    acc[0][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u], bv[u][t], acc[0][t], 0, 0, 0);
                    acc[1][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[u], bv[u][t], acc[1][t], 0, 0, 0);
                }
        }
    }
    // D[i][j]: lane j = l31, register r -> row i = 8 (r >> 2) + 4 hi + (r & 3)
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int j = t * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = cob + c * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
                if (j < nacc) atomicAdd(dw + (size_t)co * nacc + j, acc[c][t][r]);   // dw[co][tap][ci]
                else if (j == nacc && db) atomicAdd(db + co, acc[c][t][r]);
            }
        }
#endif
}

// ------------------------------------------------------------------------------------------------ normalisation + FiLM + act + dropout
// per-(sample, channel) sums over the plane: S[n][c] = sum z, Q[n][c] = sum z^2   (fp64 accumulators, zero-initialised)
__global__ __launch_bounds__(256) void t_nc_sums(const float* z, int hw, int C, int px_per_block, double* S, double* Q) {
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * px_per_block, p1 = min(p0 + px_per_block, hw);
    const int nsub = (C < 256 && 256 % C == 0) ? 256 / C : 1, sub = nsub > 1 ? (int)threadIdx.x / C : 0;
    for (int c = nsub > 1 ? (int)threadIdx.x % C : (int)threadIdx.x; c < C; c += (nsub > 1 ? C : (int)blockDim.x)) {
        double s = 0.0, q = 0.0;
        const float* zp = z + ((size_t)b * hw) * C + c;
        for (int p = p0 + sub; p < p1; p += nsub) {
            const double v = zp[(size_t)p * C];
            s += v;
            q += v * v;
        }
        if (nsub > 1) {  // the pixel groups of a channel meet in LDS: one atomic per (workgroup, channel) instead of nsub
            __shared__ double red[2][256];
            red[0][threadIdx.x] = s;
            red[1][threadIdx.x] = q;
            __syncthreads();
            if (sub != 0) continue;
            for (int j = 1; j < nsub; ++j) {
                s += red[0][j * C + c];
                q += red[1][j * C + c];
            }
        }
        atomicAdd(S + (size_t)b * C + c, s);
        atomicAdd(Q + (size_t)b * C + c, q);
    }
}

// kind 0: BatchNorm, batch statistics (index = channel; also updates the running statistics, momentum 0.1, unbiased var)
// kind 1: BatchNorm, running statistics (eval);  kind 2: GroupNorm (index = sample * groups + group)
__global__ void t_stats_finalize(int kind, const double* S, const double* Q, int n, int hw, int C, int groups, float* rmean, float* rvar,
                                 float* mean, float* rstd) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (kind == 2) {
        if (i >= n * groups) return;
        const int b = i / groups, gq = i % groups, cpg = C / groups;
        double s = 0.0, q = 0.0;
        for (int c = gq * cpg; c < (gq + 1) * cpg; ++c) {
            s += S[(size_t)b * C + c];
            q += Q[(size_t)b * C + c];
        }
        const double cnt = (double)hw * cpg, m = s / cnt, var = q / cnt - m * m;
        mean[i] = (float)m;
        rstd[i] = (float)(1.0 / sqrt((var > 0.0 ? var : 0.0) + 1e-5));
        return;
    }
    if (i >= C) return;
    if (kind == 1) {
        mean[i] = rmean[i];
        rstd[i] = 1.0f / sqrtf(rvar[i] + 1e-5f);
        return;
    }
    double s = 0.0, q = 0.0;
    for (int b = 0; b < n; ++b) {
        s += S[(size_t)b * C + i];
        q += Q[(size_t)b * C + i];
    }
    const double cnt = (double)n * hw, m = s / cnt, var = fmax(q / cnt - m * m, 0.0);
    mean[i] = (float)m;
    rstd[i] = (float)(1.0 / sqrt(var + 1e-5));
    rmean[i] = 0.9f * rmean[i] + 0.1f * (float)m;                                      // nn.BatchNorm2d, momentum 0.1
    rvar[i] = 0.9f * rvar[i] + 0.1f * (float)(cnt > 1.0 ? var * cnt / (cnt - 1.0) : var);
}

struct TNorm {
    int n, hw, C, groups, gn, act;      // gn: statistics index = (sample, group) instead of channel
    const float *mean, *rstd, *gamma, *beta, *ss;  // ss [n][2C]: FiLM (scale | shift) or null
    int drop;                           // dropout on
    float drop_scale;
    uint32_t thresh16;
    RngKey salt;
    const uint32_t* row_keys;
};

__device__ __forceinline__ float t_act(float u, int act) {
    if (act == ACT_SILU) return u / (1.0f + expf(-u));
    return act == ACT_RELU ? fmaxf(u, 0.0f) : act == ACT_LEAKY ? (u > 0.0f ? u : 0.2f * u) : u;
}
__device__ __forceinline__ float t_dact(float u, int act) {
    if (act == ACT_SILU) {
        const float sg = 1.0f / (1.0f + expf(-u));
        return sg * (1.0f + u * (1.0f - sg));
    }
    return act == ACT_RELU ? (u > 0.0f ? 1.0f : 0.0f) : act == ACT_LEAKY ? (u > 0.0f ? 1.0f : 0.2f) : 1.0f;
}

__device__ __forceinline__ float t_keep(const TNorm& a, int b, uint32_t e_in_row) {
    if (!a.drop) return 1.0f;
    const RngKey rk = rng_stream_key(RngKey{a.row_keys[2 * b], a.row_keys[2 * b + 1]}, a.salt);
    return rng_keep(e_in_row, rk, a.thresh16) ? a.drop_scale : 0.0f;
}

// Dropout on a whole NHWC tensor (input_dropout on init_conv's output, unet_simple.py:116,168): y = keep ? x / (1 - p) : 0 with the
// engine's per-(forward, global row, site) streams; its adjoint is the same map, so one kernel serves both passes (in place).
__global__ void t_dropout_map(const float* x, float* y, int n, long long per, float scale, uint32_t thresh16, RngKey salt, const uint32_t* row_keys) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= per * n) return;
    const int b = (int)(i / per);
    const RngKey rk = rng_stream_key(RngKey{row_keys[2 * b], row_keys[2 * b + 1]}, salt);
    y[i] = rng_keep((uint32_t)(i - (long long)b * per), rk, thresh16) ? x[i] * scale : 0.0f;
}

// Element-wise passes over an NHWC tape: four channels per lane when the channel count and the pointers allow (grid.y = sample, so no
// 64-bit division; the tensor, gamma / beta, the FiLM pair and the BatchNorm statistics as 16-byte loads issued together), one element
// per lane otherwise.  The arithmetic per element is the same expression either way.
struct TNormQuad {
    float mu[4], rs[4], ga[4], be[4], sc[4], sh[4];
    float s1[4], s2[4];  // backward only
};
__device__ __forceinline__ void t_f4(const float* p, float* o) {
    const float4 v = *(const float4*)p;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void t_norm_quad(const TNorm& a, int b, int c, const float* S1, const float* S2, TNormQuad& q) {
    t_f4(a.gamma + c, q.ga);
    t_f4(a.beta + c, q.be);
    if (a.ss) {
        t_f4(a.ss + (size_t)b * 2 * a.C + c, q.sc);
        t_f4(a.ss + (size_t)b * 2 * a.C + a.C + c, q.sh);
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) q.sc[k] = q.sh[k] = 0.0f;
    }
    if (!a.gn) {
        t_f4(a.mean + c, q.mu);
        t_f4(a.rstd + c, q.rs);
        if (S1) {
            t_f4(S1 + c, q.s1);
            t_f4(S2 + c, q.s2);
        }
    } else {
        const int cpg = a.C / a.groups;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int idx = b * a.groups + ((cpg & 3) == 0 ? c / cpg : (c + k) / cpg);
            q.mu[k] = a.mean[idx];
            q.rs[k] = a.rstd[idx];
            if (S1) {
                q.s1[k] = S1[idx];
                q.s2[k] = S2[idx];
            }
        }
    }
}
__device__ __forceinline__ RngKey t_row_stream(const TNorm& a, int b) {
    return a.drop ? rng_stream_key(RngKey{a.row_keys[2 * b], a.row_keys[2 * b + 1]}, a.salt) : RngKey{0u, 0u};
}
__device__ __forceinline__ float t_keep_rk(const TNorm& a, const RngKey& rk, uint32_t e_in_row) {
    if (!a.drop) return 1.0f;
    return rng_keep(e_in_row, rk, a.thresh16) ? a.drop_scale : 0.0f;
}
__global__ void t_norm_fwd(TNorm a, const float* z, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)a.hw * a.C;
    if (i >= per * a.n) return;
    const int c = (int)(i % a.C), b = (int)(i / per);
    const int idx = a.gn ? b * a.groups + c / (a.C / a.groups) : c;
    float v = (z[i] - a.mean[idx]) * a.rstd[idx] * a.gamma[c] + a.beta[c];
    if (a.ss) v = v * (1.0f + a.ss[(size_t)b * 2 * a.C + c]) + a.ss[(size_t)b * 2 * a.C + a.C + c];
    y[i] = t_act(v, a.act) * t_keep(a, b, (uint32_t)(i - (long long)b * per));
}
__global__ __launch_bounds__(256) void t_norm_fwd4(TNorm a, const float* z, float* y) {
    const uint32_t per = (uint32_t)a.hw * (uint32_t)a.C;
    const uint32_t e = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (e >= per) return;
    const int b = blockIdx.y, c = (int)(e % (uint32_t)a.C);
    const size_t i = (size_t)b * per + e;
    float zv[4], r[4];
    t_f4(z + i, zv);
    TNormQuad q;
    t_norm_quad(a, b, c, nullptr, nullptr, q);
    const RngKey rk = t_row_stream(a, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        float v = (zv[k] - q.mu[k]) * q.rs[k] * q.ga[k] + q.be[k];
        if (a.ss) v = v * (1.0f + q.sc[k]) + q.sh[k];
        r[k] = t_act(v, a.act) * t_keep_rk(a, rk, e + k);
    }
    *(float4*)(y + i) = make_float4(r[0], r[1], r[2], r[3]);
}
inline bool t_vec4_ok(int C, const void* p0, const void* p1 = nullptr, const void* p2 = nullptr) {
    return (C & 3) == 0 && (((uintptr_t)p0 | (uintptr_t)p1 | (uintptr_t)p2) & 15) == 0;
}
inline bool t_norm_vec4_ok(const TNorm& a) {
    return (a.C & 3) == 0 && (long long)a.hw * a.C < (1ll << 32) && a.n <= 65535 &&
           (((uintptr_t)a.mean | (uintptr_t)a.rstd | (uintptr_t)a.gamma | (uintptr_t)a.beta | (uintptr_t)a.ss) & 15) == 0;
}
inline void launch_t_norm_fwd(const TNorm& a, const float* z, float* y, hipStream_t st) {
    const long long per = (long long)a.hw * a.C;
    if (t_norm_vec4_ok(a) && t_vec4_ok(a.C, z, y)) hipLaunchKernelGGL(t_norm_fwd4, dim3(nblk(per / 4), a.n), dim3(256), 0, st, a, z, y);
    else hipLaunchKernelGGL(t_norm_fwd, dim3(nblk(per * a.n)), dim3(256), 0, st, a, z, y);
}

// backward reductions, per (sample, channel) over the plane:
//   A = sum dpre * v (dscale), B = sum dpre (dshift), Cc = sum dbn * xhat, Dd = sum dbn,  dbn = dpre * (1 + scale)
__global__ __launch_bounds__(256) void t_norm_bwd_sums(TNorm a, const float* z, const float* dy, int px_per_block, double* A, double* B,
                                                       double* Cc, double* Dd) {
    const int b = blockIdx.y;
    const int p0 = blockIdx.x * px_per_block, p1 = min(p0 + px_per_block, a.hw);
    const long long per = (long long)a.hw * a.C;
    // fewer channels than threads: 256 / C groups of threads share the block's pixels (coalesced rows, every lane busy)
    const int nsub = (a.C < 256 && 256 % a.C == 0) ? 256 / a.C : 1, sub = nsub > 1 ? (int)threadIdx.x / a.C : 0;
    for (int c = nsub > 1 ? (int)threadIdx.x % a.C : (int)threadIdx.x; c < a.C; c += (nsub > 1 ? a.C : (int)blockDim.x)) {
        const int idx = a.gn ? b * a.groups + c / (a.C / a.groups) : c;
        const float mu = a.mean[idx], rs = a.rstd[idx], ga = a.gamma[c], be = a.beta[c];
        const float sc = a.ss ? a.ss[(size_t)b * 2 * a.C + c] : 0.0f, sh = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + c] : 0.0f;
        double sa = 0.0, sb = 0.0, scx = 0.0, sd = 0.0;
        for (int p = p0 + sub; p < p1; p += nsub) {
            const long long e = (long long)p * a.C + c;
            const float xh = (z[(size_t)b * per + e] - mu) * rs, v = xh * ga + be, u = v * (1.0f + sc) + sh;
            const float dpre = dy[(size_t)b * per + e] * t_keep(a, b, (uint32_t)e) * t_dact(u, a.act);
            const float dbn = dpre * (1.0f + sc);
            sa += (double)dpre * v;
            sb += dpre;
            scx += (double)dbn * xh;
            sd += dbn;
        }
        if (nsub > 1) {  // (as t_nc_sums)
            __shared__ double red[4][256];
            red[0][threadIdx.x] = sa;
            red[1][threadIdx.x] = sb;
            red[2][threadIdx.x] = scx;
            red[3][threadIdx.x] = sd;
            __syncthreads();
            if (sub != 0) continue;
            for (int j = 1; j < nsub; ++j) {
                sa += red[0][j * a.C + c];
                sb += red[1][j * a.C + c];
                scx += red[2][j * a.C + c];
                sd += red[3][j * a.C + c];
            }
        }
        atomicAdd(A + (size_t)b * a.C + c, sa);
        atomicAdd(B + (size_t)b * a.C + c, sb);
        atomicAdd(Cc + (size_t)b * a.C + c, scx);
        atomicAdd(Dd + (size_t)b * a.C + c, sd);
    }
}

// combine: dgamma / dbeta (accumulated), dss[n][2C] = (A | B), and the two sums of the normalisation backward per statistics
// index: S1 = sum gamma*dbn, S2 = sum gamma*dbn*xhat
__global__ void t_norm_bwd_combine(TNorm a, const double* A, const double* B, const double* Cc, const double* Dd, float* g_gamma,
                                   float* g_beta, float* dss, float* S1, float* S2, int batch_stats) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < a.C) {
        double dg = 0.0, db = 0.0;
        for (int b = 0; b < a.n; ++b) {
            dg += Cc[(size_t)b * a.C + i];
            db += Dd[(size_t)b * a.C + i];
        }
        if (g_gamma) {
            g_gamma[i] += (float)dg;
            g_beta[i] += (float)db;
        }
        if (!a.gn) {
            S1[i] = batch_stats ? (float)(a.gamma[i] * db) : 0.0f;   // running statistics: the norm is a fixed affine map
            S2[i] = batch_stats ? (float)(a.gamma[i] * dg) : 0.0f;
        }
    }
    if (dss && i < a.n * a.C) {
        const int b = i / a.C, c = i % a.C;
        dss[(size_t)b * 2 * a.C + c] = (float)A[i];
        dss[(size_t)b * 2 * a.C + a.C + c] = (float)B[i];
    }
    if (a.gn && i < a.n * a.groups) {
        const int b = i / a.groups, gq = i % a.groups, cpg = a.C / a.groups;
        double s1 = 0.0, s2 = 0.0;
        for (int c = gq * cpg; c < (gq + 1) * cpg; ++c) {
            s1 += (double)a.gamma[c] * Dd[(size_t)b * a.C + c];
            s2 += (double)a.gamma[c] * Cc[(size_t)b * a.C + c];
        }
        S1[i] = (float)s1;
        S2[i] = (float)s2;
    }
}

// dz = rstd * (gamma*dbn - (S1 + xhat*S2) / count)
__global__ void t_norm_bwd_apply(TNorm a, const float* z, const float* dy, const float* S1, const float* S2, float inv_count, float* dz) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long per = (long long)a.hw * a.C;
    if (i >= per * a.n) return;
    const int c = (int)(i % a.C), b = (int)(i / per);
    const int idx = a.gn ? b * a.groups + c / (a.C / a.groups) : c;
    const float xh = (z[i] - a.mean[idx]) * a.rstd[idx], v = xh * a.gamma[c] + a.beta[c];
    const float sc = a.ss ? a.ss[(size_t)b * 2 * a.C + c] : 0.0f, sh = a.ss ? a.ss[(size_t)b * 2 * a.C + a.C + c] : 0.0f;
    const float u = v * (1.0f + sc) + sh;
    const float dbn = dy[i] * t_keep(a, b, (uint32_t)(i - (long long)b * per)) * t_dact(u, a.act) * (1.0f + sc);
    dz[i] = a.rstd[idx] * (a.gamma[c] * dbn - (S1[idx] + xh * S2[idx]) * inv_count);
}
__global__ __launch_bounds__(256) void t_norm_bwd_apply4(TNorm a, const float* z, const float* dy, const float* S1, const float* S2, float inv_count,
                                                         float* dz) {
    const uint32_t per = (uint32_t)a.hw * (uint32_t)a.C;
    const uint32_t e = (blockIdx.x * 256u + threadIdx.x) * 4u;
    if (e >= per) return;
    const int b = blockIdx.y, c = (int)(e % (uint32_t)a.C);
    const size_t i = (size_t)b * per + e;
    float zv[4], dv[4], r[4];
    t_f4(z + i, zv);
    t_f4(dy + i, dv);
    TNormQuad q;
    t_norm_quad(a, b, c, S1, S2, q);
    const RngKey rk = t_row_stream(a, b);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float xh = (zv[k] - q.mu[k]) * q.rs[k], v = xh * q.ga[k] + q.be[k];
        const float u = v * (1.0f + q.sc[k]) + q.sh[k];
        const float dbn = dv[k] * t_keep_rk(a, rk, e + k) * t_dact(u, a.act) * (1.0f + q.sc[k]);
        r[k] = q.rs[k] * (q.ga[k] * dbn - (q.s1[k] + xh * q.s2[k]) * inv_count);
    }
    *(float4*)(dz + i) = make_float4(r[0], r[1], r[2], r[3]);
}
inline void launch_t_norm_bwd_apply(const TNorm& a, const float* z, const float* dy, const float* S1, const float* S2, float inv_count, float* dz,
                                    hipStream_t st) {
    const long long per = (long long)a.hw * a.C;
    if (t_norm_vec4_ok(a) && t_vec4_ok(a.C, z, dy, dz) && t_vec4_ok(0, S1, S2))
        hipLaunchKernelGGL(t_norm_bwd_apply4, dim3(nblk(per / 4), a.n), dim3(256), 0, st, a, z, dy, S1, S2, inv_count, dz);
    else hipLaunchKernelGGL(t_norm_bwd_apply, dim3(nblk(per * a.n)), dim3(256), 0, st, a, z, dy, S1, S2, inv_count, dz);
}

// ------------------------------------------------------------------------------------------------ small dense layers (time MLP, FiLM heads)
// y[r][o] = b[o] + sum_k f(x[r][k]) * W[o][k];  f = identity (pre = 0) or SiLU (pre = 1)
// y[r][o] = bias[o] + sum_k f(x[r][k]) W[o][k] (f = SiLU when pre): one wave per output column o, lanes over k (coalesced rows
// of W), every row r of the (small) batch from the same W row
__global__ __launch_bounds__(256) void t_linear_fwd(const float* x, const float* W, const float* bias, int rows, int K, int O, int pre, float* y) {
    // blockIdx.y: 16-row slice of the batch (one slice walked all 64 rows of an OISST step alone, SiLU of the whole input per output
    // column: 63 us per call; launch with grid.y = ceil(rows / 16))
    const int o = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (o >= O) return;
    const int rbeg = blockIdx.y * 16;
    rows = min(rows, rbeg + 16);
    for (int r0 = rbeg; r0 < rows; r0 += 4) {
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = lane; k < K; k += 64) {
            const float wv = W[(size_t)o * K + k];
#pragma unroll
            for (int q = 0; q < 4; ++q)
                if (r0 + q < rows) {
                    float v = x[(size_t)(r0 + q) * K + k];
                    if (pre) v = v / (1.0f + expf(-v));
                    acc[q] = fmaf(v, wv, acc[q]);
                }
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float a = acc[q];
            for (int off = 32; off > 0; off >>= 1) a += __shfl_xor(a, off);
            if (lane == 0 && r0 + q < rows) y[(size_t)(r0 + q) * O + o] = a + bias[o];
        }
    }
}
__global__ void t_linear_bwd_w(const float* x, const float* dy, int rows, int K, int O, int pre, float* dW, float* db) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= O * K) return;
    const int o = i / K, k = i % K;
    float acc = 0.0f, accb = 0.0f;
    for (int r = 0; r < rows; ++r) {
        float v = x[(size_t)r * K + k];
        if (pre) v = v / (1.0f + expf(-v));
        acc = fmaf(dy[(size_t)r * O + o], v, acc);
        accb += dy[(size_t)r * O + o];
    }
    dW[i] += acc;
    if (k == 0) db[o] += accb;
}
// dx[r][k] (+)= f'(x[r][k]) * sum_o dy[r][o] * W[o][k]
// dx[r][k] = (sum_o dy[r][o] W[o][k]) * f'(x[r][k]): a block = one row r x 16 columns k, 16 groups of threads split the O axis
// (one workgroup looping over all of O took 214 us per FiLM head -- pure latency)
__global__ __launch_bounds__(256) void t_linear_bwd_x(const float* x, const float* W, const float* dy, int rows, int K, int O, int pre, int accumulate, float* dx) {
    __shared__ float red[16][17];
    const int kb = (K + 15) / 16;
    const int r = blockIdx.x / kb, k = (blockIdx.x % kb) * 16 + (threadIdx.x & 15), og = threadIdx.x >> 4;
    float acc = 0.0f;
    if (k < K)
        for (int o = og; o < O; o += 16) acc = fmaf(dy[(size_t)r * O + o], W[(size_t)o * K + k], acc);
    red[og][threadIdx.x & 15] = acc;
    __syncthreads();
    if (og == 0 && k < K) {
        for (int q = 1; q < 16; ++q) acc += red[q][threadIdx.x & 15];
        const size_t i = (size_t)r * K + k;
        if (pre) {
            const float v = x[i], sg = 1.0f / (1.0f + expf(-v));
            acc *= sg * (1.0f + v * (1.0f - sg));
        }
        dx[i] = accumulate ? dx[i] + acc : acc;
    }
}
__global__ void t_silu_bwd(const float* x, const float* dsilu, long long n, float* dx) {  // dx = dsilu * silu'(x)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float v = x[i], sg = 1.0f / (1.0f + expf(-v));
    dx[i] = dsilu[i] * sg * (1.0f + v * (1.0f - sg));
}
__global__ void t_sinusoid(const float* t, int rows, int dim, float* out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= rows * dim) return;
    const int r = i / dim, k = i % dim, half = dim / 2, j = k < half ? k : k - half;
    const float ang = t[r] * expf((float)j * (-logf(10000.0f) / (float)(half - 1)));
    out[i] = k < half ? sinf(ang) : cosf(ang);
}
__global__ void t_gelu_fwd(const float* x, long long n, float* y) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) y[i] = 0.5f * x[i] * (1.0f + erff(x[i] * 0.70710678118654752f));
}
__global__ void t_gelu_bwd(const float* x, long long n, float* d) {  // d *= gelu'(x)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) d[i] *= 0.5f * (1.0f + erff(x[i] * 0.70710678118654752f)) + x[i] * 0.3989422804014327f * expf(-0.5f * x[i] * x[i]);
}

// ------------------------------------------------------------------------------------------------ element-wise helpers
// (V = 4: channel quads; ca, cb multiples of 4)
template <int V>
__global__ __launch_bounds__(256) void t_concat2(const float* a, int ca, const float* b, int cb, long long pixels, float* out) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
    const int C = ca + cb;
    if (i >= pixels * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const float* src = c < ca ? a + p * ca + c : b + p * cb + (c - ca);
    if (V == 4) *(float4*)(out + i) = *(const float4*)src;
    else out[i] = *src;
}
inline void launch_t_concat2(const float* a, int ca, const float* b, int cb, long long pixels, float* out, hipStream_t st) {
    const long long total = pixels * (ca + cb);
    if (t_vec4_ok(ca | cb, a, b, out)) hipLaunchKernelGGL(t_concat2<4>, dim3(nblk(total / 4)), dim3(256), 0, st, a, ca, b, cb, pixels, out);
    else hipLaunchKernelGGL(t_concat2<1>, dim3(nblk(total)), dim3(256), 0, st, a, ca, b, cb, pixels, out);
}
// split the gradient of cat[a, b]: da = d[..., :ca] (assign), db += d[..., ca:]
__global__ void t_split2(const float* d, int ca, int cb, long long pixels, float* da, float* db) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int C = ca + cb;
    if (i >= pixels * C) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    if (c < ca) da[p * ca + c] = d[i];
    else db[p * cb + (c - ca)] += d[i];
}
template <int V>
__global__ __launch_bounds__(256) void t_add(float* a, const float* b, long long n) {
    const long long i = ((long long)blockIdx.x * 256 + threadIdx.x) * V;
    if (V == 4 && i + 3 < n) {
        float4 x = *(float4*)(a + i);
        const float4 y = *(const float4*)(b + i);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        *(float4*)(a + i) = x;
        return;
    }
    for (int k = 0; k < V; ++k)
        if (i + k < n) a[i + k] += b[i + k];
}
inline void launch_t_add(float* a, const float* b, long long n, hipStream_t st) {
    if (t_vec4_ok(0, a, b)) hipLaunchKernelGGL(t_add<4>, dim3(nblk((n + 3) / 4)), dim3(256), 0, st, a, b, n);
    else hipLaunchKernelGGL(t_add<1>, dim3(nblk(n)), dim3(256), 0, st, a, b, n);
}
// db[c] += sum_p d[p][c] for any (small) channel count: coalesced sweep of a slice of the tensor, per-block sums in LDS
__global__ __launch_bounds__(256) void t_bias_grad(const float* d, long long pixels, int C, long long elems_per_block, float* db) {
    extern __shared__ float bg_sh[];
    for (int c = threadIdx.x; c < C; c += 256) bg_sh[c] = 0.0f;
    __syncthreads();
    const long long total = pixels * C;
    const long long e0 = (long long)blockIdx.x * elems_per_block, e1 = e0 + elems_per_block < total ? e0 + elems_per_block : total;
    for (long long i = e0 + threadIdx.x; i < e1; i += 256) atomicAdd(&bg_sh[(int)(i % C)], d[i]);
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) atomicAdd(db + c, bg_sh[c]);
}
// the same sums for channel counts that divide 256 or are multiples of 256: coalesced rows, one slice of the pixels per block
__global__ __launch_bounds__(256) void t_bias_grad_rows(const float* d, long long pixels, int C, int rows_per_block, float* db) {
    __shared__ float red[256];
    const long long p0 = (long long)blockIdx.x * rows_per_block, p1 = p0 + rows_per_block < pixels ? p0 + rows_per_block : pixels;
    if (C <= 256) {
        const int c = threadIdx.x % C, sub = threadIdx.x / C, step = 256 / C;
        float s = 0.0f;
        for (long long p = p0 + sub; p < p1; p += step) s += d[p * C + c];
        red[threadIdx.x] = s;
        __syncthreads();
        if (sub == 0) {
            for (int k = 1; k < step; ++k) s += red[k * C + c];
            atomicAdd(db + c, s);
        }
    } else {
        for (int c = threadIdx.x; c < C; c += 256) {
            float s = 0.0f;
            for (long long p = p0; p < p1; ++p) s += d[p * C + c];
            atomicAdd(db + c, s);
        }
    }
}
// d(mean criterion)/d pred * scale: kind 0 L1 (sign), 1 MSE, 2 smooth-L1 (beta 1)
__global__ void t_criterion_grad(const float* pred, const float* target, long long n, int kind, float scale, float* d) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = pred[i] - target[i];
    float gsign = r > 0.0f ? 1.0f : (r < 0.0f ? -1.0f : 0.0f);
    float g = kind == 0 ? gsign : kind == 1 ? 2.0f * r : (fabsf(r) < 1.0f ? r : gsign);
    d[i] = g * scale / (float)n;
}

// ------------------------------------------------------------------------------------------------ host side
std::map<void*, size_t>& g_block_bytes() {  // size of every live training block (for the pool)
    static std::map<void*, size_t> m;
    return m;
}
template <typename T>
dyf_status talloc(dyf_engine* e, std::vector<void*>& owner, T** out, size_t count, bool zero = true) {
    void* p = nullptr;
    const size_t bytes = (std::max<size_t>(count * sizeof(T), 256) + 255) / 256 * 256;
    TrainState* ts = e->train;
    auto it = ts ? ts->pool.find(bytes) : std::multimap<size_t, void*>::iterator();
    if (ts && it != ts->pool.end()) {
        p = it->second;
        ts->pool.erase(it);
        ts->pool_bytes -= bytes;
    } else {
        TK(hipMalloc(&p, bytes));
    }
    if (zero) TK(hipMemsetAsync(p, 0, bytes, ts ? ts->stream : nullptr));
    else {
        // test hook (DYF_TRAIN_POISON=1): blocks handed out without zero-fill start as NaN patterns, so a kernel that consumes a
        // buffer it did not fully write shows up in the gradients instead of hiding behind whatever the block held before
        const bool poison = dyf_form("DYF_TRAIN_POISON") && atoi(dyf_form("DYF_TRAIN_POISON")) != 0;
        if (poison) TK(hipMemsetAsync(p, 0xFF, bytes, ts ? ts->stream : nullptr));
    }
    owner.push_back(p);
    g_block_bytes()[p] = bytes;
    *out = (T*)p;
    return DYF_OK;
}

dyf_status tupload(dyf_engine* e, std::vector<void*>& owner, float** out, const std::vector<float>& h) {
    dyf_status s = talloc(e, owner, out, h.size(), false);
    if (s != DYF_OK) return s;
    TK(hipMemcpy(*out, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice));
    return DYF_OK;
}

void tfree(dyf_engine* e, std::vector<void*>& owner) {  // back to the pool (train_destroy releases the pool)
    for (void* p : owner) {
        auto it = g_block_bytes().find(p);
        if (e->train && it != g_block_bytes().end()) {
            e->train->pool.emplace(it->second, p);
            e->train->pool_bytes += it->second;
        } else {
            (void)hipFree(p);
        }
    }
    owner.clear();
    // safety valve: batch sizes that keep changing leave blocks of every size behind -- past 32 GB of idle blocks, start over
    if (e->train && e->train->pool_bytes > ((size_t)32 << 30)) {
        (void)hipDeviceSynchronize();
        for (auto& kv : e->train->pool) {
            g_block_bytes().erase(kv.second);
            (void)hipFree(kv.second);
        }
        e->train->pool.clear();
        e->train->pool_bytes = 0;
    }
}

void launch_bias_grad(const float* d, long long pixels, int C, float* db, hipStream_t st) {
    const long long total = pixels * C, per = std::max<long long>(4096, (total + 1023) / 1024);
    hipLaunchKernelGGL(t_bias_grad, dim3((unsigned)((total + per - 1) / per)), dim3(256), (size_t)C * sizeof(float), st, d, pixels, C, per, db);
}

constexpr size_t TRAIN_SPLITK_FLOATS = (size_t)16 << 20;  // 64 MB: 512 tiles x 128 x 64 partial sums and change
float* splitk_ws(dyf_engine* e) {
    TrainState* ts = e->train;
    if (ts && !ts->splitk_ws && talloc(e, ts->ws_owned, &ts->splitk_ws, TRAIN_SPLITK_FLOATS, false) != DYF_OK) ts->splitk_ws = nullptr;
    return ts ? ts->splitk_ws : nullptr;
}

// DYF_TRAIN_MFMA=0 keeps the plain VALU kernels (A/B and a second implementation for the tests)
bool train_mfma() {
    const bool on = !(dyf_form("DYF_TRAIN_MFMA") && atoi(dyf_form("DYF_TRAIN_MFMA")) == 0);
    return on;
}

dyf_status conv_fwd(dyf_engine* e, const TConv& g, const float* x, const float* wt, const float* b, float* y, hipStream_t st) {
    if (train_mfma() && tgemm_conv_fwd(g, x, wt, b, y, splitk_ws(e), TRAIN_SPLITK_FLOATS, st)) {
        TK(hipGetLastError());
        return DYF_OK;
    }
    const bool small = !(dyf_form("DYF_TRAIN_SMALLC") && atoi(dyf_form("DYF_TRAIN_SMALLC")) == 0);  // =0: the round-3 VALU forms (A/B)
    const long long Mf = (long long)g.n * g.ho * g.wo;
    if (small && g.cout % 64 == 0 && g.wo % 32 == 0 && g.cin <= 8 && g.k * g.k * g.cin <= 126 && g.k < 256 && Mf >= 4096 &&
        !(dyf_form("DYF_TRAIN_SMALLC_MFMA") && atoi(dyf_form("DYF_TRAIN_SMALLC_MFMA")) == 0)) {
        const long long tiles = Mf / 32;
        const int tpw = (int)std::max<long long>(1, (tiles + 4095) / 4096);  // ~4 096 waves per 64-channel block
        hipLaunchKernelGGL(t_conv_fwd_smallc_mfma, dim3((unsigned)((tiles + 4ll * tpw - 1) / (4ll * tpw)), (unsigned)(g.cout / 64)), dim3(256), 0, st, g, x,
                           wt, b, y, tpw);
    } else if (small && g.cout % 4 == 0)
        hipLaunchKernelGGL(t_conv_fwd4, dim3(nblk((long long)g.n * g.ho * g.wo * (g.cout / 4))), dim3(256), 0, st, g, x, wt, b, y);
    else
        hipLaunchKernelGGL(t_conv_fwd, dim3(nblk((long long)g.n * g.ho * g.wo * g.cout)), dim3(256), 0, st, g, x, wt, b, y);
    TK(hipGetLastError());
    return DYF_OK;
}
dyf_status conv_dgrad(dyf_engine* e, const TConv& g, const float* dz, const float* w, const float* bias, float* dx, hipStream_t st) {
    if (train_mfma() && tgemm_conv_dgrad(g, dz, w, bias, dx, splitk_ws(e), TRAIN_SPLITK_FLOATS, st)) {
        TK(hipGetLastError());
        return DYF_OK;
    }
    const bool small = !(dyf_form("DYF_TRAIN_SMALLC") && atoi(dyf_form("DYF_TRAIN_SMALLC")) == 0);
    if (small && g.s == 2 && g.k == 4 && g.p == 1 && g.cin <= 4 && g.cout == 64 && g.h % 2 == 0 && g.w % 2 == 0 && g.ho == g.h / 2 && g.wo == g.w / 2 &&
        !(dyf_form("DYF_TRAIN_CT_ROWS") && atoi(dyf_form("DYF_TRAIN_CT_ROWS")) == 0)) {
        const int wc = (g.w / 2 + 63) / 64;
        constexpr size_t lds = (size_t)(16 * 64 * 4 + 3 * CT_COLS * CT_PITCH) * sizeof(float);
        if (!train_raise_dynamic_lds(t_conv_dgrad_smalln_s2_rows, (int)lds))
            return fail(e, DYF_ERR_HIP, "hipFuncSetAttribute(MaxDynamicSharedMemorySize) failed for t_conv_dgrad_smalln_s2_rows");
        // output row pairs per workgroup: as many as still leave ~2 048 workgroups
        const long long all_pairs = (long long)g.n * (g.h / 2) * wc;
        const int ppw = (int)std::max<long long>(1, std::min<long long>(16, all_pairs / 2048));
        const int groups = (g.h / 2 + ppw - 1) / ppw;
        hipLaunchKernelGGL(t_conv_dgrad_smalln_s2_rows, dim3((unsigned)((long long)g.n * groups * wc)), dim3(256), lds, st, g, dz, w, bias, dx, ppw);
        TK(hipGetLastError());
        return DYF_OK;
    }
    if (small && g.s == 2 && g.cin <= 4 && g.cout % 4 == 0 && g.h % 2 == 0 && g.w % 2 == 0 && (size_t)g.k * g.k * g.cout * 16 <= 65536) {
        const int wc = (g.w / 2 + 63) / 64;
        hipLaunchKernelGGL(t_conv_dgrad_smalln_s2, dim3((unsigned)((long long)g.n * (g.h / 2) * wc)), dim3(256), (size_t)g.k * g.k * g.cout * 16, st,
                           g, dz, w, bias, dx);
        TK(hipGetLastError());
        return DYF_OK;
    }
    if (g.cin <= 4 && g.cout % 64 == 0 && (size_t)g.k * g.k * g.cout * 16 <= 65536) {
        hipLaunchKernelGGL(t_conv_dgrad_smalln, dim3(nblk((long long)g.n * g.h * g.w * 16)), dim3(256), (size_t)g.k * g.k * g.cout * 16, st, g, dz, w,
                           bias, dx);
        TK(hipGetLastError());
        return DYF_OK;
    }
    hipLaunchKernelGGL(t_conv_dgrad, dim3(nblk((long long)g.n * g.h * g.w * g.cin)), dim3(256), 0, st, g, dz, w, bias, dx);
    TK(hipGetLastError());
    return DYF_OK;
}
dyf_status conv_wgrad(dyf_engine* e, const TConv& g, const float* dz, const float* x, float* dw, float* db, hipStream_t st) {
    const long long M = (long long)g.n * g.ho * g.wo;
    if (train_mfma() && tgemm_conv_wgrad(g, dz, x, dw, st)) {
        if (db) {
            if (256 % g.cout == 0 || g.cout % 256 == 0) {
                const int rpb = (int)std::max<long long>(64, (M + 1023) / 1024);
                hipLaunchKernelGGL(t_bias_grad_rows, dim3((unsigned)((M + rpb - 1) / rpb)), dim3(256), 0, st, dz, M, g.cout, rpb, db);
            } else {
                launch_bias_grad(dz, M, g.cout, db, st);
            }
        }
        TK(hipGetLastError());
        return DYF_OK;
    }
    const bool small = !(dyf_form("DYF_TRAIN_SMALLC") && atoi(dyf_form("DYF_TRAIN_SMALLC")) == 0);
    if (small && g.cout % 64 == 0 && M >= 4096 && g.wo % 2 == 0 && g.k * g.k * g.cin + 1 <= 128 && g.cin <= 8 &&
        !(dyf_form("DYF_TRAIN_SMALLC_MFMA") && atoi(dyf_form("DYF_TRAIN_SMALLC_MFMA")) == 0)) {  // read per call: tests run both forms
        const long long pairs = M / 2;
        const int ppw = (int)std::max<long long>(64, (pairs + 2047) / 2048);  // ~2 048 waves per 64-channel block
        const dim3 grid((unsigned)((pairs + 4ll * ppw - 1) / (4ll * ppw)), (unsigned)(g.cout / 64));
        if (g.k * g.k * g.cin + 1 <= 64)
            hipLaunchKernelGGL(t_conv_wgrad_smallc_mfma<2>, grid, dim3(256), 0, st, g, dz, x, dw, db, ppw);
        else
            hipLaunchKernelGGL(t_conv_wgrad_smallc_mfma<4>, grid, dim3(256), 0, st, g, dz, x, dw, db, ppw);
        TK(hipGetLastError());
        return DYF_OK;
    }
    if (small && g.cout % 64 == 0 && M >= 4096) {
        const int cob = g.cout / 64;
        const long long blocks = std::max<long long>(1, std::min<long long>(M / 512, 1024 / cob));  // >= 512 pixels per workgroup
        const int ppb = (int)((M + blocks - 1) / blocks);
        const dim3 grid((unsigned)((M + ppb - 1) / ppb), (unsigned)cob);
#define SMALLC(KK, CC) if (g.k == KK && g.cin == CC) { hipLaunchKernelGGL((t_conv_wgrad_smallc<KK, CC>), grid, dim3(256), 0, st, g, dz, x, dw, db, ppb); launched = true; }
        bool launched = false;
        SMALLC(1, 1) SMALLC(1, 2) SMALLC(1, 3) SMALLC(1, 4) SMALLC(1, 5) SMALLC(1, 6) SMALLC(1, 7) SMALLC(1, 8)
        SMALLC(4, 1) SMALLC(4, 2) SMALLC(4, 3) SMALLC(4, 4)
        SMALLC(7, 1) SMALLC(7, 2)  // the ResNet-UNet's 7 x 7 init conv on 1-2 input channels (98 sums per lane)
#undef SMALLC
        if (launched) {
            TK(hipGetLastError());
            return DYF_OK;
        }
    }
    const int tiles = g.k * g.k * ((g.cout + 15) / 16) * ((g.cin + 15) / 16);
    long long slices = std::max<long long>(1, std::min<long long>((M + 255) / 256, (4096 + tiles - 1) / tiles));
    const int ppb = (int)(((M + slices - 1) / slices + 15) / 16 * 16);
    slices = (M + ppb - 1) / ppb;
    hipLaunchKernelGGL(t_conv_wgrad, dim3((unsigned)(tiles * slices)), dim3(256), 0, st, g, dz, x, dw, db, ppb);
    TK(hipGetLastError());
    return DYF_OK;
}

TConv block_geom(const UBlock& b, int nb) {
    return TConv{nb, b.in_h, b.in_w, b.cin, b.out_h, b.out_w, b.cout, b.k, b.stride, b.pad};
}

}  // namespace

#include "train_resnet.inc"  // recorded forward / backward of the ResNet-UNet (arch unet.Unet)

namespace dyf {

void train_destroy(dyf_engine* e) {
    if (!e->train) return;
    rn_train_destroy(e);
    for (auto& n : e->train->net) tfree(e, n.owned);
    for (auto& t : e->train->tape) tfree(e, t.owned);
    tfree(e, e->train->ws_owned);
    for (auto& kv : e->train->pool) {
        g_block_bytes().erase(kv.second);
        (void)hipFree(kv.second);
    }
    delete e->train;
    e->train = nullptr;
}

// called by dyf_load_weights (arch unet_simple): keep an fp32 copy of the parameters in the training layout
dyf_status train_store_weights(dyf_engine* e, int which, std::map<std::string, TensorView>& sd) {
    if (!e->train) e->train = new TrainState();
    TNet& t = e->train->net[which];
    const Net& n = e->net[which];
    TK(hipDeviceSynchronize());
    tfree(e, t.owned);
    t = TNet{};
    auto V = [&](const std::string& k) { const TensorView& v = sd.at(k); return std::vector<float>(v.data, v.data + v.numel()); };
    auto grad = [&](float** g, size_t cnt) -> dyf_status {
        dyf_status s = talloc(e, t.owned, g, cnt);
        if (s == DYF_OK) t.grads.emplace_back(*g, cnt);
        return s;
    };
#define TS(expr) do { dyf_status _s = (expr); if (_s != DYF_OK) return _s; } while (0)
    for (int i = 0; i < 12; ++i) {
        const UBlock& b = n.blk[i];
        TBlockW& w = t.blk[i];
        const std::string pre = (i < 6 ? "input_ops." + std::to_string(i) : "output_ops." + std::to_string(i - 6));
        const std::string conv = pre + ".ops." + (b.transposed ? "1" : "0"), norm = pre + ".ops." + (b.transposed ? "2" : "1");
        const std::vector<float> cw = V(conv + ".weight");
        const int taps = b.k * b.k;
        std::vector<float> a((size_t)b.cout * taps * b.cin), at(a.size());
        for (int co = 0; co < b.cout; ++co)
            for (int ci = 0; ci < b.cin; ++ci)
                for (int tp = 0; tp < taps; ++tp) {
                    const float v = cw[((size_t)co * b.cin + ci) * taps + tp];
                    a[((size_t)co * taps + tp) * b.cin + ci] = v;
                    at[((size_t)tp * b.cin + ci) * b.cout + co] = v;
                }
        TS(tupload(e, t.owned, &w.w, a)); TS(tupload(e, t.owned, &w.wt, at));
        TS(tupload(e, t.owned, &w.b, V(conv + ".bias")));
        TS(tupload(e, t.owned, &w.gamma, V(norm + ".weight"))); TS(tupload(e, t.owned, &w.beta, V(norm + ".bias")));
        if (!b.gn) { TS(tupload(e, t.owned, &w.rmean, V(norm + ".running_mean"))); TS(tupload(e, t.owned, &w.rvar, V(norm + ".running_var"))); }
        TS(grad(&w.g_w, a.size())); TS(grad(&w.g_b, b.cout)); TS(grad(&w.g_gamma, b.cout)); TS(grad(&w.g_beta, b.cout));
        if (n.cfg.with_time_emb) {
            TS(tupload(e, t.owned, &w.fw, V(pre + ".time_mlp.1.weight"))); TS(tupload(e, t.owned, &w.fb, V(pre + ".time_mlp.1.bias")));
            TS(grad(&w.g_fw, (size_t)2 * b.cout * n.tdim)); TS(grad(&w.g_fb, (size_t)2 * b.cout));
        }
    }
    if (n.cfg.with_time_emb) {
        TS(tupload(e, t.owned, &t.t_w1, V("time_emb_mlp.1.weight"))); TS(tupload(e, t.owned, &t.t_b1, V("time_emb_mlp.1.bias")));
        TS(tupload(e, t.owned, &t.t_w2, V("time_emb_mlp.3.weight"))); TS(tupload(e, t.owned, &t.t_b2, V("time_emb_mlp.3.bias")));
        TS(grad(&t.g_t_w1, (size_t)n.tdim * n.dim)); TS(grad(&t.g_t_b1, n.tdim)); TS(grad(&t.g_t_w2, (size_t)n.tdim * n.tdim)); TS(grad(&t.g_t_b2, n.tdim));
    }
    {
        const std::vector<float> sw = V("init_conv.weight");  // [dim][cin] (1x1)
        std::vector<float> swt(sw.size());
        for (int d = 0; d < n.dim; ++d)
            for (int c = 0; c < n.cin_total; ++c) swt[(size_t)c * n.dim + d] = sw[(size_t)d * n.cin_total + c];
        TS(tupload(e, t.owned, &t.stem_w, sw)); TS(tupload(e, t.owned, &t.stem_wt, swt)); TS(tupload(e, t.owned, &t.stem_b, V("init_conv.bias")));
        TS(grad(&t.g_stem_w, sw.size())); TS(grad(&t.g_stem_b, n.dim));
    }
    {   // readout ConvTranspose2d(dim -> C, k4, s2, p1) as the dgrad form of a conv C: (C ch, 2h x 2w) -> (dim ch, h x w):
        // Wc[co = dim][tap][ci = C] = W_T[co][ci][ky][kx]
        const std::vector<float> rw = V("readout.0.weight");
        const int oc = n.cfg.out_channels;
        std::vector<float> a((size_t)n.dim * 16 * oc), at(a.size());
        for (int co = 0; co < n.dim; ++co)
            for (int ci = 0; ci < oc; ++ci)
                for (int tp = 0; tp < 16; ++tp) {
                    const float v = rw[((size_t)co * oc + ci) * 16 + tp];
                    a[((size_t)co * 16 + tp) * oc + ci] = v;
                    at[((size_t)tp * oc + ci) * n.dim + co] = v;
                }
        TS(tupload(e, t.owned, &t.ro_w, a)); TS(tupload(e, t.owned, &t.ro_wt, at)); TS(tupload(e, t.owned, &t.ro_b, V("readout.0.bias")));
        TS(grad(&t.g_ro_w, a.size())); TS(grad(&t.g_ro_b, oc));
    }
#undef TS
    t.ready = true;
    return DYF_OK;
}

}  // namespace dyf

extern "C" {

dyf_status dyf_train_zero_grads(dyf_engine* e, int32_t which) {
    if (!e || which < 0 || which > 1) return DYF_ERR_INVALID_ARGUMENT;
    if (e->net[which].rn) { TK(hipSetDevice(e->cfg.device)); return rn_train_zero_grads(e, which); }
    if (!e->train || !e->train->net[which].ready) return fail(e, DYF_ERR_STATE, "training needs arch unet_simple / unet with loaded weights");
    TK(hipSetDevice(e->cfg.device));
    for (auto& g : e->train->net[which].grads) TK(hipMemsetAsync(g.first, 0, g.second * sizeof(float), 0));
    TK(hipDeviceSynchronize());
    return DYF_OK;
}

dyf_status dyf_train_set_precision(dyf_engine* e, int32_t bits) {
    if (!e || (bits != 0 && bits != 16 && bits != 32)) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_set_precision: bits must be 0, 16 or 32");
    e->train_precision = bits;
    return DYF_OK;
}
int32_t dyf_train_precision(const dyf_engine* e) { return e ? e->train_precision : -1; }

dyf_status dyf_train_forward(dyf_engine* e, int32_t which, int32_t slot, const float* inputs_dev, const float* time_dev,
                             const float* cond_dev, float* out_dev, int32_t nb, int32_t flags, void* stream) {
    if (!e || which < 0 || which > 1 || slot < 0 || slot > 3 || !inputs_dev || !out_dev || nb < 1)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_forward: bad arguments");
    const TrainPrecisionScope precision(e->train_precision);
    if (e->net[which].rn) {
        TK(hipSetDevice(e->cfg.device));
        if (e->train) {  // the slot now belongs to this forward: drop a unet_simple tape that may sit in it
            tfree(e, e->train->tape[slot].owned);
            e->train->tape[slot] = TTape{};
        }
        return rn_train_forward(e, which, slot, inputs_dev, time_dev, cond_dev, out_dev, nb, flags, (hipStream_t)stream);
    }
    if (e->net[which].sc) return fail(e, DYF_ERR_UNSUPPORTED, "training step: arch unet_simple and unet (SimpleConvNet is the CPU plumbing config)");
    if (!e->train || !e->train->net[which].ready) return fail(e, DYF_ERR_STATE, "training needs arch unet_simple with loaded weights");
    Net& n = e->net[which];
    TNet& w = e->train->net[which];
    if ((n.cfg.cond_channels > 0) != (cond_dev != nullptr)) return fail(e, DYF_ERR_INVALID_ARGUMENT, "condition must be given iff num_conditional_channels > 0");
    if (n.cfg.with_time_emb && !time_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "time must be given when with_time_emb");
    TK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    e->train->stream = st;
    TTape& t = e->train->tape[slot];
    TK(hipStreamSynchronize(st));
    tfree(e, t.owned);
    t = TTape{};
    t.net = which; t.nb = nb; t.flags = flags;
    const bool bn_batch = flags & DYF_TRAIN_BATCH_STATS, drop_on = (flags & DYF_TRAIN_DROPOUT) && n.cfg.dropout > 0.0f;
    const bool in_drop_on = (flags & DYF_TRAIN_DROPOUT) && n.cfg.input_dropout > 0.0f;
    const int H = e->cfg.height, W = e->cfg.width, hw = H * W, cin = n.cin_total, C = n.cfg.out_channels;
#define TS(expr) do { dyf_status _s = (expr); if (_s != DYF_OK) return _s; } while (0)
#define TA(ptr, count) TS(talloc(e, t.owned, &(ptr), (size_t)(count), false))
    if (drop_on || in_drop_on) {  // this forward's dropout streams (engine generator, keyed per global row); kept for the backward
        if (nb > 2 * e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch larger than the engine's row-key table");
        TK(launch_rng_begin_forward(e->rng_state, e->row_keys, nb, nb, st));
        TS(talloc(e, t.owned, &t.row_keys, (size_t)2 * nb, false));
        TK(hipMemcpyAsync(t.row_keys, e->row_keys, (size_t)2 * nb * sizeof(uint32_t), hipMemcpyDeviceToDevice, st));
    }
    // ---- time embedding chain: sinusoid -> Linear -> GELU -> Linear ; every block: SiLU -> Linear -> (scale | shift)
    if (n.cfg.with_time_emb) {
        TA(t.e0, nb * n.dim); TA(t.l1, nb * n.tdim); TA(t.gl, nb * n.tdim); TA(t.temb, nb * n.tdim);
        hipLaunchKernelGGL(t_sinusoid, dim3(nblk(nb * n.dim)), dim3(256), 0, st, time_dev, nb, n.dim, t.e0);
        hipLaunchKernelGGL(t_linear_fwd, dim3((unsigned)((n.tdim + 3) / 4), (unsigned)((nb + 15) / 16)), dim3(256), 0, st, t.e0, w.t_w1, w.t_b1, nb, n.dim, n.tdim, 0, t.l1);
        hipLaunchKernelGGL(t_gelu_fwd, dim3(nblk((long long)nb * n.tdim)), dim3(256), 0, st, t.l1, (long long)nb * n.tdim, t.gl);
        hipLaunchKernelGGL(t_linear_fwd, dim3((unsigned)((n.tdim + 3) / 4), (unsigned)((nb + 15) / 16)), dim3(256), 0, st, t.gl, w.t_w2, w.t_b2, nb, n.tdim, n.tdim, 0, t.temb);
    }
    // ---- stem: cat -> outer resample -> 1x1 conv
    TA(t.x_in, (size_t)nb * hw * cin);
    hipLaunchKernelGGL(t_nchw_cat_to_nhwc, dim3(nblk((long long)nb * hw * cin)), dim3(256), 0, st, inputs_dev, n.cfg.in_channels, cond_dev,
                       n.cfg.cond_channels, (const float*)nullptr, 0, nb, hw, t.x_in);
    if (n.uh != H || n.uw != W) {
        TA(t.x_up, (size_t)nb * n.uh * n.uw * cin);
        hipLaunchKernelGGL(t_resize_fwd, dim3(nblk((long long)nb * n.uh * n.uw * cin)), dim3(256), 0, st, t.x_in, nb, H, W, cin, n.uh, n.uw, n.cfg.outer_nearest, t.x_up);
    } else {
        t.x_up = t.x_in;
    }
    TA(t.s0, (size_t)nb * n.uh * n.uw * n.dim);
    TS(conv_fwd(e, TConv{nb, n.uh, n.uw, cin, n.uh, n.uw, n.dim, 1, 1, 0}, t.x_up, w.stem_wt, w.stem_b, t.s0, st));
    if (in_drop_on) {  // dropout_input (site DYF_INPUT_DROP_SITE), in place: the stem's conv needs its input only for the weight gradient
        const long long per = (long long)n.uh * n.uw * n.dim;
        hipLaunchKernelGGL(t_dropout_map, dim3(nblk(per * nb)), dim3(256), 0, st, t.s0, t.s0, nb, per, 1.0f / (1.0f - n.cfg.input_dropout),
                           keep_threshold16(n.cfg.input_dropout), rng_layer_salt(DYF_INPUT_DROP_SITE), t.row_keys);
    }
    double *S = nullptr, *Q = nullptr;
    TS(talloc(e, t.owned, &S, (size_t)nb * 1024 * 2));
    Q = S + (size_t)nb * 1024;
    // ---- the 12 blocks
    const float* x = t.s0;
    const float* x2 = nullptr;  // second part of a pending torch.cat([x, x2]) (channels xc | x2c): the x2 upsample reads both parts
    int xc = 0, x2c = 0;
    int lh = n.uh, lw = n.uw;
    for (int i = 0; i < 12; ++i) {
        const UBlock& b = n.blk[i];
        if (b.cout > 1024) return fail(e, DYF_ERR_UNSUPPORTED, "training path: more than 1024 channels per block");
        const bool up2 = b.transposed && b.cin % 4 == 0 && b.in_h == 2 * lh && b.in_w == 2 * lw;
        if (x2 && !(up2 && xc % 4 == 0 && x2c % 4 == 0)) {  // no consumer that reads two parts: materialise the concatenation
            float* cat = nullptr;
            TA(cat, (size_t)nb * lh * lw * (xc + x2c));
            launch_t_concat2(x, xc, x2, x2c, (long long)nb * lh * lw, cat, st);
            x = cat;
            x2 = nullptr;
        }
        const float* cx = x;
        if (b.transposed) {  // x2 bilinear upsample in front of the conv
            float* u = nullptr;
            TA(u, (size_t)nb * b.in_h * b.in_w * b.cin);
            if (up2)
                hipLaunchKernelGGL(t_up2x_fwd, dim3(nblk((long long)nb * b.in_h * b.in_w * (b.cin / 4))), dim3(256), 0, st, x, nb, lh, lw, b.cin / 4, u,
                                   x2 ? xc / 4 : b.cin / 4, x2);
            else
                hipLaunchKernelGGL(t_resize_fwd, dim3(nblk((long long)nb * b.in_h * b.in_w * b.cin)), dim3(256), 0, st, x, nb, lh, lw, b.cin, b.in_h, b.in_w, 0, u);
            cx = u;
            x2 = nullptr;
        }
        t.cin_ptr[i] = (float*)cx;
        const long long out_el = (long long)nb * b.out_h * b.out_w * b.cout;
        TA(t.z[i], out_el); TA(t.y[i], out_el);
        TS(conv_fwd(e, block_geom(b, nb), cx, w.blk[i].wt, w.blk[i].b, t.z[i], st));
        const int ohw = b.out_h * b.out_w, nidx = b.gn ? nb * 8 : b.cout;
        TA(t.mean[i], nidx); TA(t.rstd[i], nidx);
        const int kind = b.gn ? 2 : (bn_batch ? 0 : 1);
        if (kind != 1) {
            TK(hipMemsetAsync(S, 0, (size_t)nb * 1024 * 2 * sizeof(double), st));
            const int ppb = std::max(16, (ohw + 255) / 256);
            hipLaunchKernelGGL(t_nc_sums, dim3((ohw + ppb - 1) / ppb, nb), dim3(256), 0, st, t.z[i], ohw, b.cout, ppb, S, Q);
        }
        hipLaunchKernelGGL(t_stats_finalize, dim3(nblk(std::max(nidx, b.cout))), dim3(256), 0, st, kind, S, Q, nb, ohw, b.cout, 8, w.blk[i].rmean,
                           w.blk[i].rvar, t.mean[i], t.rstd[i]);
        if (n.cfg.with_time_emb) {
            TA(t.ss[i], (size_t)nb * 2 * b.cout);
            hipLaunchKernelGGL(t_linear_fwd, dim3((unsigned)((2 * b.cout + 3) / 4), (unsigned)((nb + 15) / 16)), dim3(256), 0, st, t.temb, w.blk[i].fw, w.blk[i].fb, nb, n.tdim, 2 * b.cout, 1, t.ss[i]);
        }
        TNorm a{nb, ohw, b.cout, 8, b.gn ? 1 : 0, b.act, t.mean[i], t.rstd[i], w.blk[i].gamma, w.blk[i].beta, t.ss[i], drop_on ? 1 : 0,
                1.0f / (1.0f - n.cfg.dropout), keep_threshold16(n.cfg.dropout), rng_layer_salt((uint32_t)i), t.row_keys};
        launch_t_norm_fwd(a, t.z[i], t.y[i], st);
        TK(hipGetLastError());
        x = t.y[i];
        lh = b.out_h; lw = b.out_w;
        if (i >= 6 && i < 11) {  // torch.cat([x, skip]) (unet_simple.py:176-177): left to the next block's upsample (two sources)
            x2 = t.y[10 - i];
            xc = b.cout;
            x2c = n.blk[10 - i].cout;
        }
    }
    t.xlast = (float*)x;
    // ---- readout: ConvTranspose2d (dgrad form of conv C) + final resample
    float *r = nullptr, *o = nullptr;
    TA(r, (size_t)nb * 4 * lh * lw * C);
    TS(conv_dgrad(e, TConv{nb, 2 * lh, 2 * lw, C, lh, lw, n.dim, 4, 2, 1}, x, w.ro_w, w.ro_b, r, st));
    TA(o, (size_t)nb * hw * C);
    hipLaunchKernelGGL(t_resize_fwd, dim3(nblk((long long)nb * hw * C)), dim3(256), 0, st, r, nb, 2 * lh, 2 * lw, C, H, W, n.cfg.outer_nearest, o);
    hipLaunchKernelGGL(t_nhwc_to_nchw, dim3(nblk((long long)nb * hw * C)), dim3(256), 0, st, o, nb, hw, C, 0, C, out_dev);
    TK(hipGetLastError());
#undef TA
#undef TS
    return DYF_OK;
}

dyf_status dyf_train_backward(dyf_engine* e, int32_t slot, const float* dout_dev, float* dinputs_dev, int32_t param_grads, void* stream) {
    if (!e || slot < 0 || slot > 3 || !dout_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_backward: bad arguments");
    const TrainPrecisionScope precision(e->train_precision);
    if (e->train && e->train->rtape[slot] && e->train->tape[slot].net < 0) {  // the slot holds a ResNet-UNet forward
        TK(hipSetDevice(e->cfg.device));
        return rn_train_backward(e, slot, dout_dev, dinputs_dev, param_grads, (hipStream_t)stream);
    }
    if (!e->train || e->train->tape[slot].net < 0) return fail(e, DYF_ERR_STATE, "no forward recorded in this tape slot");
    TK(hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    e->train->stream = st;
    TTape& t = e->train->tape[slot];
    Net& n = e->net[t.net];
    TNet& w = e->train->net[t.net];
    const int nb = t.nb, H = e->cfg.height, W = e->cfg.width, hw = H * W, cin = n.cin_total, C = n.cfg.out_channels;
    const bool bn_batch = t.flags & DYF_TRAIN_BATCH_STATS, drop_on = (t.flags & DYF_TRAIN_DROPOUT) && n.cfg.dropout > 0.0f;
    std::vector<void*> tmp;
#define TS(expr) do { dyf_status _s = (expr); if (_s != DYF_OK) { tfree(e, tmp); return _s; } } while (0)
#define TA(ptr, count) TS(talloc(e, tmp, &(ptr), (size_t)(count), false))
#define TZ(ptr, count) TS(talloc(e, tmp, &(ptr), (size_t)(count), true))
    const int lh = n.blk[11].out_h, lw = n.blk[11].out_w;
    // ---- final resample adjoint, readout
    float *d_o = nullptr, *d_r = nullptr, *dx = nullptr;
    TA(d_o, (size_t)nb * hw * C);
    hipLaunchKernelGGL(t_nchw_to_nhwc, dim3(nblk((long long)nb * hw * C)), dim3(256), 0, st, dout_dev, nb, hw, C, d_o);
    TZ(d_r, (size_t)nb * 4 * lh * lw * C);
    hipLaunchKernelGGL(t_resize_bwd, dim3(nblk((long long)nb * hw * C)), dim3(256), 0, st, d_o, nb, 2 * lh, 2 * lw, C, H, W, n.cfg.outer_nearest, d_r);
    const TConv gc{nb, 2 * lh, 2 * lw, C, lh, lw, n.dim, 4, 2, 1};
    if (param_grads) {
        TS(conv_wgrad(e, gc, t.xlast, d_r, w.g_ro_w, nullptr, st));
        launch_bias_grad(d_r, (long long)nb * 4 * lh * lw, C, w.g_ro_b, st);
    }
    TA(dx, (size_t)nb * lh * lw * n.dim);
    TS(conv_fwd(e, gc, d_r, w.ro_wt, nullptr, dx, st));
    // ---- blocks, last to first
    bool presplit = false;  // dy of the next (lower) block is already d y_i, its skip part already in dskip
    float* dskip[5] = {};
    for (int j = 0; j < 5; ++j) TZ(dskip[j], (size_t)nb * n.blk[j].out_h * n.blk[j].out_w * n.blk[j].cout);
    float* dsilu = nullptr;
    if (n.cfg.with_time_emb) TZ(dsilu, (size_t)nb * n.tdim);
    double* R = nullptr;
    TS(talloc(e, tmp, &R, (size_t)nb * 1024 * 4));
    float *S1 = nullptr, *S2 = nullptr, *dss = nullptr;
    TA(S1, std::max(1024, nb * 8)); TA(S2, std::max(1024, nb * 8)); TA(dss, (size_t)nb * 2 * 1024);
    float* dy = dx;  // gradient w.r.t. the output of block i (for i < 11: the y part of the concat)
    for (int i = 11; i >= 0; --i) {
        const UBlock& b = n.blk[i];
        const int ohw = b.out_h * b.out_w;
        const long long out_el = (long long)nb * ohw * b.cout;
        if (i >= 6 && i < 11 && presplit) {  // the upsample adjoint of block i + 1 already wrote the two parts
            presplit = false;
        } else if (i >= 6 && i < 11) {  // dy currently holds d cat[y_i, skip]: split
            const UBlock& sk = n.blk[10 - i];
            float* dyi = nullptr;
            TA(dyi, out_el);
            hipLaunchKernelGGL(t_split2, dim3(nblk((long long)nb * ohw * (b.cout + sk.cout))), dim3(256), 0, st, dy, b.cout, sk.cout, (long long)nb * ohw, dyi,
                               dskip[10 - i]);
            dy = dyi;
        } else if (i < 5) {  // encoder outputs also feed the decoder through the skips
            launch_t_add(dy, dskip[i], out_el, st);
        }
        TNorm a{nb, ohw, b.cout, 8, b.gn ? 1 : 0, b.act, t.mean[i], t.rstd[i], w.blk[i].gamma, w.blk[i].beta, t.ss[i], drop_on ? 1 : 0,
                1.0f / (1.0f - n.cfg.dropout), keep_threshold16(n.cfg.dropout), rng_layer_salt((uint32_t)i), t.row_keys};
        TK(hipMemsetAsync(R, 0, (size_t)nb * 1024 * 4 * sizeof(double), st));
        double *A = R, *B = R + (size_t)nb * 1024, *Cc = R + (size_t)2 * nb * 1024, *Dd = R + (size_t)3 * nb * 1024;
        const int ppb = std::max(16, (ohw + 255) / 256);
        hipLaunchKernelGGL(t_norm_bwd_sums, dim3((ohw + ppb - 1) / ppb, nb), dim3(256), 0, st, a, t.z[i], dy, ppb, A, B, Cc, Dd);
        const bool batch_stats = b.gn || bn_batch;
        hipLaunchKernelGGL(t_norm_bwd_combine, dim3(nblk(std::max(nb * b.cout, nb * 8))), dim3(256), 0, st, a, A, B, Cc, Dd,
                           param_grads ? w.blk[i].g_gamma : (float*)nullptr, param_grads ? w.blk[i].g_beta : (float*)nullptr,
                           n.cfg.with_time_emb ? dss : (float*)nullptr, S1, S2, batch_stats ? 1 : 0);
        float* dz = nullptr;
        TA(dz, out_el);
        const float inv_count = b.gn ? 1.0f / ((float)ohw * (b.cout / 8)) : 1.0f / ((float)nb * ohw);
        launch_t_norm_bwd_apply(a, t.z[i], dy, S1, S2, inv_count, dz, st);
        if (n.cfg.with_time_emb) {  // FiLM head: ss = W silu(temb) + b
            if (param_grads)
                hipLaunchKernelGGL(t_linear_bwd_w, dim3(nblk(2 * b.cout * n.tdim)), dim3(256), 0, st, t.temb, dss, nb, n.tdim, 2 * b.cout, 1, w.blk[i].g_fw,
                                   w.blk[i].g_fb);
            // d silu(temb) accumulated over the blocks (the SiLU derivative is applied once below)
            hipLaunchKernelGGL(t_linear_bwd_x, dim3((unsigned)(nb * ((n.tdim + 15) / 16))), dim3(256), 0, st, t.temb, w.blk[i].fw, dss, nb, n.tdim, 2 * b.cout, 0, 1, dsilu);
        }
        const TConv g = block_geom(b, nb);
        if (param_grads) TS(conv_wgrad(e, g, dz, t.cin_ptr[i], w.blk[i].g_w, w.blk[i].g_b, st));
        float* dcx = nullptr;  // gradient w.r.t. the conv input (block 0's feeds the stem's 1x1 conv)
        TA(dcx, (size_t)nb * b.in_h * b.in_w * b.cin);
        TS(conv_dgrad(e, g, dz, w.blk[i].w, nullptr, dcx, st));
        if (b.transposed) {  // adjoint of the x2 upsample
            const int ph = b.in_h / 2, pw = b.in_w / 2;
            float* dlow = nullptr;
            if (b.cin % 4 == 0 && b.in_h == 2 * ph && b.in_w == 2 * pw) {
                // the upsampled tensor of blocks 7..11 is cat[y_{i-1}, skip]: its gradient leaves as the two parts (no split pass)
                const int ca = i >= 7 ? n.blk[i - 1].cout : b.cin, cb = b.cin - ca;
                if (i >= 7 && ca % 4 == 0 && cb % 4 == 0 && cb == n.blk[11 - i].cout) {
                    TA(dlow, (size_t)nb * ph * pw * ca);
                    hipLaunchKernelGGL(t_up2x_bwd, dim3(nblk((long long)nb * ph * pw * (b.cin / 4))), dim3(256), 0, st, dcx, nb, ph, pw, b.cin / 4, dlow, ca / 4,
                                       dskip[11 - i]);
                    presplit = true;
                } else {
                    TA(dlow, (size_t)nb * ph * pw * b.cin);
                    hipLaunchKernelGGL(t_up2x_bwd, dim3(nblk((long long)nb * ph * pw * (b.cin / 4))), dim3(256), 0, st, dcx, nb, ph, pw, b.cin / 4, dlow, b.cin / 4,
                                       (float*)nullptr);
                }
            } else {
                TZ(dlow, (size_t)nb * ph * pw * b.cin);
                hipLaunchKernelGGL(t_resize_bwd, dim3(nblk((long long)nb * b.in_h * b.in_w * b.cin)), dim3(256), 0, st, dcx, nb, ph, pw, b.cin, b.in_h, b.in_w, 0, dlow);
            }
            dcx = dlow;
        }
        dy = dcx;  // gradient w.r.t. the previous tensor (block i-1's output, a concat for i in 7..11, the stem for i == 0)
        TK(hipGetLastError());
    }
    // ---- stem
    if ((t.flags & DYF_TRAIN_DROPOUT) && n.cfg.input_dropout > 0.0f) {  // adjoint of dropout_input: the same keep map on the gradient
        const long long per = (long long)n.uh * n.uw * n.dim;
        hipLaunchKernelGGL(t_dropout_map, dim3(nblk(per * nb)), dim3(256), 0, st, dy, (float*)dy, nb, per, 1.0f / (1.0f - n.cfg.input_dropout),
                           keep_threshold16(n.cfg.input_dropout), rng_layer_salt(DYF_INPUT_DROP_SITE), t.row_keys);
    }
    const TConv gs{nb, n.uh, n.uw, cin, n.uh, n.uw, n.dim, 1, 1, 0};
    if (param_grads) TS(conv_wgrad(e, gs, dy, t.x_up, w.g_stem_w, w.g_stem_b, st));
    if (dinputs_dev) {
        float* dxu = nullptr;
        TA(dxu, (size_t)nb * n.uh * n.uw * cin);
        TS(conv_dgrad(e, gs, dy, w.stem_w, nullptr, dxu, st));
        float* dxi = dxu;
        if (t.x_up != t.x_in) {
            TZ(dxi, (size_t)nb * hw * cin);
            hipLaunchKernelGGL(t_resize_bwd, dim3(nblk((long long)nb * n.uh * n.uw * cin)), dim3(256), 0, st, dxu, nb, H, W, cin, n.uh, n.uw, n.cfg.outer_nearest, dxi);
        }
        hipLaunchKernelGGL(t_nhwc_to_nchw, dim3(nblk((long long)nb * hw * n.cfg.in_channels)), dim3(256), 0, st, dxi, nb, hw, cin, 0, n.cfg.in_channels, dinputs_dev);
    }
    // ---- time MLP: dsilu = sum over blocks of dss . W_film ; temb -> SiLU is shared by all blocks
    if (n.cfg.with_time_emb && param_grads) {
        float *dtemb = nullptr, *dgl = nullptr;
        TA(dtemb, (size_t)nb * n.tdim); TA(dgl, (size_t)nb * n.tdim);
        hipLaunchKernelGGL(t_silu_bwd, dim3(nblk((long long)nb * n.tdim)), dim3(256), 0, st, t.temb, dsilu, (long long)nb * n.tdim, dtemb);
        hipLaunchKernelGGL(t_linear_bwd_w, dim3(nblk(n.tdim * n.tdim)), dim3(256), 0, st, t.gl, dtemb, nb, n.tdim, n.tdim, 0, w.g_t_w2, w.g_t_b2);
        hipLaunchKernelGGL(t_linear_bwd_x, dim3((unsigned)(nb * ((n.tdim + 15) / 16))), dim3(256), 0, st, t.gl, w.t_w2, dtemb, nb, n.tdim, n.tdim, 0, 0, dgl);
        hipLaunchKernelGGL(t_gelu_bwd, dim3(nblk((long long)nb * n.tdim)), dim3(256), 0, st, t.l1, (long long)nb * n.tdim, dgl);
        hipLaunchKernelGGL(t_linear_bwd_w, dim3(nblk(n.tdim * n.dim)), dim3(256), 0, st, t.e0, dgl, nb, n.dim, n.tdim, 0, w.g_t_w1, w.g_t_b1);
        TK(hipGetLastError());
    }
    TK(hipStreamSynchronize(st));
    tfree(e, tmp);
#undef TA
#undef TZ
#undef TS
    return DYF_OK;
}

// Copy gradients (and the updated BatchNorm running statistics) out, addressed by the reference's state_dict names, in
// PyTorch's layouts: conv weights (cout, cin, kh, kw), ConvTranspose2d (cin, cout, kh, kw), Linear (out, in).
// [co][tap][ci] -> (co, ci, tap), on the device
__global__ void t_unpack_conv(const float* g, int cout, int cin, int taps, float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)cout * cin * taps) return;
    const int tp = (int)(i % taps), ci = (int)((i / taps) % cin), co = (int)(i / ((long long)taps * cin));
    out[i] = g[((size_t)co * taps + tp) * cin + ci];
}
static dyf_status train_export_impl(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names, float* const* out_host, bool dev);
dyf_status dyf_train_export(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names, float* const* out_host) {
    return train_export_impl(e, which, n_tensors, names, out_host, false);
}
// the same into DEVICE buffers (contiguous fp32, on the engine's GPU): no host round trip
dyf_status dyf_train_export_dev(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names, float* const* out_dev) {
    return train_export_impl(e, which, n_tensors, names, out_dev, true);
}
static dyf_status train_export_impl(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names, float* const* out_host, bool dev) {
    if (!e || which < 0 || which > 1 || !names || !out_host) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_export: bad arguments");
    if (e->net[which].rn) { TK(hipSetDevice(e->cfg.device)); return rn_train_export(e, which, n_tensors, names, out_host, dev); }
    if (!e->train || !e->train->net[which].ready) return fail(e, DYF_ERR_STATE, "training needs arch unet_simple with loaded weights");
    TK(hipSetDevice(e->cfg.device));
    TK(hipDeviceSynchronize());
    const Net& n = e->net[which];
    const TNet& w = e->train->net[which];
    auto pull = [&](const float* dev, size_t cnt) {
        std::vector<float> h(cnt);
        (void)hipMemcpy(h.data(), dev, cnt * sizeof(float), hipMemcpyDeviceToHost);
        return h;
    };
    for (int q = 0; q < n_tensors; ++q) {
        const std::string name = names[q];
        float* out = out_host[q];
        bool done = false;
        auto flat = [&](const std::string& key, const float* dptr, size_t cnt) {
            if (!done && name == key && dptr) {
                if (dev) {
                    (void)hipMemcpy(out, dptr, cnt * sizeof(float), hipMemcpyDeviceToDevice);
                } else {
                    const std::vector<float> h = pull(dptr, cnt);
                    std::copy(h.begin(), h.end(), out);
                }
                done = true;
            }
        };
        for (int i = 0; i < 12 && !done; ++i) {
            const UBlock& b = n.blk[i];
            const TBlockW& bw = w.blk[i];
            const std::string pre = (i < 6 ? "input_ops." + std::to_string(i) : "output_ops." + std::to_string(i - 6));
            const std::string conv = pre + ".ops." + (b.transposed ? "1" : "0"), norm = pre + ".ops." + (b.transposed ? "2" : "1");
            if (name == conv + ".weight") {  // [co][tap][ci] -> (co, ci, kh, kw)
                const int taps = b.k * b.k;
                if (dev) {
                    hipLaunchKernelGGL(t_unpack_conv, dim3(nblk((long long)b.cout * taps * b.cin)), dim3(256), 0, nullptr, bw.g_w, b.cout, b.cin, taps, out);
                    done = true;
                    continue;
                }
                const std::vector<float> h = pull(bw.g_w, (size_t)b.cout * taps * b.cin);
                for (int co = 0; co < b.cout; ++co)
                    for (int ci = 0; ci < b.cin; ++ci)
                        for (int tp = 0; tp < taps; ++tp) out[((size_t)co * b.cin + ci) * taps + tp] = h[((size_t)co * taps + tp) * b.cin + ci];
                done = true;
            }
            flat(conv + ".bias", bw.g_b, b.cout);
            flat(norm + ".weight", bw.g_gamma, b.cout);
            flat(norm + ".bias", bw.g_beta, b.cout);
            flat(norm + ".running_mean", bw.rmean, b.cout);
            flat(norm + ".running_var", bw.rvar, b.cout);
            flat(pre + ".time_mlp.1.weight", bw.g_fw, (size_t)2 * b.cout * n.tdim);
            flat(pre + ".time_mlp.1.bias", bw.g_fb, (size_t)2 * b.cout);
        }
        flat("time_emb_mlp.1.weight", w.g_t_w1, (size_t)n.tdim * n.dim);
        flat("time_emb_mlp.1.bias", w.g_t_b1, n.tdim);
        flat("time_emb_mlp.3.weight", w.g_t_w2, (size_t)n.tdim * n.tdim);
        flat("time_emb_mlp.3.bias", w.g_t_b2, n.tdim);
        flat("init_conv.weight", w.g_stem_w, (size_t)n.dim * n.cin_total);
        flat("init_conv.bias", w.g_stem_b, n.dim);
        flat("readout.0.bias", w.g_ro_b, n.cfg.out_channels);
        if (!done && name == "readout.0.weight") {  // [co = dim][tap][ci = C] -> ConvTranspose2d (dim, C, 4, 4)
            const int oc = n.cfg.out_channels;
            if (dev) {
                hipLaunchKernelGGL(t_unpack_conv, dim3(nblk((long long)n.dim * 16 * oc)), dim3(256), 0, nullptr, w.g_ro_w, n.dim, oc, 16, out);
                continue;
            }
            const std::vector<float> h = pull(w.g_ro_w, (size_t)n.dim * 16 * oc);
            for (int co = 0; co < n.dim; ++co)
                for (int ci = 0; ci < oc; ++ci)
                    for (int tp = 0; tp < 16; ++tp) out[((size_t)co * oc + ci) * 16 + tp] = h[((size_t)co * 16 + tp) * oc + ci];
            done = true;
        }
        if (!done) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_export: unknown tensor '" + name + "'");
    }
    if (dev) TK(hipDeviceSynchronize());
    return DYF_OK;
}

// d(scale * mean-criterion(pred, target)) / d pred  (get_loss, src/utilities/utils.py:201-212; kinds as dyf_criterion)
dyf_status dyf_criterion_grad(dyf_engine* e, const float* pred_dev, const float* target_dev, int64_t count, int32_t kind, float scale,
                              float* dpred_dev, void* stream) {
    if (!e || !pred_dev || !target_dev || !dpred_dev || count < 1 || kind < 0 || kind > 2)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_criterion_grad: bad arguments");
    TK(hipSetDevice(e->cfg.device));
    hipLaunchKernelGGL(t_criterion_grad, dim3(nblk(count)), dim3(256), 0, (hipStream_t)stream, pred_dev, target_dev, (long long)count, kind, scale, dpred_dev);
    TK(hipGetLastError());
    return DYF_OK;
}

// [co][ci][taps] (PyTorch conv weight) -> w [co][tap][ci] and wt [tap][ci][co], on the device
__global__ void t_repack_conv(const float* raw, int cout, int cin, int taps, float* w, float* wt) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)cout * cin * taps) return;
    const int tp = (int)(i % taps), ci = (int)((i / taps) % cin), co = (int)(i / ((long long)taps * cin));
    const float v = raw[i];
    w[((size_t)co * taps + tp) * cin + ci] = v;
    wt[((size_t)tp * cin + ci) * cout + co] = v;
}

// In-place refresh of an existing training copy (same shapes): one H2D copy per tensor, the two conv layouts written by a kernel.
// Gradient buffers, tapes and every allocation stay as they are.
static dyf_status train_refresh_weights(dyf_engine* e, int which, std::map<std::string, TensorView>& sd, bool dev) {
    TNet& t = e->train->net[which];
    const Net& n = e->net[which];
    size_t stage_el = 0;
    for (int i = 0; i < 12; ++i) stage_el = std::max(stage_el, (size_t)n.blk[i].cout * n.blk[i].cin * n.blk[i].k * n.blk[i].k);
    stage_el = std::max(stage_el, (size_t)n.dim * 16 * n.cfg.out_channels);
    stage_el = std::max(stage_el, (size_t)n.dim * n.cin_total);
    std::vector<void*> tmp;
    float* stage = nullptr;
    dyf_status st0 = talloc(e, tmp, &stage, stage_el, false);
    if (st0 != DYF_OK) return st0;
    TK(hipDeviceSynchronize());
    auto put = [&](float* dst, const std::string& key, size_t want) -> bool {
        const TensorView& v = sd.at(key);
        if ((size_t)v.numel() != want) throw std::out_of_range("size of " + key);
        return hipMemcpy(dst, v.data, want * sizeof(float), dev ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice) == hipSuccess;
    };
    auto put_conv = [&](const std::string& key, int cout, int cin, int taps, float* w, float* wt) -> bool {
        const size_t el = (size_t)cout * cin * taps;
        if (dev) {  // the source is device memory: repack straight from it
            const TensorView& v = sd.at(key);
            if ((size_t)v.numel() != el) throw std::out_of_range("size of " + key);
            hipLaunchKernelGGL(t_repack_conv, dim3(nblk((long long)el)), dim3(256), 0, nullptr, v.data, cout, cin, taps, w, wt);
            return hipGetLastError() == hipSuccess;
        }
        if (!put(stage, key, el)) return false;
        hipLaunchKernelGGL(t_repack_conv, dim3(nblk((long long)el)), dim3(256), 0, nullptr, stage, cout, cin, taps, w, wt);
        return hipDeviceSynchronize() == hipSuccess;  // the staging buffer is overwritten by the next tensor
    };
    bool ok = true;
    for (int i = 0; i < 12 && ok; ++i) {
        const UBlock& b = n.blk[i];
        TBlockW& w = t.blk[i];
        const std::string pre = (i < 6 ? "input_ops." + std::to_string(i) : "output_ops." + std::to_string(i - 6));
        const std::string conv = pre + ".ops." + (b.transposed ? "1" : "0"), norm = pre + ".ops." + (b.transposed ? "2" : "1");
        ok = put_conv(conv + ".weight", b.cout, b.cin, b.k * b.k, w.w, w.wt) && put(w.b, conv + ".bias", b.cout) &&
             put(w.gamma, norm + ".weight", b.cout) && put(w.beta, norm + ".bias", b.cout);
        if (ok && !b.gn) ok = put(w.rmean, norm + ".running_mean", b.cout) && put(w.rvar, norm + ".running_var", b.cout);
        if (ok && n.cfg.with_time_emb)
            ok = put(w.fw, pre + ".time_mlp.1.weight", (size_t)2 * b.cout * n.tdim) && put(w.fb, pre + ".time_mlp.1.bias", (size_t)2 * b.cout);
    }
    if (ok && n.cfg.with_time_emb)
        ok = put(t.t_w1, "time_emb_mlp.1.weight", (size_t)n.tdim * n.dim) && put(t.t_b1, "time_emb_mlp.1.bias", n.tdim) &&
             put(t.t_w2, "time_emb_mlp.3.weight", (size_t)n.tdim * n.tdim) && put(t.t_b2, "time_emb_mlp.3.bias", n.tdim);
    if (ok) ok = put_conv("init_conv.weight", n.dim, n.cin_total, 1, t.stem_w, t.stem_wt) && put(t.stem_b, "init_conv.bias", n.dim);
    if (ok) ok = put_conv("readout.0.weight", n.dim, n.cfg.out_channels, 16, t.ro_w, t.ro_wt) && put(t.ro_b, "readout.0.bias", n.cfg.out_channels);
    if (dev) (void)hipDeviceSynchronize();
    tfree(e, tmp);
    if (!ok) return fail(e, DYF_ERR_HIP, "dyf_train_load_weights: upload failed");
    return DYF_OK;
}

// Refresh ONLY the training copy of a network's parameters (fp32, both conv layouts) -- what a training loop needs after every
// optimizer.step().  dyf_load_weights also rebuilds everything the sampling path derives from the weights (BatchNorm folding,
// phase-decomposed / fragment-ordered bf16 packs, FiLM tables: ~170 ms of host work for a unet_simple of dim 64); the sampling copy
// is left as it is and must be reloaded with dyf_load_weights before the network is sampled again (the Python module tracks both).
static dyf_status train_load_weights_impl(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                                          const float* const* data, const int64_t* const* shapes, const int32_t* ndims, bool dev);
dyf_status dyf_train_load_weights(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names, const float* const* data,
                                  const int64_t* const* shapes, const int32_t* ndims) {
    return train_load_weights_impl(e, which, n_tensors, names, data, shapes, ndims, false);
}
// the same with DEVICE pointers (contiguous fp32 tensors on the engine's GPU, e.g. the parameters of a module moved to the GPU)
dyf_status dyf_train_load_weights_dev(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                                      const float* const* data_dev, const int64_t* const* shapes, const int32_t* ndims) {
    return train_load_weights_impl(e, which, n_tensors, names, data_dev, shapes, ndims, true);
}
static dyf_status train_load_weights_impl(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                                          const float* const* data, const int64_t* const* shapes, const int32_t* ndims, bool dev) {
    if (!e || which < 0 || which > 1 || n_tensors < 1 || !names || !data || !shapes || !ndims)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_load_weights: bad arguments");
    TK(hipSetDevice(e->cfg.device));
    const Net& n = e->net[which];
    if (n.rn && n.loaded) {  // ResNet-UNet: the training copy is rebuilt (device tensors are staged through the host)
        std::map<std::string, TensorView> sd;
        std::vector<std::vector<float>> stage;
        if (dev) TK(hipDeviceSynchronize());
        for (int i = 0; i < n_tensors; ++i) {
            TensorView v;
            v.shape.assign(shapes[i], shapes[i] + ndims[i]);
            v.data = data[i];
            if (dev) {
                stage.emplace_back((size_t)v.numel());
                TK(hipMemcpy(stage.back().data(), data[i], stage.back().size() * sizeof(float), hipMemcpyDeviceToHost));
            }
            sd[names[i]] = v;
        }
        if (dev) {
            size_t k = 0;
            for (int i = 0; i < n_tensors; ++i) sd[names[i]].data = stage[k++].data();
        }
        return rn_train_store_weights(e, which, sd);
    }
    if (n.rn || n.sc || !n.loaded || !e->train || !e->train->net[which].ready)
        return fail(e, DYF_ERR_STATE, "dyf_train_load_weights: arch unet_simple / unet with weights loaded once by dyf_load_weights");
    std::map<std::string, TensorView> sd;
    for (int i = 0; i < n_tensors; ++i) {
        TensorView v;
        v.data = data[i];
        v.shape.assign(shapes[i], shapes[i] + ndims[i]);
        sd[names[i]] = v;
    }
    // same tensors and shapes as the copy in place (dyf_load_weights validated those)
    try {
        return train_refresh_weights(e, which, sd, dev);
    } catch (const std::exception& ex) {
        e->train->net[which].ready = false;
        return fail(e, DYF_ERR_INVALID_ARGUMENT, std::string("dyf_train_load_weights: state_dict does not match the loaded network: ") + ex.what());
    }
}

// Test seam (include/dyffusion_hip_testing.h): one training convolution on hash-random fp32 data through the fp32 matrix-core
// form (train_gemm.hip; with the split-K workspace, and -- forward / dgrad -- once more without it) against the plain VALU kernel.
// kind 0 forward, 1 dgrad, 2 wgrad.  out_host[0] = max |mfma - valu| / max |valu|, out_host[1] = the same for the unsplit launch,
// out_host[2] = 1 if the matrix-core form took the shape (0: it declined and nothing was compared).
__global__ void t_fill_hash(float* p, long long n, uint32_t seed) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint32_t h = fmix32((uint32_t)i * 0x9E3779B1u + seed);
    p[i] = ((float)(h >> 8) * (1.0f / 8388608.0f) - 1.0f);  // uniform in [-1, 1)
}
// in place: every value rounded to bf16, the training operand format (what the 16-bit-operand convs do while staging) -- on such data the
// fp32 reference kernels and the 16-bit matrix-core forms differ only by summation order
__global__ void t_scale_exp2(float* p, long long n, int k) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = ldexpf(p[i], k);
}
__global__ void t_round16(float* p, long long n) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = t16_to_f32(f32_to_t16(p[i]));  // bf16: the training operand format of both builds (train_internal.h)
}
__global__ void t_maxabs2(const float* a, const float* b, long long n, unsigned* out) {  // out[0] = max |a - b|, out[1] = max |b| (bit patterns)
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicMax(out, __float_as_uint(fabsf(a[i] - b[i])));
    atomicMax(out + 1, __float_as_uint(fabsf(b[i])));
    atomicMax(out + 2, __float_as_uint(fabsf(a[i])));
}

dyf_status dyf_train_conv_check(dyf_engine* e, int32_t kind, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout, int32_t k,
                                int32_t s, int32_t p, uint32_t seed, float* out_host) {
    if (!e || !out_host || kind < 0 || kind > 4 || n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1 || k < 1 || s < 1)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_train_conv_check: bad arguments");
    TK(hipSetDevice(e->cfg.device));
    if (!e->train) e->train = new TrainState();
    const int ho = (h + 2 * p - k) / s + 1, wo = (w + 2 * p - k) / s + 1;
    const TConv g{n, h, w, cin, ho, wo, cout, k, s, p};
    const long long nx = (long long)n * h * w * cin, nz = (long long)n * ho * wo * cout, nw = (long long)cout * k * k * cin;
    std::vector<void*> tmp;
    float *x = nullptr, *z = nullptr, *wgt = nullptr, *bias = nullptr, *ref = nullptr, *got = nullptr;
    unsigned* mx = nullptr;
    const long long nout = (kind == 0 || kind == 4) ? nz : kind == 1 ? nx : nw;
#define CK(expr) do { dyf_status _s = (expr); if (_s != DYF_OK) { tfree(e, tmp); return _s; } } while (0)
    CK(talloc(e, tmp, &x, (size_t)nx, false)); CK(talloc(e, tmp, &z, (size_t)nz, false)); CK(talloc(e, tmp, &wgt, (size_t)nw, false));
    CK(talloc(e, tmp, &bias, (size_t)std::max(cin, cout), false));
    CK(talloc(e, tmp, &ref, (size_t)nout, true)); CK(talloc(e, tmp, &got, (size_t)nout, true)); CK(talloc(e, tmp, &mx, (size_t)4, true));
    hipStream_t st = nullptr;
    hipLaunchKernelGGL(t_fill_hash, dim3(nblk(nx)), dim3(256), 0, st, x, nx, seed);
    hipLaunchKernelGGL(t_fill_hash, dim3(nblk(nz)), dim3(256), 0, st, z, nz, seed + 1u);
    hipLaunchKernelGGL(t_fill_hash, dim3(nblk(nw)), dim3(256), 0, st, wgt, nw, seed + 2u);   // used in BOTH weight layouts' index spaces
    hipLaunchKernelGGL(t_fill_hash, dim3(nblk(std::max(cin, cout))), dim3(256), 0, st, bias, (long long)std::max(cin, cout), seed + 3u);
    // test seam DYF_TRAIN_CHECK_DZ_EXP2 = k: the output gradient is scaled by 2^-k first -- the magnitudes a mean-reduced loss gives at
    // real batch sizes (1e-6 .. 1e-7) -- to show that the 16-bit gradient operand keeps its accuracy there (bf16 in both builds: an
    // fp16 operand would be subnormal or zero; a power of two commutes with the rounding, so the check's tolerance is unchanged)
    if (const char* k2 = dyf_form("DYF_TRAIN_CHECK_DZ_EXP2"))
        hipLaunchKernelGGL(t_scale_exp2, dim3(nblk(nz)), dim3(256), 0, st, z, nz, -atoi(k2));
    const TrainPrecisionScope precision(e->train_precision);
    if (train_operands16()) {
        hipLaunchKernelGGL(t_round16, dim3(nblk(nx)), dim3(256), 0, st, x, nx);
        hipLaunchKernelGGL(t_round16, dim3(nblk(nz)), dim3(256), 0, st, z, nz);
        hipLaunchKernelGGL(t_round16, dim3(nblk(nw)), dim3(256), 0, st, wgt, nw);
    }
    float res[3] = {0.0f, 0.0f, 0.0f};
    for (int pass = 0; pass < 2; ++pass) {  // pass 0: with the split-K workspace, pass 1: without
        float* ws = pass == 0 ? splitk_ws(e) : nullptr;
        bool took = false;
        TK(hipMemsetAsync(got, 0, (size_t)nout * sizeof(float), st));
        TK(hipMemsetAsync(ref, 0, (size_t)nout * sizeof(float), st));
        if (kind == 4) {  // whatever conv_fwd picks for the shape (the small-channel forms sit behind it)
            CK(conv_fwd(e, g, x, wgt, bias, got, st));
            took = true;
            hipLaunchKernelGGL(t_conv_fwd, dim3(nblk(nz)), dim3(256), 0, st, g, x, wgt, bias, ref);
        } else if (kind == 0) {
            took = tgemm_conv_fwd(g, x, wgt, bias, got, ws, TRAIN_SPLITK_FLOATS, st);
            hipLaunchKernelGGL(t_conv_fwd, dim3(nblk(nz)), dim3(256), 0, st, g, x, wgt, bias, ref);
        } else if (kind == 1) {
            took = tgemm_conv_dgrad(g, z, wgt, bias, got, ws, TRAIN_SPLITK_FLOATS, st);
            hipLaunchKernelGGL(t_conv_dgrad, dim3(nblk(nx)), dim3(256), 0, st, g, z, wgt, bias, ref);
        } else {
            // kind 3: whatever conv_wgrad picks for the shape (the small-channel forms sit behind it, not behind tgemm_conv_wgrad)
            if (kind == 3) { CK(conv_wgrad(e, g, z, x, got, nullptr, st)); took = true; }
            else took = tgemm_conv_wgrad(g, z, x, got, st);
            const long long M = (long long)n * ho * wo;
            const int tiles = k * k * ((cout + 15) / 16) * ((cin + 15) / 16);
            long long slices = std::max<long long>(1, std::min<long long>((M + 255) / 256, (4096 + tiles - 1) / tiles));
            const int ppb = (int)(((M + slices - 1) / slices + 15) / 16 * 16);
            slices = (M + ppb - 1) / ppb;
            hipLaunchKernelGGL(t_conv_wgrad, dim3((unsigned)(tiles * slices)), dim3(256), 0, st, g, z, x, ref, (float*)nullptr, ppb);
        }
        TK(hipGetLastError());
        res[2] = took ? 1.0f : 0.0f;
        if (took) {
            TK(hipMemsetAsync(mx, 0, 16, st));
            hipLaunchKernelGGL(t_maxabs2, dim3(nblk(nout)), dim3(256), 0, st, got, ref, nout, mx);
            unsigned hm[3];
            TK(hipMemcpy(hm, mx, sizeof(hm), hipMemcpyDeviceToHost));
            float d, m, ga;
            memcpy(&d, &hm[0], 4); memcpy(&m, &hm[1], 4); memcpy(&ga, &hm[2], 4);
            res[pass] = m > 0.0f ? d / m : 1.0f;  // an all-zero reference is a failed check, not a perfect one
            if (!(ga > 0.0f)) res[pass] = 1.0f;
            if (dyf_form("DYF_TRAIN_CHECK_VERBOSE")) fprintf(stderr, "conv_check kind %d pass %d: max|diff| %g max|ref| %g max|got| %g\n", kind, pass, d, m, ga);
        }
        if (kind >= 2) { res[1] = res[0]; break; }
    }
#undef CK
    TK(hipDeviceSynchronize());
    tfree(e, tmp);
    out_host[0] = res[0]; out_host[1] = res[1]; out_host[2] = res[2];
    return DYF_OK;
}

}  // extern "C"
