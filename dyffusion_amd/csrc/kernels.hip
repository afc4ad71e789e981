// HBM-bound kernels of the DYffusion engine (gfx950): stem, x2 bilinear upsample, GroupNorm, readout, time/FiLM heads,
// sampler elementwise.  Every kernel replaces an unfused ATen sequence of the reference (cited per kernel).
#include "kernels.h"

#include <cstdlib>

// ------------------------------------------------------------------------------------------------ block reduce
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

__device__ __forceinline__ float block_sum(float v, float* scratch /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) scratch[wave] = v;
    __syncthreads();
    float t = 0.0f;
    for (int i = 0; i < nw; ++i) t += scratch[i];
    return t;
}

// ------------------------------------------------------------------------------------------------ K9 time MLP
// misc.py:20-32 (SinusoidalPosEmb), :63-66 (Linear -> GELU(erf) -> Linear); the trailing SiLU is the first op of
// every UNetBlock.time_mlp (unet_simple.py:21-23) and is shared by all blocks.
__global__ void time_mlp_kernel(TimeMlpArgs a) {
    extern __shared__ float sh[];  // e[dim] | h[tdim]
    const int row = blockIdx.x, tdim = 2 * a.dim, half = a.dim / 2;
    const int dim = a.learned_w ? 2 * a.learned_half + 1 : a.dim;  // number of time features
    float* e = sh;
    float* h = sh + dim;
    const float t = a.time[row];
    if (a.learned_w) {
        for (int i = threadIdx.x; i < dim; i += blockDim.x) {
            if (i == 0) { e[0] = t; continue; }
            const int j = (i - 1) % a.learned_half;
            const float ang = t * a.learned_w[j] * 6.283185307179586f;
            e[i] = i <= a.learned_half ? sinf(ang) : cosf(ang);
        }
    } else {
        const float step = -logf(10000.0f) / (float)(half - 1);
        for (int i = threadIdx.x; i < dim; i += blockDim.x) {
            const int j = i < half ? i : i - half;
            const float ang = t * expf((float)j * step);
            e[i] = i < half ? sinf(ang) : cosf(ang);
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tdim; j += blockDim.x) {
        float acc = a.b1[j];
        const float* w = a.w1 + (size_t)j * dim;
        for (int i = 0; i < dim; ++i) acc = fmaf(w[i], e[i], acc);
        h[j] = 0.5f * acc * (1.0f + erff(acc * 0.70710678118654752f));
    }
    __syncthreads();
    for (int j = threadIdx.x; j < tdim; j += blockDim.x) {
        float acc = a.b2[j];
        const float* w = a.w2 + (size_t)j * tdim;
        for (int i = 0; i < tdim; ++i) acc = fmaf(w[i], h[i], acc);
        a.silu_out[(size_t)row * tdim + j] = acc / (1.0f + expf(-acc));
    }
}

hipError_t launch_time_mlp(const TimeMlpArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(time_mlp_kernel, dim3(a.rows), dim3(128), (size_t)(3 * a.dim + 2 * a.learned_half + 1) * sizeof(float), s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K9 FiLM heads
// unet_simple.py:72-78: time_mlp Linear -> chunk(scale, shift) -> x*(scale+1)+shift, folded with the block's
// eval-mode BatchNorm (and conv bias): y = conv*A + C.
__global__ void film_kernel(FilmArgs a) {
    const int fc = blockIdx.x * blockDim.x + threadIdx.x;
    const int row = blockIdx.y;
    if (fc >= a.total_c) return;
    float scale = 0.0f, shift = 0.0f;
    if (a.silu != nullptr) {
        const int b = a.blk_of[fc];
        const int cl = fc - a.blk_off[b];
        const int r_scale = 2 * a.blk_off[b] + cl;
        const int r_shift = r_scale + a.blk_cout[b];
        const float* sv = a.silu + (size_t)row * a.tdim;
        const float* ws = a.wf + (size_t)r_scale * a.tdim;
        const float* wh = a.wf + (size_t)r_shift * a.tdim;
        scale = a.bf[r_scale];
        shift = a.bf[r_shift];
        for (int i = 0; i < a.tdim; ++i) {
            scale = fmaf(ws[i], sv[i], scale);
            shift = fmaf(wh[i], sv[i], shift);
        }
    }
    const float na = a.norm_a[fc], nc = a.norm_c[fc];
    a.coef_a[(size_t)row * a.total_c + fc] = na * (1.0f + scale);
    a.coef_c[(size_t)row * a.total_c + fc] = fmaf(nc, 1.0f + scale, shift);
}

hipError_t launch_film(const FilmArgs& a, hipStream_t s) {
    hipLaunchKernelGGL(film_kernel, dim3((a.total_c + 127) / 128, a.rows), dim3(128), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K4+K1 stem
// unet_simple.py:184-195: cat[inputs, condition] -> Upsample(size, bilinear) -> init_conv 1x1.  One thread per
// resampled pixel; the sampled channel vector is parked in LDS so the 1x1 can index it dynamically.
__global__ __launch_bounds__(256) void stem_kernel(StemArgs a) {
    extern __shared__ float vals[];  // [cin][256]
    const long long pix = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.n * a.uh * a.uw;
    const bool live = pix < total;
    const int tid = threadIdx.x;
    if (live) {
        const int n = (int)(pix / ((long long)a.uh * a.uw));
        const int rem = (int)(pix % ((long long)a.uh * a.uw));
        const int y = rem / a.uw, x = rem % a.uw;
        int y0 = y, y1 = y, x0 = x, x1 = x;
        float ly = 0.0f, lx = 0.0f;
        if (a.resample) {
            bilinear_coord(y, (float)a.h / (float)a.uh, a.h, y0, y1, ly, a.nearest != 0);
            bilinear_coord(x, (float)a.w / (float)a.uw, a.w, x0, x1, lx, a.nearest != 0);
        }
        int cbase = 0;
        for (int s = 0; s < a.nsrc; ++s) {
            const float* src = a.src[s] + (size_t)(a.src_rows > 0 ? n % a.src_rows : n) * a.ch[s] * a.h * a.w;
            for (int c = 0; c < a.ch[s]; ++c) {
                const float* p = src + (size_t)c * a.h * a.w;
                const float v00 = p[y0 * a.w + x0], v01 = p[y0 * a.w + x1];
                const float v10 = p[y1 * a.w + x0], v11 = p[y1 * a.w + x1];
                // same association as ATen's CPU kernel: rows first, then columns
                const float top = v00 * (1.0f - lx) + v01 * lx;
                const float bot = v10 * (1.0f - lx) + v11 * lx;
                vals[(cbase + c) * 256 + tid] = top * (1.0f - ly) + bot * ly;
            }
            cbase += a.ch[s];
        }
    }
    __syncthreads();
    if (!live) return;
    el16_t* out = a.out + (size_t)pix * a.dim;
    const int n_row = (int)(pix / ((long long)a.uh * a.uw));
    const RngKey dkey = drop_row_key(a.drop, n_row);
    const uint32_t row0 = (uint32_t)n_row * (uint32_t)(a.uh * a.uw * a.dim);
    for (int d0 = 0; d0 < a.dim; d0 += 8) {
        float acc[8];
#pragma unroll
        for (int t = 0; t < 8; ++t) acc[t] = (d0 + t < a.dim) ? a.bias[d0 + t] : 0.0f;
        for (int c = 0; c < a.cin; ++c) {
            const float v = vals[c * 256 + tid];
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (d0 + t < a.dim) acc[t] = fmaf(a.wgt[(size_t)(d0 + t) * a.cin + c], v, acc[t]);
        }
        if (a.drop.mode != 0) {  // dropout_input (unet_simple.py:168)
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (d0 + t < a.dim) acc[t] = drop_apply(acc[t], (uint32_t)(pix * a.dim + d0 + t), row0, a.drop, dkey);
        }
        if ((a.dim & 7) == 0) {
            uint4 o;
            o.x = pack_el16x2(acc[0], acc[1]);
            o.y = pack_el16x2(acc[2], acc[3]);
            o.z = pack_el16x2(acc[4], acc[5]);
            o.w = pack_el16x2(acc[6], acc[7]);
            *(uint4*)(out + d0) = o;
        } else {
#pragma unroll
            for (int t = 0; t < 8; ++t)
                if (d0 + t < a.dim) out[d0 + t] = f32_to_el16(acc[t]);
        }
    }
}

hipError_t launch_stem(const StemArgs& a, hipStream_t s) {
    const long long total = (long long)a.n * a.uh * a.uw;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)a.cin * 256 * sizeof(float),
                       s, a);
    return hipGetLastError();
}

// rng_begin_forward_kernel's work done by block 0 of a stem launch (StemArgs::rng_state != null): one launch (and one kernel
// boundary) less per forward that draws masks -- 6-7 us of a 250 us one-row forward.  Same arithmetic, same table.
__device__ __forceinline__ void stem_rng_begin(const StemArgs& a) {
    if (a.rng_state == nullptr || blockIdx.x != 0) return;
    const uint32_t lo = a.rng_state[0], hi = a.rng_state[1], fwd0 = a.rng_state[2], row0 = a.rng_state[3];
    for (int r = threadIdx.x; r < a.rng_rows; r += blockDim.x) {
        const RngKey k = rng_row_key(lo, hi, fwd0 + (uint32_t)(r / a.rng_rows_per_fwd), row0 + (uint32_t)(r % a.rng_rows_per_fwd));
        a.rng_row_keys[2 * r] = k.k0;
        a.rng_row_keys[2 * r + 1] = k.k1;
    }
    __syncthreads();  // (every thread of block 0 has read fwd0)
    if (threadIdx.x == 0) a.rng_state[2] = fwd0 + (uint32_t)((a.rng_rows + a.rng_rows_per_fwd - 1) / a.rng_rows_per_fwd);
}

// init_conv is linear and nothing non-linear sits between it and the first encoder conv (dropout_input has p = 0),
// so conv4x4(init_conv(x)) is ONE 4x4 conv on the resampled raw channels.  This kernel only resamples.
__global__ __launch_bounds__(256) void stem16_kernel(StemArgs a) {
    stem_rng_begin(a);
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const int ph = a.uh + 2, pw = a.uw + 2;
    const long long total = (long long)a.n * ph * pw;
    if (idx >= total) return;
    const int n = (int)(idx / ((long long)ph * pw));
    const int rem = (int)(idx % ((long long)ph * pw));
    const int y = rem / pw - 1, x = rem % pw - 1;
    uint32_t w[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    if ((unsigned)y < (unsigned)a.uh && (unsigned)x < (unsigned)a.uw) {
        int y0 = y, y1 = y, x0 = x, x1 = x;
        float ly = 0.0f, lx = 0.0f;
        if (a.resample) {
            bilinear_coord(y, (float)a.h / (float)a.uh, a.h, y0, y1, ly, a.nearest != 0);
            bilinear_coord(x, (float)a.w / (float)a.uw, a.w, x0, x1, lx, a.nearest != 0);
        }
        // channel k of the concatenation lives in source s(k): a static loop over the 16 output slots (the dynamic
        // (source, channel) walk of the first version cost a 16-way select chain per channel)
        const int e0 = a.ch[0], e1 = e0 + (a.nsrc > 1 ? a.ch[1] : 0), e2 = e1 + (a.nsrc > 2 ? a.ch[2] : 0);
        const int nrow = a.src_rows > 0 ? n % a.src_rows : n;
        const int o00 = y0 * a.w + x0, o01 = y0 * a.w + x1, o10 = y1 * a.w + x0, o11 = y1 * a.w + x1;
        const size_t plane = (size_t)a.h * a.w;
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            float val = 0.0f;
            if (k < a.cin) {
                const int si = k < e0 ? 0 : (k < e1 ? 1 : (k < e2 ? 2 : 3));
                const int cl = k - (si == 0 ? 0 : (si == 1 ? e0 : (si == 2 ? e1 : e2)));
                const float* p = a.src[si] + ((size_t)nrow * a.ch[si] + cl) * plane;
                const float top = p[o00] * (1.0f - lx) + p[o01] * lx;
                const float bot = p[o10] * (1.0f - lx) + p[o11] * lx;
                val = top * (1.0f - ly) + bot * ly;
            } else if (k == a.cin) {
                val = 1.0f;  // carries init_conv's bias through the composed enc0 weights
            }
            v[k] = val;
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) w[k] = pack_el16x2(v[2 * k], v[2 * k + 1]);
    }
    uint4* o = (uint4*)(a.out + (size_t)idx * 16);
    o[0] = make_uint4(w[0], w[1], w[2], w[3]);
    o[1] = make_uint4(w[4], w[5], w[6], w[7]);
}

// Row-block form of stem16_kernel: one workgroup per (sample, block of ST_ROWS padded output rows).  The source rows the
// block touches (<= ST_ROWS + 2 when the grid is upsampled: all channels, native width) are staged in LDS as [row][x][channel]
// once, with loads that are contiguous along x; every lane then reads its four taps as 16-byte LDS rows and two lanes share a
// pixel (8 channels = 16 bytes each), so a wave's store instruction writes 1 KB of consecutive bytes.  stem16_kernel issues 4
// scattered 4-byte global loads per channel and pixel (32 per thread for the NS inputs).  The expressions are the same (x
// first, then y); the compiler contracts them into FMAs differently, so the two forms agree to a 16-bit rounding tie, not bit
// for bit.  (One ROW per workgroup was slower than stem16_kernel: 41 280 workgroups each waiting for its own staging round trip.)
constexpr int ST_ROWS = 8;
template <int CP>  // channels padded to a multiple of 4 (LDS row of a source pixel)
__global__ __launch_bounds__(512) void stem16_rows_kernel(StemArgs a, int nblk) {
    extern __shared__ __attribute__((aligned(16))) float st_rows[];  // [ST_ROWS + 2][w][CP]
    stem_rng_begin(a);
    const int ph = a.uh + 2, pw = a.uw + 2;
    const int n = blockIdx.x / nblk, rb = blockIdx.x - n * nblk;
    const int py0 = rb * ST_ROWS, py1 = min(py0 + ST_ROWS, ph);           // padded rows of this block
    const int ya = max(py0 - 1, 0), yb = min(py1 - 2, a.uh - 1);          // image rows among them (ya > yb: border only)
    const float ys = (float)a.h / (float)a.uh, xs = (float)a.w / (float)a.uw;
    int sbase = 0;
    if (ya <= yb) {
        int s0 = ya, s1 = yb, t0, t1;
        float l;
        if (a.resample) {
            bilinear_coord(ya, ys, a.h, s0, t0, l, a.nearest != 0);
            bilinear_coord(yb, ys, a.h, t1, s1, l, a.nearest != 0);
        }
        sbase = s0;
        const int nsrc_rows = s1 - s0 + 1;  // <= ST_ROWS + 2 (host: h <= uh)
        const int nrow = a.src_rows > 0 ? n % a.src_rows : n;
        const size_t plane = (size_t)a.h * a.w;
        const int per_row = a.cin * a.w;
        for (int j = threadIdx.x; j < per_row; j += 512) {  // a thread keeps its (channel, x) and walks the rows: no divisions inside
            const int k = j / a.w, x = j - k * a.w;
            int si = 0, cl = k;
            while (cl >= a.ch[si]) cl -= a.ch[si++];
            const float* sp = a.src[si] + ((size_t)nrow * a.ch[si] + cl) * plane + (size_t)s0 * a.w + x;
            float* dp = st_rows + (size_t)x * CP + k;
            // all row loads in flight together (a rolled loop waited for each load before its LDS store: ~10 serial round trips
            // were the lifetime of a workgroup)
            float tmp[ST_ROWS + 2];
#pragma unroll
            for (int r = 0; r < ST_ROWS + 2; ++r) tmp[r] = r < nsrc_rows ? sp[(size_t)r * a.w] : 0.0f;
#pragma unroll
            for (int r = 0; r < ST_ROWS + 2; ++r)
                if (r < nsrc_rows) dp[(size_t)r * a.w * CP] = tmp[r];
        }
    }
    // per-column (x0, x1, lx) and per-row (y0, y1, ly) stencils, once per workgroup: the item loop below is instruction-bound
    // (the two source-coordinate evaluations and an integer division per 16 bytes were half of it)
    int4* xtab = (int4*)(st_rows + (size_t)(ST_ROWS + 2) * a.w * CP);   // [pw]: x0 * CP, x1 * CP, lx; .w = valid
    int4* ytab = xtab + pw;                                             // [ST_ROWS]: row offsets into st_rows, ly; .w = valid
    for (int px = threadIdx.x; px < pw + ST_ROWS; px += 512) {
        if (px < pw) {
            const int x = px - 1;
            int x0 = x, x1 = x;
            float lx = 0.0f;
            const bool ok = (unsigned)x < (unsigned)a.uw;
            if (ok && a.resample) bilinear_coord(x, xs, a.w, x0, x1, lx, a.nearest != 0);
            xtab[px] = make_int4(ok ? x0 * CP : 0, ok ? x1 * CP : 0, __float_as_int(lx), ok);
        } else {
            const int r = px - pw, y = py0 + r - 1;
            int y0 = y, y1 = y;
            float ly = 0.0f;
            const bool ok = (unsigned)y < (unsigned)a.uh && py0 + r < py1;
            if (ok && a.resample) bilinear_coord(y, ys, a.h, y0, y1, ly, a.nearest != 0);
            ytab[r] = make_int4(ok ? (y0 - sbase) * a.w * CP : 0, ok ? (y1 - sbase) * a.w * CP : 0, __float_as_int(ly), ok);
        }
    }
    __syncthreads();
    uint4* oblk = (uint4*)(a.out + (((size_t)n * ph + py0) * pw) * 16);
    if (a.cin <= 8 && (a.uw & 63) == 0) {
        // At most 8 input channels (the NS nets: 8): channels 8-15 of a pixel are constants (the bias slot, zeros), so only ONE lane per
        // pixel does arithmetic -- the loop is instruction-bound -- and the pixel's two 16-byte halves pass through a per-wave
        // LDS tile so that a wave still stores 2 x 1 KB of consecutive bytes for its 64 consecutive pixels.
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        unsigned char* stg = (unsigned char*)(ytab + ST_ROWS) + wave * 2048;  // [64 pixels][32 B]
        const uint4 zero4 = make_uint4(0, 0, 0, 0);
        const uint4 half1 = a.cin == 8 ? make_uint4(pack_el16x2(1.0f, 0.0f), 0, 0, 0) : zero4;  // slot cin = 1 (init_conv's bias)
        for (int i = threadIdx.x; i < (py1 - py0) * 4; i += 512) {  // the two zero border pixels of every row
            const int rr = i >> 2, q = i & 3;
            oblk[(size_t)rr * 2 * pw + (q < 2 ? q : 2 * pw - 4 + q)] = zero4;
        }
        const int total_in = (py1 - py0) * a.uw;
        int r = 0, j = threadIdx.x;  // interior pixel (row r of the block, column j)
        while (j >= a.uw) { j -= a.uw; ++r; }
        for (int i = threadIdx.x; i < total_in; i += 512) {
            const int4 xt = xtab[j + 1], yt = ytab[r];
            uint4 w = zero4;
            if (yt.w) {
                const int x0 = xt.x, x1 = xt.y;
                const float lx = __int_as_float(xt.z), ly = __int_as_float(yt.z);
                const float* r0 = st_rows + yt.x;
                const float* r1 = st_rows + yt.y;
                float v[8];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    float t00[4] = {0, 0, 0, 0}, t01[4] = {0, 0, 0, 0}, t10[4] = {0, 0, 0, 0}, t11[4] = {0, 0, 0, 0};
                    if (4 * q < CP) {
                        const float4 p00 = *(const float4*)(r0 + x0 + 4 * q), p01 = *(const float4*)(r0 + x1 + 4 * q);
                        const float4 p10 = *(const float4*)(r1 + x0 + 4 * q), p11 = *(const float4*)(r1 + x1 + 4 * q);
                        t00[0] = p00.x; t00[1] = p00.y; t00[2] = p00.z; t00[3] = p00.w;
                        t01[0] = p01.x; t01[1] = p01.y; t01[2] = p01.z; t01[3] = p01.w;
                        t10[0] = p10.x; t10[1] = p10.y; t10[2] = p10.z; t10[3] = p10.w;
                        t11[0] = p11.x; t11[1] = p11.y; t11[2] = p11.z; t11[3] = p11.w;
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int k = 4 * q + e;
                        const float top = t00[e] * (1.0f - lx) + t01[e] * lx;
                        const float bot = t10[e] * (1.0f - lx) + t11[e] * lx;
                        float val = top * (1.0f - ly) + bot * ly;
                        if (k >= a.cin) val = k == a.cin ? 1.0f : 0.0f;
                        v[k] = val;
                    }
                }
                w = make_uint4(pack_el16x2(v[0], v[1]), pack_el16x2(v[2], v[3]), pack_el16x2(v[4], v[5]), pack_el16x2(v[6], v[7]));
            }
            *(uint4*)(stg + lane * 32) = w;
            *(uint4*)(stg + lane * 32 + 16) = yt.w ? half1 : zero4;
            uint4* dst = oblk + (size_t)r * 2 * pw + 2 * (1 + j - lane) + lane;  // the wave's 64 pixels: 128 consecutive 16-byte chunks
            dst[0] = *(const uint4*)(stg + lane * 16);
            dst[64] = *(const uint4*)(stg + 1024 + lane * 16);
            j += 512;
            while (j >= a.uw) { j -= a.uw; ++r; }
        }
        return;
    }
    const int total = (py1 - py0) * 2 * pw;
    int r = 0, j = threadIdx.x;  // item i = r * 2 pw + j, advanced without a division (512 <= 2 pw is checked by the launcher)
    for (int i = threadIdx.x; i < total; i += 512) {
        const int px = j >> 1, half = j & 1;
        const int4 xt = xtab[px], yt = ytab[r];
        uint32_t w[4] = {0, 0, 0, 0};
        if (xt.w && yt.w) {
            const int x0 = xt.x, x1 = xt.y;
            const float lx = __int_as_float(xt.z), ly = __int_as_float(yt.z);
            const float* r0 = st_rows + yt.x;
            const float* r1 = st_rows + yt.y;
            float v[8];
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
                const int q = 2 * half + q2;  // 4-channel group of the concatenation
                float t00[4] = {0, 0, 0, 0}, t01[4] = {0, 0, 0, 0}, t10[4] = {0, 0, 0, 0}, t11[4] = {0, 0, 0, 0};
                if (4 * q < CP) {
                    const float4 p00 = *(const float4*)(r0 + x0 + 4 * q), p01 = *(const float4*)(r0 + x1 + 4 * q);
                    const float4 p10 = *(const float4*)(r1 + x0 + 4 * q), p11 = *(const float4*)(r1 + x1 + 4 * q);
                    t00[0] = p00.x; t00[1] = p00.y; t00[2] = p00.z; t00[3] = p00.w;
                    t01[0] = p01.x; t01[1] = p01.y; t01[2] = p01.z; t01[3] = p01.w;
                    t10[0] = p10.x; t10[1] = p10.y; t10[2] = p10.z; t10[3] = p10.w;
                    t11[0] = p11.x; t11[1] = p11.y; t11[2] = p11.z; t11[3] = p11.w;
                }
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = 4 * q + e;
                    const float top = t00[e] * (1.0f - lx) + t01[e] * lx;
                    const float bot = t10[e] * (1.0f - lx) + t11[e] * lx;
                    float val = top * (1.0f - ly) + bot * ly;
                    if (k >= a.cin) val = k == a.cin ? 1.0f : 0.0f;  // slot cin carries init_conv's bias through the composed enc0 weights
                    v[4 * q2 + e] = val;
                }
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = pack_el16x2(v[2 * k], v[2 * k + 1]);
        }
        oblk[i] = make_uint4(w[0], w[1], w[2], w[3]);
        j += 512;
        if (j >= 2 * pw) { j -= 2 * pw; ++r; }
    }
}

hipError_t launch_stem16(const StemArgs& a, hipStream_t s) {
    const int cp = (a.cin + 3) / 4 * 4;  // cin <= 15 (the fused stem's bias channel is slot cin)
    const size_t lds = (size_t)(ST_ROWS + 2) * a.w * cp * sizeof(float) + (size_t)(a.uw + 2 + ST_ROWS) * 16 + 8 * 2048;  // rows + stencil tables + per-wave output tiles
    const char* env = dyf_form("DYF_STEM16_ROWS");  // read per launch: the parity test flips it
    if (!(env && atoi(env) == 0) && lds <= 48 * 1024 && cp <= 16 && a.h <= a.uh && 2 * (a.uw + 2) >= 512) {
        const int nblk = (a.uh + 2 + ST_ROWS - 1) / ST_ROWS;
        const dim3 grid((unsigned)(a.n * nblk)), block(512);
        dyf_form_note("stem16_rows_kernel", a.n);
        KernelProf kp("stem16_rows_kernel", s, (double)(a.src_rows ? a.src_rows : a.n) * a.h * a.w * a.cin * 4.0 + (double)a.n * (a.uh + 2) * (a.uw + 2) * 16 * 2.0);
        if (cp == 4) hipLaunchKernelGGL(stem16_rows_kernel<4>, grid, block, lds, s, a, nblk);
        else if (cp == 8) hipLaunchKernelGGL(stem16_rows_kernel<8>, grid, block, lds, s, a, nblk);
        else if (cp == 12) hipLaunchKernelGGL(stem16_rows_kernel<12>, grid, block, lds, s, a, nblk);
        else hipLaunchKernelGGL(stem16_rows_kernel<16>, grid, block, lds, s, a, nblk);
        return hipGetLastError();
    }
    const long long total = (long long)a.n * (a.uh + 2) * (a.uw + 2);
    dyf_form_note("stem16_kernel", a.n);
    hipLaunchKernelGGL(stem16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K2 x2 upsample
// unet_simple.py:42 nn.Upsample(scale_factor=2, bilinear) applied to cat[x, skip] (:176-177).
template <int VEC>
__global__ __launch_bounds__(256) void up2x_kernel(Up2xArgs a, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = a.c0 + a.c1;
    const int groups = c / VEC;
    const int g = (int)(idx % groups);
    const long long pix = idx / groups;
    const int ow = 2 * a.w, oh = 2 * a.h;
    const int n = (int)(pix / ((long long)oh * ow));
    const int rem = (int)(pix % ((long long)oh * ow));
    const int y = rem / ow, x = rem % ow;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_coord(y, 0.5f, a.h, y0, y1, ly);
    bilinear_coord(x, 0.5f, a.w, x0, x1, lx);
    int ch = g * VEC;
    const el16_t* src = a.src0;
    int cs = a.c0;
    if (ch >= a.c0) {
        src = a.src1;
        cs = a.c1;
        ch -= a.c0;
    }
    const size_t base = (size_t)n * a.h * a.w;
    const el16_t* p00 = src + ((base + (size_t)y0 * a.w + x0) * cs + ch);
    const el16_t* p01 = src + ((base + (size_t)y0 * a.w + x1) * cs + ch);
    const el16_t* p10 = src + ((base + (size_t)y1 * a.w + x0) * cs + ch);
    const el16_t* p11 = src + ((base + (size_t)y1 * a.w + x1) * cs + ch);
    el16_t* o = a.out + ((size_t)pix * c + g * VEC);
    if (VEC == 8) {
        const uint4 q00 = *(const uint4*)p00, q01 = *(const uint4*)p01, q10 = *(const uint4*)p10, q11 = *(const uint4*)p11;
        const uint32_t* w00 = (const uint32_t*)&q00;
        const uint32_t* w01 = (const uint32_t*)&q01;
        const uint32_t* w10 = (const uint32_t*)&q10;
        const uint32_t* w11 = (const uint32_t*)&q11;
        uint32_t r[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float lo, hi;
            {
                const float a00 = el16_lo(w00[t]), a01 = el16_lo(w01[t]);
                const float a10 = el16_lo(w10[t]), a11 = el16_lo(w11[t]);
                const float top = a00 * (1.0f - lx) + a01 * lx, bot = a10 * (1.0f - lx) + a11 * lx;
                lo = top * (1.0f - ly) + bot * ly;
            }
            {
                const float a00 = el16_hi(w00[t]), a01 = el16_hi(w01[t]);
                const float a10 = el16_hi(w10[t]), a11 = el16_hi(w11[t]);
                const float top = a00 * (1.0f - lx) + a01 * lx, bot = a10 * (1.0f - lx) + a11 * lx;
                hi = top * (1.0f - ly) + bot * ly;
            }
            r[t] = pack_el16x2(lo, hi);
        }
        *(uint4*)o = make_uint4(r[0], r[1], r[2], r[3]);
    } else {
        const float a00 = el16_to_f32(*p00), a01 = el16_to_f32(*p01), a10 = el16_to_f32(*p10), a11 = el16_to_f32(*p11);
        const float top = a00 * (1.0f - lx) + a01 * lx, bot = a10 * (1.0f - lx) + a11 * lx;
        *o = f32_to_el16(top * (1.0f - ly) + bot * ly);
    }
}

// Quad form: a thread produces the 2 x 2 output pixels of one input pixel (one 16-byte channel chunk) from its 3 x 3 input
// neighbourhood: 9 loads and one index decomposition per FOUR outputs (up2x_kernel<8>: 16 loads and three 64-bit divisions per
// four), same stencils (bilinear_coord) and the same expressions per output.
__global__ __launch_bounds__(256) void up2x_quad_kernel(Up2xArgs a, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = a.c0 + a.c1, groups = c >> 3;
    const int g = (int)(idx % groups);
    const int pix = (int)(idx / groups);           // input pixel (n, i, j)
    const int plane = a.h * a.w;
    const int n = pix / plane, rem = pix - n * plane;
    const int i = rem / a.w, j = rem - i * a.w;
    int ch = g * 8;
    const el16_t* src = a.src0;
    int cs = a.c0;
    if (ch >= a.c0) {
        src = a.src1;
        cs = a.c1;
        ch -= a.c0;
    }
    // rows / columns the four outputs read: output 2i reads (i-1, i), output 2i+1 reads (i, i+1), clamped by bilinear_coord
    int ya[2], yb[2], xa[2], xb[2];
    float ly[2], lx[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        bilinear_coord(2 * i + d, 0.5f, a.h, ya[d], yb[d], ly[d]);
        bilinear_coord(2 * j + d, 0.5f, a.w, xa[d], xb[d], lx[d]);
    }
    const int ys[3] = {ya[0], i, yb[1]}, xs[3] = {xa[0], j, xb[1]};  // (ya[1] == yb[0] == i except at a clamped border, where the weight of the other tap is 0 or the indices coincide)
    uint4 t[3][3];
    const el16_t* base = src + (size_t)n * plane * cs + ch;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) t[r][q] = *(const uint4*)(base + ((size_t)ys[r] * a.w + xs[q]) * cs);
    const int ow = 2 * a.w;
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            // taps of output (2i+dy, 2j+dx): rows ys[dy], ys[dy+1], columns xs[dx], xs[dx+1].  At a clamped border (output 0 of
            // input 0) bilinear_coord's second tap is index 1 with weight exactly 0; slot 1 holds index 0 there: finite * 0 either way
            const uint4 q00 = t[dy][dx], q01 = t[dy][dx + 1], q10 = t[dy + 1][dx], q11 = t[dy + 1][dx + 1];
            const uint32_t* w00 = (const uint32_t*)&q00;
            const uint32_t* w01 = (const uint32_t*)&q01;
            const uint32_t* w10 = (const uint32_t*)&q10;
            const uint32_t* w11 = (const uint32_t*)&q11;
            const float fx = lx[dx], fy = ly[dy];
            uint32_t o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float lo, hi;
                {
                    const float a00 = el16_lo(w00[k]), a01 = el16_lo(w01[k]), a10 = el16_lo(w10[k]), a11 = el16_lo(w11[k]);
                    const float top = a00 * (1.0f - fx) + a01 * fx, bot = a10 * (1.0f - fx) + a11 * fx;
                    lo = top * (1.0f - fy) + bot * fy;
                }
                {
                    const float a00 = el16_hi(w00[k]), a01 = el16_hi(w01[k]), a10 = el16_hi(w10[k]), a11 = el16_hi(w11[k]);
                    const float top = a00 * (1.0f - fx) + a01 * fx, bot = a10 * (1.0f - fx) + a11 * fx;
                    hi = top * (1.0f - fy) + bot * fy;
                }
                o[k] = pack_el16x2(lo, hi);
            }
            *(uint4*)(a.out + (((size_t)n * 2 * a.h + 2 * i + dy) * ow + 2 * j + dx) * c + g * 8) = make_uint4(o[0], o[1], o[2], o[3]);
        }
}

hipError_t launch_up2x(const Up2xArgs& a, hipStream_t s) {
    const int c = a.c0 + a.c1;
    const bool vec = (a.c0 % 8 == 0) && (a.c1 % 8 == 0);
    const long long total = (long long)a.n * 4 * a.h * a.w * (vec ? c / 8 : c);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    const char* env = dyf_form("DYF_UP2X_QUAD");  // read per launch (parity test)
    if (vec && !(env && atoi(env) == 0) && a.h >= 2 && a.w >= 2) {
        const long long quads = total / 4;
        dyf_form_note("up2x_quad_kernel", a.n);
        KernelProf kp("up2x_quad_kernel", s, (double)a.n * a.h * a.w * c * 2.0 * 5.0);  // read once, write 4x
        hipLaunchKernelGGL(up2x_quad_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, s, a, quads);
    } else if (vec)
        hipLaunchKernelGGL(up2x_kernel<8>, dim3(blocks), dim3(256), 0, s, a, total);
    else
        hipLaunchKernelGGL(up2x_kernel<1>, dim3(blocks), dim3(256), 0, s, a, total);
    return hipGetLastError();
}

// Upsample + epilogue of the commuted 1 x 1 decoder blocks (Up2xEpiArgs, kernels.h).  A thread owns one LOW-res pixel and 4 channels:
// 9 float4 loads (its 3 x 3 neighbourhood, stencils of bilinear_coord as in up2x_quad_kernel) make the 2 x 2 outputs.
__global__ __launch_bounds__(256) void up2x_epilogue_kernel(Up2xEpiArgs a, long long total) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int groups = a.c >> 2;
    const int g = (int)(idx % groups);
    const int pix = (int)(idx / groups);
    const int plane = a.h * a.w;
    const int n = pix / plane, rem = pix - n * plane;
    const int i = rem / a.w, j = rem - i * a.w;
    int ya[2], yb[2], xa[2], xb[2];
    float ly[2], lx[2];
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        bilinear_coord(2 * i + d, 0.5f, a.h, ya[d], yb[d], ly[d]);
        bilinear_coord(2 * j + d, 0.5f, a.w, xa[d], xb[d], lx[d]);
    }
    const int ys[3] = {ya[0], i, yb[1]}, xs[3] = {xa[0], j, xb[1]};
    float4 t[3][3];
    const float* base = a.lo + (size_t)n * plane * a.c + g * 4;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int q = 0; q < 3; ++q) t[r][q] = *(const float4*)(base + ((size_t)ys[r] * a.w + xs[q]) * a.c);
    const size_t ci = (size_t)(a.coef_div > 1 ? n / a.coef_div : n) * a.coef_stride + g * 4;
    const float4 ca = *(const float4*)(a.coef_a + ci), cc = *(const float4*)(a.coef_c + ci);
    const RngKey key = drop_row_key(a.drop, n);
    const int ow = 2 * a.w, oh = 2 * a.h;
    const uint32_t row0 = (uint32_t)n * (uint32_t)(oh * ow * a.c);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy)
#pragma unroll
        for (int dx = 0; dx < 2; ++dx) {
            const float4 q00 = t[dy][dx], q01 = t[dy][dx + 1], q10 = t[dy + 1][dx], q11 = t[dy + 1][dx + 1];
            const float fx = lx[dx], fy = ly[dy];
            auto lerp = [&](float a00, float a01, float a10, float a11) {
                const float top = a00 * (1.0f - fx) + a01 * fx, bot = a10 * (1.0f - fx) + a11 * fx;
                return top * (1.0f - fy) + bot * fy;
            };
            float v[4] = {fmaf(lerp(q00.x, q01.x, q10.x, q11.x), ca.x, cc.x), fmaf(lerp(q00.y, q01.y, q10.y, q11.y), ca.y, cc.y),
                          fmaf(lerp(q00.z, q01.z, q10.z, q11.z), ca.z, cc.z), fmaf(lerp(q00.w, q01.w, q10.w, q11.w), ca.w, cc.w)};
            const uint32_t e0 = (uint32_t)(((n * oh + 2 * i + dy) * ow + 2 * j + dx)) * (uint32_t)a.c + (uint32_t)(g * 4);
            act_drop<4>(v, e0, row0, a.act, a.drop, key);
            *(uint2*)(a.out + (size_t)e0) = make_uint2(pack_el16x2(v[0], v[1]), pack_el16x2(v[2], v[3]));
        }
}

hipError_t launch_up2x_epilogue(const Up2xEpiArgs& a, hipStream_t s) {
    if ((a.c & 3) != 0 || a.h < 1 || a.w < 1 || (size_t)a.n * 4 * a.h * a.w * a.c >= 0xFFFFFFF0ull) return hipErrorInvalidValue;
    const long long total = (long long)a.n * a.h * a.w * (a.c >> 2);
    dyf_form_note("up2x_epilogue_kernel", a.n);
    KernelProf kp("up2x_epilogue_kernel", s, (double)a.n * a.h * a.w * a.c * (4.0 + 4 * 2.0));  // fp32 low-res in, 16-bit x 4 out
    hipLaunchKernelGGL(up2x_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a, total);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K5 GroupNorm
// nn.GroupNorm(8, C) of the last encoder block (unet_simple.py:56) + FiLM + LeakyReLU + Dropout (:72-80).
__global__ __launch_bounds__(256) void groupnorm_kernel(GroupNormArgs a) {
    __shared__ float scratch[16];
    const int n = blockIdx.x / a.groups, g = blockIdx.x % a.groups;
    const int cpg = a.c / a.groups;
    const int count = a.hw * cpg;
    const float* x = a.x + (size_t)n * a.hw * a.c + g * cpg;
    float s = 0.0f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) s += x[(size_t)(i / cpg) * a.c + (i % cpg)];
    const float mean = block_sum(s, scratch) / (float)count;
    float v = 0.0f;
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const float d = x[(size_t)(i / cpg) * a.c + (i % cpg)] - mean;
        v = fmaf(d, d, v);
    }
    const float var = block_sum(v, scratch) / (float)count;  // biased, as torch
    const float rstd = rsqrtf(var + 1e-5f);
    const RngKey key = drop_row_key(a.drop, n);
    for (int i = threadIdx.x; i < count; i += blockDim.x) {
        const int p = i / cpg, ch = g * cpg + (i % cpg);
        float y = (x[(size_t)p * a.c + (i % cpg)] - mean) * rstd * a.gamma[ch] + a.beta[ch];
        const size_t fi = (size_t)(a.film_div > 1 ? n / a.film_div : n) * a.film_stride + ch;
        y = fmaf(y, a.film_a[fi], a.film_c[fi]);
        y = apply_act(y, a.act);
        const size_t e = ((size_t)n * a.hw + p) * a.c + ch;
        y = drop_apply(y, (uint32_t)e, (uint32_t)((size_t)n * a.hw * a.c), a.drop, key);
        a.out[e] = f32_to_el16(y);
    }
}

// Small planes (the GroupNorm block of unet_simple sits at the bottleneck: 4 x 4 pixels x 64 channels per group): ONE WAVE per
// (sample, group) keeps its <= 32 elements per lane in registers -- one pass over memory instead of three, wave reductions instead of
// two block reductions -- mean first, then the variance of the centred values (as groupnorm_kernel and ATen do).
template <int PER>  // float4 pieces per lane
__global__ __launch_bounds__(256) void groupnorm_wave_kernel(GroupNormArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int ng = blockIdx.x * 4 + wv;
    if (ng >= a.n * a.groups) return;
    const int n = ng / a.groups, g = ng - n * a.groups;
    const int cpg = a.c / a.groups, q4 = cpg >> 2;   // float4 pieces per pixel of the group
    const int pieces = a.hw * q4;
    const float* x = a.x + (size_t)n * a.hw * a.c + g * cpg;
    float4 v[PER];
    float s = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pc = lane + 64 * k;
        v[k] = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
        if (pc < pieces) {
            v[k] = *(const float4*)(x + (size_t)(pc / q4) * a.c + (pc % q4) * 4);
            s += (v[k].x + v[k].y) + (v[k].z + v[k].w);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) s += __shfl_xor(s, d, 64);
    const float count = (float)(a.hw * cpg);
    const float mean = s / count;
    float var = 0.0f;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        if (lane + 64 * k < pieces) {
            const float dx = v[k].x - mean, dy = v[k].y - mean, dz = v[k].z - mean, dw = v[k].w - mean;
            var = fmaf(dx, dx, var); var = fmaf(dy, dy, var); var = fmaf(dz, dz, var); var = fmaf(dw, dw, var);
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) var += __shfl_xor(var, d, 64);
    const float rstd = rsqrtf(var / count + 1e-5f);
    const RngKey key = drop_row_key(a.drop, n);
    const uint32_t row0 = (uint32_t)((size_t)n * a.hw * a.c);
    const size_t frow = (size_t)(a.film_div > 1 ? n / a.film_div : n) * a.film_stride;
#pragma unroll
    for (int k = 0; k < PER; ++k) {
        const int pc = lane + 64 * k;
        if (pc >= pieces) continue;
        const int p = pc / q4, ch = g * cpg + (pc % q4) * 4;
        const float4 ga = *(const float4*)(a.gamma + ch), be = *(const float4*)(a.beta + ch);
        const float4 fa = *(const float4*)(a.film_a + frow + ch), fc = *(const float4*)(a.film_c + frow + ch);
        float y[4] = {(v[k].x - mean) * rstd * ga.x + be.x, (v[k].y - mean) * rstd * ga.y + be.y, (v[k].z - mean) * rstd * ga.z + be.z,
                      (v[k].w - mean) * rstd * ga.w + be.w};
        y[0] = fmaf(y[0], fa.x, fc.x); y[1] = fmaf(y[1], fa.y, fc.y); y[2] = fmaf(y[2], fa.z, fc.z); y[3] = fmaf(y[3], fa.w, fc.w);
        const uint32_t e0 = row0 + (uint32_t)(p * a.c + ch);
        act_drop<4>(y, e0, row0, a.act, a.drop, key);
        *(uint2*)(a.out + (size_t)e0) = make_uint2(pack_el16x2(y[0], y[1]), pack_el16x2(y[2], y[3]));
    }
}

hipError_t launch_groupnorm(const GroupNormArgs& a, hipStream_t s) {
    const int cpg = a.groups > 0 ? a.c / a.groups : 0;
    const char* we = dyf_form("DYF_GN_WAVE");  // read per launch (parity test)
    if (!(we && atoi(we) == 0) && cpg > 0 && cpg * a.groups == a.c && (cpg & 3) == 0 && a.act != ACT_GELU &&
        (size_t)a.n * a.hw * a.c < 0xFFFFFFF0ull) {
        const int pieces = a.hw * (cpg >> 2);
        const unsigned blocks = (unsigned)((a.n * a.groups + 3) / 4);
        if (pieces <= 256) {
            dyf_form_note("groupnorm_wave_kernel", a.n);
            hipLaunchKernelGGL(groupnorm_wave_kernel<4>, dim3(blocks), dim3(256), 0, s, a);
            return hipGetLastError();
        }
        if (pieces <= 512) {
            dyf_form_note("groupnorm_wave_kernel", a.n);
            hipLaunchKernelGGL(groupnorm_wave_kernel<8>, dim3(blocks), dim3(256), 0, s, a);
            return hipGetLastError();
        }
    }
    hipLaunchKernelGGL(groupnorm_kernel, dim3(a.n * a.groups), dim3(256), 0, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K3+K4 readout
// ConvTranspose2d(dim -> C, k4, s2, p1) followed by the final bilinear resample to the native grid
// (unet_simple.py:141-151, :195).  The resample (no antialias) reads only 4 neighbours of each native pixel, so the
// transposed conv is evaluated at exactly those positions instead of materialising the 2x-resolution tensor.
// out[u] of a k4/s2/p1 transposed conv gathers input rows i with kh = u + 1 - 2i in [0, 3]: i = (u+1)>>1 and i - 1.
__global__ __launch_bounds__(256) void readout_kernel(ReadoutArgs a) {
    extern __shared__ float wsh[];  // [16][cin][cout]
    const int wcount = 16 * a.cin * a.cout;
    for (int i = threadIdx.x; i < wcount; i += blockDim.x) wsh[i] = a.wgt[i];
    __syncthreads();
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.n * a.oh * a.ow;
    if (idx >= total) return;
    const int n = (int)(idx / ((long long)a.oh * a.ow));
    const int rem = (int)(idx % ((long long)a.oh * a.ow));
    const int oy = rem / a.ow, ox = rem % a.ow;
    const int th = 2 * a.ih, tw = 2 * a.iw;  // transposed-conv output grid
    int u0, u1, v0, v1;
    float lu, lv;
    bilinear_coord(oy, (float)th / (float)a.oh, th, u0, u1, lu, a.nearest != 0);
    bilinear_coord(ox, (float)tw / (float)a.ow, tw, v0, v1, lv, a.nearest != 0);
    float acc[DYF_MAX_OUT_CH];
#pragma unroll
    for (int c = 0; c < DYF_MAX_OUT_CH; ++c) acc[c] = 0.0f;
    const int us[2] = {u0, u1}, vs[2] = {v0, v1};
    const float wu[2] = {1.0f - lu, lu}, wv[2] = {1.0f - lv, lv};
#pragma unroll
    for (int a_ = 0; a_ < 2; ++a_) {
#pragma unroll
        for (int b_ = 0; b_ < 2; ++b_) {
            const float bw = wu[a_] * wv[b_];
            const int u = us[a_], v = vs[b_];
            const int i_hi = (u + 1) >> 1, j_hi = (v + 1) >> 1;
#pragma unroll
            for (int di = 0; di < 2; ++di) {
                const int i = i_hi - di, kh = u + 1 - 2 * i;
                if ((unsigned)i >= (unsigned)a.ih) continue;
#pragma unroll
                for (int dj = 0; dj < 2; ++dj) {
                    const int j = j_hi - dj, kw = v + 1 - 2 * j;
                    if ((unsigned)j >= (unsigned)a.iw) continue;
                    const el16_t* px = a.x + (((size_t)n * a.ih + i) * a.iw + j) * a.cin;
                    const float* wt = wsh + (size_t)(kh * 4 + kw) * a.cin * a.cout;
                    for (int ci = 0; ci < a.cin; ++ci) {
                        const float xv = bw * el16_to_f32(px[ci]);
#pragma unroll
                        for (int co = 0; co < DYF_MAX_OUT_CH; ++co)
                            if (co < a.cout) acc[co] = fmaf(xv, wt[ci * a.cout + co], acc[co]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int co = 0; co < DYF_MAX_OUT_CH; ++co)
        if (co < a.cout) a.out[(((size_t)n * a.cout + co) * a.oh + oy) * a.ow + ox] = acc[co] + a.bias[co];
}

// Fast form for cin % 8 == 0 with cin/8 a power of two: LANES = cin/8 lanes share one native pixel, each owns an
// 8-channel slice, so every tap is ONE 16-B load per lane (a full 128-B line per pixel for dim 64) instead of 64
// scattered 2-B loads; partial sums are combined with a butterfly over the LANES lanes.
template <int LANES>
__global__ __launch_bounds__(256) void readout_sliced_kernel(ReadoutArgs a) {
    extern __shared__ float wsh[];  // [16][cin][cout]
    const int wcount = 16 * a.cin * a.cout;
    for (int i = threadIdx.x; i < wcount; i += blockDim.x) wsh[i] = a.wgt[i];
    __syncthreads();
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long total = (long long)a.n * a.oh * a.ow;
    long long idx = gid / LANES;
    const int slice = (int)(gid % LANES);
    const bool live = idx < total;
    if (!live) idx = total - 1;  // keep the whole wave in the shuffles
    const int n = (int)(idx / ((long long)a.oh * a.ow));
    const int rem = (int)(idx % ((long long)a.oh * a.ow));
    const int oy = rem / a.ow, ox = rem % a.ow;
    const int th = 2 * a.ih, tw = 2 * a.iw;
    int u0, u1, v0, v1;
    float lu, lv;
    bilinear_coord(oy, (float)th / (float)a.oh, th, u0, u1, lu, a.nearest != 0);
    bilinear_coord(ox, (float)tw / (float)a.ow, tw, v0, v1, lv, a.nearest != 0);
    float acc[DYF_MAX_OUT_CH];
#pragma unroll
    for (int c = 0; c < DYF_MAX_OUT_CH; ++c) acc[c] = 0.0f;
    const int ci0 = slice * 8;
    // all 16 (neighbour, tap) input vectors are requested up front (one 16-B load each, independent), so the kernel is
    // not a chain of 16 dependent L2 round trips; taps outside the image contribute zero
    uint4 qv[16];
    int wofs[16];
    float bwv[16];
#pragma unroll
    for (int nb4 = 0; nb4 < 4; ++nb4) {
        const int u = (nb4 & 2) ? u1 : u0, v = (nb4 & 1) ? v1 : v0;
        const float bw = ((nb4 & 2) ? lu : 1.0f - lu) * ((nb4 & 1) ? lv : 1.0f - lv);
        const int i_hi = (u + 1) >> 1, j_hi = (v + 1) >> 1;
#pragma unroll
        for (int t4 = 0; t4 < 4; ++t4) {
            const int i = i_hi - (t4 >> 1), j = j_hi - (t4 & 1);
            const int kh = u + 1 - 2 * i, kw = v + 1 - 2 * j;
            const bool ok = (unsigned)i < (unsigned)a.ih && (unsigned)j < (unsigned)a.iw;
            const int ic = ok ? i : 0, jc = ok ? j : 0;
            const uint4 ld = *(const uint4*)(a.x + (((size_t)n * a.ih + ic) * a.iw + jc) * a.cin + ci0);
            qv[nb4 * 4 + t4] = ok ? ld : make_uint4(0u, 0u, 0u, 0u);
            bwv[nb4 * 4 + t4] = ok ? bw : 0.0f;
            wofs[nb4 * 4 + t4] = ((kh * 4 + kw) * a.cin + ci0) * a.cout;
        }
    }
#pragma unroll
    for (int s16 = 0; s16 < 16; ++s16) {
        const uint32_t qw[4] = {qv[s16].x, qv[s16].y, qv[s16].z, qv[s16].w};
        const float* wt = wsh + wofs[s16];
        const float bw = bwv[s16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float xv = bw * ((c & 1) ? el16_hi(qw[c >> 1]) : el16_lo(qw[c >> 1]));
#pragma unroll
            for (int co = 0; co < DYF_MAX_OUT_CH; ++co)
                if (co < a.cout) acc[co] = fmaf(xv, wt[c * a.cout + co], acc[co]);
        }
    }
#pragma unroll
    for (int co = 0; co < DYF_MAX_OUT_CH; ++co) {
        if (co < a.cout) {
            float v = acc[co];
#pragma unroll
            for (int off = LANES >> 1; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (live && slice == 0) a.out[(((size_t)n * a.cout + co) * a.oh + oy) * a.ow + ox] = v + a.bias[co];
        }
    }
}

// Register-weight form for cin == 64 (the NS backbone): the sliced kernel above is bound by its 384 LDS weight reads per
// thread.  Every native pixel uses each of the 16 (kh, kw) taps of the transposed conv exactly once (u0/u1 and v0/v1 have
// opposite parities), reading a 3x3 window of decoder pixels.  So 32 lanes share a pixel: lane (kh = 0..3, slice = 0..7)
// keeps the weights of its kernel row and 8-channel slice in REGISTERS (4 kw x 8 c x CO), loads the 4 input vectors of
// its row (16 B each; 8 slices = one 128-B line) and the 32 partial sums are combined with a butterfly.  A half-wave
// walks a contiguous run of native pixels (oy fastest: consecutive pixels share two of the three input rows in L1).
template <int CO>
__global__ __launch_bounds__(256) void readout_regw_kernel(ReadoutArgs a, int px_per_half) {
    const int lane = threadIdx.x & 63;
    const int half = (int)((blockIdx.x * 256 + threadIdx.x) >> 5);
    const int kh = (lane >> 3) & 3, slice = lane & 7, ci0 = slice * 8;
    float w[4][8][CO];
#pragma unroll
    for (int kw = 0; kw < 4; ++kw)
#pragma unroll
        for (int c = 0; c < 8; ++c)
#pragma unroll
            for (int co = 0; co < CO; ++co) w[kw][c][co] = a.wgt[((size_t)(kh * 4 + kw) * a.cin + ci0 + c) * CO + co];
    float bias[CO];
#pragma unroll
    for (int co = 0; co < CO; ++co) bias[co] = a.bias[co];
    const int total = a.n * a.oh * a.ow;
    const int th = 2 * a.ih, tw = 2 * a.iw;
    const float sh = (float)th / (float)a.oh, sw = (float)tw / (float)a.ow;
    const int pu = (kh + 1) & 1;
    // pixel order: (n, ox, oy), oy fastest; the position is advanced incrementally (no per-pixel divisions)
    int idx = half * px_per_half;
    int n, ox, oy;
    {
        const int ic = idx < total ? idx : total - 1;
        n = ic / (a.oh * a.ow);
        const int rem = ic - n * (a.oh * a.ow);
        ox = rem / a.oh;
        oy = rem - ox * a.oh;
    }
    // one pixel's inputs: 4 vectors of this lane's kernel row + their bilinear weights
    struct Px {
        uint4 q[4];
        float bw[4];
        int out_off;  // element offset of channel 0 in `out`, or -1
    };
    auto fetch = [&](Px& p) {
        const bool live = idx < total;
        int u0, u1, v0, v1;
        float lu, lv;
        bilinear_coord(oy, sh, th, u0, u1, lu, a.nearest != 0);
        bilinear_coord(ox, sw, tw, v0, v1, lv, a.nearest != 0);
        // kernel row kh pairs with the neighbour u of parity (kh + 1) & 1: kh = u + 1 - 2 i, i = ((u + 1) >> 1) - (kh >> 1)
        const float wu = ((u0 & 1) == pu ? 1.0f - lu : 0.0f) + ((u1 & 1) == pu ? lu : 0.0f);
        const int u = (u1 & 1) == pu ? u1 : u0;
        const int i = ((u + 1) >> 1) - (kh >> 1);
        const bool row_ok = (unsigned)i < (unsigned)a.ih;
        const el16_t* row = a.x + ((size_t)n * a.ih + (row_ok ? i : 0)) * a.iw * a.cin + ci0;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const int pv = (kw + 1) & 1;
            const float wv = ((v0 & 1) == pv ? 1.0f - lv : 0.0f) + ((v1 & 1) == pv ? lv : 0.0f);
            const int v = (v1 & 1) == pv ? v1 : v0;
            const int j = ((v + 1) >> 1) - (kw >> 1);
            const bool ok = row_ok && (unsigned)j < (unsigned)a.iw;
            const uint4 qv = *(const uint4*)(row + (size_t)(ok ? j : 0) * a.cin);
            // taps outside the image contribute exactly zero whatever the (possibly never written) fallback pixel holds
            p.q[kw] = ok ? qv : make_uint4(0u, 0u, 0u, 0u);
            p.bw[kw] = ok ? wu * wv : 0.0f;
        }
        p.out_off = live ? ((n * CO) * a.oh + oy) * a.ow + ox : -1;
        // advance (clamped at the last pixel so that the whole wave stays in the shuffles)
        ++idx;
        if (idx < total) {
            if (++oy == a.oh) {
                oy = 0;
                if (++ox == a.ow) { ox = 0; ++n; }
            }
        }
    };
    auto reduce_store = [&](const Px& p) {
        float acc[CO];
#pragma unroll
        for (int co = 0; co < CO; ++co) acc[co] = 0.0f;
#pragma unroll
        for (int kw = 0; kw < 4; ++kw) {
            const uint32_t qw[4] = {p.q[kw].x, p.q[kw].y, p.q[kw].z, p.q[kw].w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float xv = p.bw[kw] * ((c & 1) ? el16_hi(qw[c >> 1]) : el16_lo(qw[c >> 1]));
#pragma unroll
                for (int co = 0; co < CO; ++co) acc[co] = fmaf(xv, w[kw][c][co], acc[co]);
            }
        }
#pragma unroll
        for (int co = 0; co < CO; ++co) {
            float v = acc[co];
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
            if (p.out_off >= 0 && (lane & 31) == 0) a.out[(size_t)p.out_off + (size_t)co * a.oh * a.ow] = v + bias[co];
        }
    };
    // two pixels in flight: the loads of pixel it+1 are issued before the FMAs of pixel it
    Px p0, p1;
    fetch(p0);
    for (int it = 0; it < px_per_half; it += 2) {
        fetch(p1);
        reduce_store(p0);
        fetch(p0);
        reduce_store(p1);
    }
}

// MFMA form for cin == 64, cout <= 4 (the NS backbone): the register-weight form above is VALU-bound (~230 VALU per lane and
// pixel).  Here a wave takes 16 native pixels at a time; for each of the 16 (kh, kw) taps the 64-channel contraction of all 16
// pixels is two v_mfma_f32_16x16x32_bf16 (A = the tap's weights as a [16 (cout, zero-padded)][32 k] fragment held in
// registers for the whole kernel, B = the pixels' input vectors: lane (pixel p, k-group g) loads 16 B of pixel p's tap
// vector, exactly its MFMA operand), and the bilinear weight of the tap scales the 3 result rows with 3 FMAs.  Coordinates
// are computed once per pixel by 4 lanes instead of 32.  Weights are bf16 here (fp32 in the other forms).
typedef __attribute__((ext_vector_type(4))) float ro_f32x4;

__global__ __launch_bounds__(256, 4) void readout_mfma_kernel(ReadoutArgs a, int groups_per_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
    // weight fragments [tap][k half][lane] x 16 B in LDS (32 KB): lane (m = lane & 15, k-group) holds
    // W[tap][half*32 + kg*8 + e][co = m] (0 for m >= cout); fragment reads are lane-linear, i.e. conflict-free
    extern __shared__ __attribute__((aligned(16))) char ro_smem[];
    for (int i = threadIdx.x; i < 16 * 2 * 64; i += 256) ((uint4*)ro_smem)[i] = ((const uint4*)a.wfrag)[i];
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int wave_id = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
    const int p = lane & 15, kg = lane >> 4;
    const el16x8_t* wl = (const el16x8_t*)ro_smem + lane;
    const int total = a.n * a.oh * a.ow;
    const int th = 2 * a.ih, tw = 2 * a.iw;
    const float sh = (float)th / (float)a.oh, sw = (float)tw / (float)a.ow;
    const int plane = a.oh * a.ow;
    for (int g = 0; g < groups_per_wave; ++g) {
        const int idx0 = (wave_id * groups_per_wave + g) * 16;
        if (idx0 >= total) break;  // wave-uniform
        int idx = idx0 + p;
        const bool live = idx < total;
        if (!live) idx = total - 1;
        // pixel order (n, oy, ox), ox fastest: the 16 pixels of a group read the same three input rows.  (Tried: vertical strips,
        // oy fastest, whose 4-row windows overlap -- 90 instead of 256 distinct input pixels per group, but every load then
        // touches 16 different rows: 312 vs 253 us per paired launch.)
        const int n = idx / plane;
        const int rem = idx - n * plane;
        const int oy = rem / a.ow, ox = rem - oy * a.ow;
        int u0, u1, v0, v1;
        float lu, lv;
        bilinear_coord(oy, sh, th, u0, u1, lu, a.nearest != 0);
        bilinear_coord(ox, sw, tw, v0, v1, lv, a.nearest != 0);
        // per kernel row / column: byte offset of the input row / column (or -1) and bilinear weight (see readout_regw_kernel)
        int ro[4], co_[4];
        float rw[4], cw[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int par = (k + 1) & 1;
            rw[k] = ((u0 & 1) == par ? 1.0f - lu : 0.0f) + ((u1 & 1) == par ? lu : 0.0f);
            const int u = (u1 & 1) == par ? u1 : u0;
            const int i = ((u + 1) >> 1) - (k >> 1);
            ro[k] = (unsigned)i < (unsigned)a.ih ? i * a.iw_store * 64 : -1;
            cw[k] = ((v0 & 1) == par ? 1.0f - lv : 0.0f) + ((v1 & 1) == par ? lv : 0.0f);
            const int v = (v1 & 1) == par ? v1 : v0;
            const int j = ((v + 1) >> 1) - (k >> 1);
            int js = (unsigned)j < (unsigned)a.iw ? j : -1;
            if (js >= 0 && a.col_map) js = a.col_map[js];  // compact input: stored column of j (always stored when read)
            co_[k] = js >= 0 ? js * 64 : -1;
        }
        // LOADS are issued with a coalesced lane map -- lane s fetches 16-B chunk (s & 3) of pixel (s >> 2), so a quad of
        // lanes reads 64 contiguous bytes (with the MFMA's own lane map, pixel = lane & 15, every quad touches four cache
        // lines and the texture path needs 4x the cycles) -- and moved to their MFMA lane with ds_bpermute:
        // MFMA lane t = (pixel t & 15, k-group t >> 4) takes the registers of load lane (t & 15) * 4 + (t >> 4).
        const int lp = lane >> 2, lc = lane & 3;  // pixel / chunk this lane LOADS
        const int src4 = (lp)*4;                  // byte address (lane * 4) of a lane that holds pixel lp's coordinates: lane lp
        int l_ro[4], l_co[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            l_ro[k] = __builtin_amdgcn_ds_bpermute(src4, ro[k]);
            l_co[k] = __builtin_amdgcn_ds_bpermute(src4, co_[k]);
        }
        const int l_n = __builtin_amdgcn_ds_bpermute(src4, n);
        const el16_t* img = a.x + (size_t)l_n * a.ih * a.iw_store * 64 + lc * 8;
        const int take4 = ((lane & 15) * 4 + (lane >> 4)) * 4;
        float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // two batches of 8 taps (2 kernel rows): all 16 vector loads of a batch are in flight before its MFMAs
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            uint4 q0[8], q1[8];
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int kh = b * 2 + (t >> 2), kw = t & 3;
                const bool ok = l_ro[kh] >= 0 && l_co[kw] >= 0;
                const el16_t* px = img + (ok ? l_ro[kh] + l_co[kw] : 0);
                q0[t] = *(const uint4*)px;
                q1[t] = *(const uint4*)(px + 32);
                if (!ok) { q0[t] = make_uint4(0u, 0u, 0u, 0u); q1[t] = q0[t]; }  // taps outside the image: data 0
            }
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                const int kh = b * 2 + (t >> 2), kw = t & 3, tap = kh * 4 + kw;
                uint4 m0, m1;
                m0.x = __builtin_amdgcn_ds_bpermute(take4, q0[t].x); m0.y = __builtin_amdgcn_ds_bpermute(take4, q0[t].y);
                m0.z = __builtin_amdgcn_ds_bpermute(take4, q0[t].z); m0.w = __builtin_amdgcn_ds_bpermute(take4, q0[t].w);
                m1.x = __builtin_amdgcn_ds_bpermute(take4, q1[t].x); m1.y = __builtin_amdgcn_ds_bpermute(take4, q1[t].y);
                m1.z = __builtin_amdgcn_ds_bpermute(take4, q1[t].z); m1.w = __builtin_amdgcn_ds_bpermute(take4, q1[t].w);
                ro_f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                d = DYF_MFMA_16x16x32(wl[(tap * 2 + 0) * 64], __builtin_bit_cast(el16x8_t, m0), d, 0, 0, 0);
                d = DYF_MFMA_16x16x32(wl[(tap * 2 + 1) * 64], __builtin_bit_cast(el16x8_t, m1), d, 0, 0, 0);
                const float bw = (ro[kh] >= 0 && co_[kw] >= 0) ? rw[kh] * cw[kw] : 0.0f;  // this lane's own pixel
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = fmaf(bw, d[r], out[r]);
            }
        }
        // rows 0..3 of the 16x16 result (= output channels) live in lanes 0-15 (k-group 0), column = pixel
        if (live && kg == 0) {
            float* o = a.out + ((size_t)n * a.cout * a.oh + oy) * a.ow + ox;
            const size_t cs = (size_t)a.oh * a.ow;
            o[0] = out[0] + a.bias[0];
            if (a.cout > 1) o[cs] = out[1] + a.bias[1];
            if (a.cout > 2) o[2 * cs] = out[2] + a.bias[2];
            if (a.cout > 3) o[3 * cs] = out[3] + a.bias[3];
        }
    }
#endif
}

// Tap tables of the readout for one geometry: the coordinate arithmetic of readout_mfma_kernel, evaluated once per output row /
// column instead of once per pixel and forward (the MFMA forms spend most of their issue slots on it: 16 pixels per wave).
__global__ void readout_tables_kernel(ReadoutArgs a, uint4* row_tab, uint4* col_tab) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int th = 2 * a.ih, tw = 2 * a.iw;
    if (t < a.oh) {
        int u0, u1;
        float lu;
        bilinear_coord(t, (float)th / (float)a.oh, th, u0, u1, lu, a.nearest != 0);
        int off[4];
        float w[4];
        for (int k = 0; k < 4; ++k) {
            const int par = (k + 1) & 1;
            w[k] = ((u0 & 1) == par ? 1.0f - lu : 0.0f) + ((u1 & 1) == par ? lu : 0.0f);
            const int u = (u1 & 1) == par ? u1 : u0;
            const int i = ((u + 1) >> 1) - (k >> 1);
            off[k] = (unsigned)i < (unsigned)a.ih ? i * a.iw_store * 128 : -1;
        }
        row_tab[2 * t] = make_uint4((unsigned)off[0], (unsigned)off[1], (unsigned)off[2], (unsigned)off[3]);
        row_tab[2 * t + 1] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    }
    if (t < a.ow) {
        int v0, v1;
        float lv;
        bilinear_coord(t, (float)tw / (float)a.ow, tw, v0, v1, lv, a.nearest != 0);
        int off[4];
        float w[4];
        for (int k = 0; k < 4; ++k) {
            const int par = (k + 1) & 1;
            w[k] = ((v0 & 1) == par ? 1.0f - lv : 0.0f) + ((v1 & 1) == par ? lv : 0.0f);
            const int v = (v1 & 1) == par ? v1 : v0;
            const int j = ((v + 1) >> 1) - (k >> 1);
            int js = (unsigned)j < (unsigned)a.iw ? j : -1;
            if (js >= 0 && a.col_map) js = a.col_map[js];
            off[k] = js >= 0 ? js * 128 : -1;
        }
        col_tab[2 * t] = make_uint4((unsigned)off[0], (unsigned)off[1], (unsigned)off[2], (unsigned)off[3]);
        col_tab[2 * t + 1] = make_uint4(__float_as_uint(w[0]), __float_as_uint(w[1]), __float_as_uint(w[2]), __float_as_uint(w[3]));
    }
}

hipError_t launch_readout_tables(const ReadoutArgs& a, hipStream_t s) {
    const int n = a.oh > a.ow ? a.oh : a.ow;
    hipLaunchKernelGGL(readout_tables_kernel, dim3((n + 255) / 256), dim3(256), 0, s, a, (uint4*)a.row_tab, (uint4*)a.col_tab);
    return hipGetLastError();
}

// The same arithmetic with the gather staged through LDS by DMA and the tap coordinates from tables.  Measured per paired launch
// (160 rows): 256 us (register shuffles) -> 247 (DMA staging, two-deep batch pipeline) -> 239 (tap tables).  Tried and slower: a
// vertical-strip form that stages every input row segment once (2.9x less gather traffic: 284 us) -- the kernel is bound by none
// of gather bytes, shuffles or latency alone but by the LDS pipe serving 64 fragment reads per 16 pixels for an MFMA whose 16
// result rows hold 3 output channels.  readout_mfma_kernel loads with a coalesced lane map (a quad of
// lanes = 64 contiguous bytes of one pixel) and moves every 16-byte piece to its MFMA lane with four ds_bpermute: 128 shuffles
// per 16 pixels, and the LDS pipe -- not the gather -- sets its time.  Here `buffer_load ... lds` writes the quads straight into
// a per-wave staging area (lane i fetches piece (i & 3) ^ (pixel & 3) of pixel i >> 2 into slot i: the swizzle keeps the
// fragment reads conflict-free) and the MFMA lane (pixel p, k-group g) reads its operand back with ONE ds_read_b128 from slot
// 4p + (g ^ (p & 3)).  Taps outside the image get the out-of-range offset (DMA writes zeros).  Only the 4 non-zero rows of the
// weight fragments are kept in LDS (8 KB + 4 x 8 KB staging per workgroup: four workgroups per CU as before).
__global__ __launch_bounds__(256, 4) void readout_dma_kernel(ReadoutArgs a, int groups_per_wave) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char rd_smem[];
    // weights: [tap][half][k-group][row m < 4] x 16 B  (row m of the 16-row A fragment; rows >= cout are zero in wfrag)
    for (int i = threadIdx.x; i < 16 * 2 * 4 * 4; i += 256) {
        const int m = i & 3, kg = (i >> 2) & 3, th = i >> 4;
        ((uint4*)rd_smem)[i] = ((const uint4*)a.wfrag)[th * 64 + kg * 16 + m];
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wave_id = (int)((blockIdx.x * 256 + threadIdx.x) >> 6);
    const int p = lane & 15, kg = lane >> 4;
    char* stage = rd_smem + 8192 + wave * 8192;  // [4 taps][2 halves][64 slots] x 16 B
    const uint4* wl = (const uint4*)rd_smem + kg * 4 + (p & 3);
    const bool wrow = p < 4;
    const int total = a.n * a.oh * a.ow;
    const int th = 2 * a.ih, tw = 2 * a.iw;
    const float sh = (float)th / (float)a.oh, sw = (float)tw / (float)a.ow;
    const int plane = a.oh * a.ow;
    const size_t xbytes = (size_t)a.n * a.ih * a.iw_store * 64 * sizeof(el16_t);
    const auto rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a.x, 0, (int)(unsigned)xbytes, 0x00020000);
    const int lp = lane >> 2, lc = lane & 3;             // pixel / slot piece this lane FETCHES
    const unsigned piece_off = (unsigned)((lc ^ (lp & 3)) * 16);
    const int src4 = lp * 4;                             // ds_bpermute byte address of lane lp (holds pixel lp's coordinates)
    const unsigned rd_off = (unsigned)((p * 4 + (kg ^ (p & 3))) * 16);
    for (int g = 0; g < groups_per_wave; ++g) {
        const int idx0 = (wave_id * groups_per_wave + g) * 16;
        if (idx0 >= total) break;  // wave-uniform
        int idx = idx0 + p;
        const bool live = idx < total;
        if (!live) idx = total - 1;
        const int n = idx / plane;
        const int rem = idx - n * plane;
        const int oy = rem / a.ow, ox = rem - oy * a.ow;
        // tap tables (readout_tables_kernel): row offsets / weights of this output row, column offsets / weights of this column
        const uint4 rto = a.row_tab[2 * oy], rtw = a.row_tab[2 * oy + 1], cto = a.col_tab[2 * ox], ctw = a.col_tab[2 * ox + 1];
        const int nbase = n * a.ih * a.iw_store * 128;
        const int rto_[4] = {(int)rto.x, (int)rto.y, (int)rto.z, (int)rto.w};
        const int ro[4] = {rto_[0] >= 0 ? nbase + rto_[0] : -1, rto_[1] >= 0 ? nbase + rto_[1] : -1, rto_[2] >= 0 ? nbase + rto_[2] : -1,
                           rto_[3] >= 0 ? nbase + rto_[3] : -1};
        const int co_[4] = {(int)cto.x, (int)cto.y, (int)cto.z, (int)cto.w};
        const float rw[4] = {__uint_as_float(rtw.x), __uint_as_float(rtw.y), __uint_as_float(rtw.z), __uint_as_float(rtw.w)};
        const float cw[4] = {__uint_as_float(ctw.x), __uint_as_float(ctw.y), __uint_as_float(ctw.z), __uint_as_float(ctw.w)};
        int l_ro[4], l_co[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            l_ro[k] = __builtin_amdgcn_ds_bpermute(src4, ro[k]);
            l_co[k] = __builtin_amdgcn_ds_bpermute(src4, co_[k]);
        }
        float out[4] = {0.0f, 0.0f, 0.0f, 0.0f};
        // 8 batches of 2 taps, two staging halves: the DMA of batch b + 1 is in flight while batch b is consumed (with one batch
        // of 4 taps at a time every group paid four exposed memory round trips -- the kernel's time was that latency chain)
        auto issue = [&](int bt) {
            const int kh = bt >> 1;
            char* st = stage + (bt & 1) * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kw = (bt & 1) * 2 + t;
                const bool ok = l_ro[kh] >= 0 && l_co[kw] >= 0;
                const unsigned vo = ok ? (unsigned)(l_ro[kh] + l_co[kw]) + piece_off : 0xFFFFFFFFu;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(st + (t * 2 + 0) * 1024), 16, vo, 0, 0, 0);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(st + (t * 2 + 1) * 1024), 16,
                                                         ok ? vo + 64u : 0xFFFFFFFFu, 0, 0, 0);
            }
        };
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the previous group's fragment reads are complete
        issue(0);
#pragma unroll
        for (int bt = 0; bt < 8; ++bt) {
            if (bt + 1 < 8) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // batch bt - 1 has been read out of the half that is refilled
                issue(bt + 1);
                asm volatile("s_waitcnt vmcnt(4)" ::: "memory");    // batch bt has landed
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            const int kh = bt >> 1;
            const char* st = stage + (bt & 1) * 4096;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int kw = (bt & 1) * 2 + t, tap = kh * 4 + kw;
                const uint4 b0 = *(const uint4*)(st + (t * 2 + 0) * 1024 + rd_off);
                const uint4 b1 = *(const uint4*)(st + (t * 2 + 1) * 1024 + rd_off);
                uint4 w0 = make_uint4(0u, 0u, 0u, 0u), w1 = w0;
                if (wrow) {
                    w0 = wl[(tap * 2 + 0) * 16];
                    w1 = wl[(tap * 2 + 1) * 16];
                }
                ro_f32x4 d = {0.0f, 0.0f, 0.0f, 0.0f};
                d = DYF_MFMA_16x16x32(__builtin_bit_cast(el16x8_t, w0), __builtin_bit_cast(el16x8_t, b0), d, 0, 0, 0);
                d = DYF_MFMA_16x16x32(__builtin_bit_cast(el16x8_t, w1), __builtin_bit_cast(el16x8_t, b1), d, 0, 0, 0);
                const float bw = (ro[kh] >= 0 && co_[kw] >= 0) ? rw[kh] * cw[kw] : 0.0f;  // this lane's own pixel
#pragma unroll
                for (int r = 0; r < 4; ++r) out[r] = fmaf(bw, d[r], out[r]);
            }
        }
        if (live && kg == 0) {
            float* o = a.out + ((size_t)n * a.cout * a.oh + oy) * a.ow + ox;
            const size_t cs = (size_t)a.oh * a.ow;
            o[0] = out[0] + a.bias[0];
            if (a.cout > 1) o[cs] = out[1] + a.bias[1];
            if (a.cout > 2) o[2 * cs] = out[2] + a.bias[2];
            if (a.cout > 3) o[3 * cs] = out[3] + a.bias[3];
        }
    }
#endif
}

hipError_t launch_readout(const ReadoutArgs& a, hipStream_t s) {
    const long long total = (long long)a.n * a.oh * a.ow;
    const bool regw = !(dyf_form("DYF_READOUT_REGW") && atoi(dyf_form("DYF_READOUT_REGW")) == 0);
    const bool use_mfma = !(dyf_form("DYF_READOUT_MFMA") && atoi(dyf_form("DYF_READOUT_MFMA")) == 0);
    if (a.col_map && !(a.wfrag && a.cin == 64 && a.cout >= 1 && a.cout <= 4)) return hipErrorInvalidValue;
    if ((use_mfma || a.col_map) && a.wfrag && a.cin == 64 && a.cout >= 1 && a.cout <= 4 && total < (1ll << 30)) {
        const long long groups = (total + 15) / 16;
        long long waves = 256ll * 16;  // 4 waves per SIMD
        int per = (int)((groups + waves - 1) / waves);
        if (per < 1) per = 1;
        waves = (groups + per - 1) / per;
        // DMA-staged gather (DYF_READOUT_DMA=0: the register-shuffle form); the DMA's buffer descriptor addresses < 4 GB
        const bool use_dma = !(dyf_form("DYF_READOUT_DMA") && atoi(dyf_form("DYF_READOUT_DMA")) == 0);
        const bool dma = use_dma && a.row_tab && a.col_tab && (size_t)a.n * a.ih * a.iw_store * 128 < 0x7F000000ull;
        dyf_form_note(dma ? "readout_dma_kernel" : "readout_mfma_kernel", a.n);
        KernelProf kp(dma ? "readout_dma_kernel" : "readout_mfma_kernel", s,
                      (double)a.n * a.ih * a.iw_store * a.cin * 2.0 + (double)a.n * a.cout * a.oh * a.ow * 4.0);
        if (dma)
            hipLaunchKernelGGL(readout_dma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 8192 + 4 * 8192, s, a, per);
        else
            hipLaunchKernelGGL(readout_mfma_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 32768, s, a, per);
        return hipGetLastError();
    }
    if (regw && a.cin == 64 && a.cout >= 1 && a.cout <= 4 && total < (1ll << 30)) {
        // ~16 waves per CU; a half-wave owns a contiguous run of pixels
        int halves = 256 * 16 * 2;
        int per = (int)((total + halves - 1) / halves);
        if (per < 8) per = 8;
        per += per & 1;  // the kernel walks two pixels per iteration
        halves = (int)((total + per - 1) / per);
        const unsigned blocks = (unsigned)((halves + 7) / 8);
#define RW_CASE(C)                                                                                      \
        if (a.cout == C) {                                                                              \
            hipLaunchKernelGGL(readout_regw_kernel<C>, dim3(blocks), dim3(256), 0, s, a, per);          \
            return hipGetLastError();                                                                   \
        }
        RW_CASE(1) RW_CASE(2) RW_CASE(3) RW_CASE(4)
#undef RW_CASE
    }
    const size_t lds = (size_t)16 * a.cin * a.cout * sizeof(float);
    const int lanes = (a.cin % 8 == 0) ? a.cin / 8 : 0;
#define RO_CASE(L)                                                                                                   \
    if (lanes == L) {                                                                                                \
        hipLaunchKernelGGL(readout_sliced_kernel<L>, dim3((unsigned)((total * L + 255) / 256)), dim3(256), lds, s, a); \
        return hipGetLastError();                                                                                    \
    }
    RO_CASE(1) RO_CASE(2) RO_CASE(4) RO_CASE(8) RO_CASE(16)
#undef RO_CASE
    hipLaunchKernelGGL(readout_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), lds, s, a);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ K10 sampler
// dyffusion.py:388  x_s = x_s - I(s) + I(s_next), kept in fp32 (SURVEY hard-part (f))
// copy: second destination of the new x_s (the forecast-stack slot of a step that emits a prediction; saves the copy node), or null
__global__ void cold_update_kernel(float* x_s, const float* x_cur, const float* x_next, long long count, float* copy) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) {
        const float v = x_s[i] - x_cur[i] + x_next[i];
        x_s[i] = v;
        if (copy) copy[i] = v;
    }
}

hipError_t launch_cold_update(float* x_s, const float* x_cur, const float* x_next, long long count, hipStream_t s, float* copy) {
    hipLaunchKernelGGL(cold_update_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, x_s, x_cur, x_next,
                       count, copy);
    return hipGetLastError();
}

// dyffusion.py:219-227  forward_cond = tfactor*condition + (1-tfactor)*randn_like(condition)
// Engine draws (noise == nullptr): Box-Muller on two hashed words per element; the stream is keyed by (seed, noise-call
// counter rng_state[4], GLOBAL batch row rng_state[3] + n) and the element index inside the row, like the dropout masks.
__global__ void noisy_condition_kernel(float* out, const float* cond, const float* noise, float tau, long long count,
                                       int row_elems, const uint32_t* rng_state) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    float z;
    if (noise != nullptr) {
        z = noise[i];
    } else {
        const uint32_t n = (uint32_t)(i / row_elems), e = (uint32_t)(i - (long long)n * row_elems);
        const RngKey rk = rng_row_key(rng_state[0], rng_state[1] ^ 0x4E6F6973u, rng_state[4], rng_state[3] + n);
        const uint32_t w0 = fmix32(e * 0x9E3779B1u + rk.k0), w1 = fmix32(w0 ^ rk.k1);
        const float u1 = ((float)(w0 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        const float u2 = ((float)(w1 >> 8) + 0.5f) * (1.0f / 16777216.0f);
        z = sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
    }
    out[i] = tau * cond[i] + (1.0f - tau) * z;
}

__global__ void rng_bump_kernel(uint32_t* state, int slot, uint32_t by) {
    if (threadIdx.x == 0 && blockIdx.x == 0) state[slot] += by;
}

// row groups (engine.hip): a child engine's stream position = the parent's, its batch row 0 = the parent's row 0 + row_add;
// counters_only: the parent adopts the counters a child advanced (words 2 and 4)
__global__ void rng_clone_kernel(uint32_t* dst, const uint32_t* src, uint32_t row_add, int counters_only) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (counters_only) {
        dst[2] = src[2];
        dst[4] = src[4];
        return;
    }
    for (int i = 0; i < DYF_RNG_STATE_WORDS; ++i) dst[i] = src[i];
    dst[3] = src[3] + row_add;
}

hipError_t launch_rng_clone(uint32_t* dst, const uint32_t* src, uint32_t row_add, bool counters_only, hipStream_t s) {
    hipLaunchKernelGGL(rng_clone_kernel, dim3(1), dim3(64), 0, s, dst, src, row_add, counters_only ? 1 : 0);
    return hipGetLastError();
}

hipError_t launch_noisy_condition(float* out, const float* cond, const float* noise, float tau, long long count,
                                  int row_elems, uint32_t* rng_state, hipStream_t s) {
    hipLaunchKernelGGL(noisy_condition_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, out, cond, noise,
                       tau, count, row_elems, rng_state);
    if (noise == nullptr)  // every call draws a fresh field (dyffusion.py:227 randn_like), also on graph replay
        hipLaunchKernelGGL(rng_bump_kernel, dim3(1), dim3(64), 0, s, rng_state, 4, 1u);
    return hipGetLastError();
}

// Start of a forward that draws dropout masks: fill the row-key table of the launch and advance the forward counter.
// Launch row r belongs to forward (counter + r / rows_per_fwd) and global batch row (row offset + r % rows_per_fwd): a
// paired interpolator launch (2 nb rows = two forwards of nb rows) draws exactly the masks of two separate forwards.
__global__ __launch_bounds__(256) void rng_begin_forward_kernel(uint32_t* state, uint32_t* row_keys, int rows, int rows_per_fwd) {
    const uint32_t lo = state[0], hi = state[1], fwd0 = state[2], row0 = state[3];
    for (int r = threadIdx.x; r < rows; r += blockDim.x) {
        const RngKey k = rng_row_key(lo, hi, fwd0 + (uint32_t)(r / rows_per_fwd), row0 + (uint32_t)(r % rows_per_fwd));
        row_keys[2 * r] = k.k0;
        row_keys[2 * r + 1] = k.k1;
    }
    __syncthreads();
    if (threadIdx.x == 0) state[2] = fwd0 + (uint32_t)((rows + rows_per_fwd - 1) / rows_per_fwd);
}

hipError_t launch_rng_begin_forward(uint32_t* rng_state, uint32_t* row_keys, int rows, int rows_per_fwd, hipStream_t s) {
    hipLaunchKernelGGL(rng_begin_forward_kernel, dim3(1), dim3(256), 0, s, rng_state, row_keys, rows, rows_per_fwd);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ boundary conditions
// PhysicalSystemsBenchmarkDataModule.boundary_conditions (src/datamodules/physical_systems_benchmark.py:245-297), applied by
// _evaluation_step to every predicted field (forecasting_multi_horizon.py:175-182): ONE masked write over the whole
// (fields, rows, C, H, W) stack instead of a Python loop over batch elements with boolean-mask index_put_ calls.
//   navier-stokes: preds[fixed_mask] = 0, then channel 0 / first grid row = parabolic inflow
//                  in_velocity * 4 * y * (0.41 - y) / 0.41^2 * (1 - exp(-5 t)), evaluated in fp32 in the reference's order
//   spring-mesh:   preds = where(fixed_mask, boundary, preds), boundary = cat[0 (p), q of the first time step]
// row_meta[row] is the batch element whose metadata applies to the row, -1 = row untouched (the reference's indexing of
// ensemble stacks is resolved on the host, dyffusion_amd/boundary.py).
__global__ void boundary_conditions_kernel(BcArgs a) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int chw = a.c * a.h * a.w;
    const long long total = (long long)a.n_fields * a.rows * chw;
    if (i >= total) return;
    const int e = (int)(i % chw);
    const int row = (int)((i / chw) % a.rows), f = (int)(i / ((long long)chw * a.rows));
    const int b = a.row_meta[row];
    if (b < 0) return;
    const bool fixed = a.fixed_mask[(size_t)b * chw + e] != 0;
    if (a.kind == 0) {
        if (e < a.w) {  // channel 0, grid row 0: left_boundary_indexing[0, 0, :]
            const float growth = a.time_factor[a.times_per_meta ? f * a.n_meta + b : f];  // (float)(1 - exp(-5 t)), host double math
            const float vy = a.vertex_y[(size_t)b * a.w + e];
            const float s0 = (float)((double)a.in_velocity[b] * 4.0);           // python float product, cast by the tensor op
            float v = s0 * vy;
            v = v * (0.41f - vy);
            v = v / (float)(0.41 * 0.41);
            a.preds[i] = v * growth;
        } else if (fixed) {
            a.preds[i] = 0.0f;
        }
    } else if (fixed) {
        a.preds[i] = a.boundary[(size_t)b * chw + e];
    }
}

hipError_t launch_boundary_conditions(const BcArgs& a, hipStream_t s) {
    const long long total = (long long)a.n_fields * a.rows * a.c * a.h * a.w;
    hipLaunchKernelGGL(boundary_conditions_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// test seam (dyf_debug_read_block_output): NHWC bf16 activation -> NCHW fp32; col_map (or null) maps a dense column to its
// column in a compact tensor of width w_store, -1 = not stored (NaN)
__global__ void nhwc_to_nchw_f32_kernel(const el16_t* src, int n, int h, int w, int w_store, int c, const int16_t* col_map,
                                        float* out) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)n * c * h * w;
    if (i >= total) return;
    const int x = (int)(i % w), y = (int)((i / w) % h), ch = (int)((i / ((long long)w * h)) % c), b = (int)(i / ((long long)w * h * c));
    const int xs = col_map ? col_map[x] : x;
    out[i] = xs < 0 ? __uint_as_float(0x7fc00000u) : el16_to_f32(src[(((size_t)b * h + y) * w_store + xs) * c + ch]);
}

hipError_t launch_nhwc_to_nchw_f32(const el16_t* src, int n, int h, int w, int w_store, int c, const int16_t* col_map,
                                   float* out, hipStream_t s) {
    const long long total = (long long)n * c * h * w;
    hipLaunchKernelGGL(nhwc_to_nchw_f32_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, src, n, h, w, w_store, c,
                       col_map, out);
    return hipGetLastError();
}

__global__ void fill_f32_kernel(float* p, float v, long long count) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) p[i] = v;
}

hipError_t launch_fill_f32(float* p, float v, long long count, hipStream_t s) {
    hipLaunchKernelGGL(fill_f32_kernel, dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s, p, v, count);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ ensemble metrics
// On-device form of src/utilities/evaluation.py:10-118 (evaluate_ensemble_prediction): for every grid point p of the
// (B, C, H, W) target the N ensemble members are reduced to
//   (mean_n x_n - y)^2                     -> "mse" of the ensemble mean            (evaluation.py:43-45)
//   var_n x_n  (population variance)       -> spread^2 of the spread-skill ratio    (evaluation.py:107-116)
//   mean_n |x_n - y| - 1/(2 N^2) sum_{n,m} |x_n - x_m|  -> CRPS of the empirical ensemble CDF (xskillscore/properscoring
//                                            crps_ensemble, evaluation.py:83-95)
// and summed over points in fp64 (one atomic per wave and quantity).  HBM-bound: every prediction is read exactly once
// (members staged in LDS, [N][256] floats); the O(N^2) pair term runs out of LDS.
__global__ __launch_bounds__(256) void ensemble_metrics_kernel(const float* preds, const float* targets, int n_members,
                                                               long long n_points, double* sums) {
    extern __shared__ float xs[];  // [n_members][256]
    const int tid = threadIdx.x;
    const long long p = (long long)blockIdx.x * 256 + tid;
    const bool live = p < n_points;
    float se = 0.0f, var = 0.0f, crps = 0.0f;
    if (live) {
        const float y = targets[p];
        float sum = 0.0f, sabs = 0.0f;
        for (int n = 0; n < n_members; ++n) {
            const float x = preds[(size_t)n * n_points + p];
            xs[n * 256 + tid] = x;
            sum += x;
            sabs += fabsf(x - y);
        }
        const float inv = 1.0f / (float)n_members;
        const float mean = sum * inv;
        float m2 = 0.0f, pair = 0.0f;
        for (int n = 0; n < n_members; ++n) {
            const float x = xs[n * 256 + tid];
            m2 += (x - mean) * (x - mean);
            for (int m = n + 1; m < n_members; ++m) pair += fabsf(x - xs[m * 256 + tid]);
        }
        se = (mean - y) * (mean - y);
        var = m2 * inv;
        crps = sabs * inv - pair * inv * inv;  // sum over ordered pairs = 2 * pair; 0.5 * 2 * pair / N^2
    }
    // wave reduction in fp64, then one atomic per wave
    double d0 = se, d1 = var, d2 = crps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_xor(d0, off, 64);
        d1 += __shfl_xor(d1, off, 64);
        d2 += __shfl_xor(d2, off, 64);
    }
    if ((tid & 63) == 0) {
        atomicAdd(sums + 0, d0);
        atomicAdd(sums + 1, d1);
        atomicAdd(sums + 2, d2);
    }
}

// More than 64 members (the reference has no cap, evaluation.py:10-80): the [N][256] staging tile no longer fits the 64 KB of LDS,
// so a workgroup takes P < 256 points (N * P floats staged) and T = 256 / P threads share a point: thread (point pi = tid % P,
// share sh = tid / P) owns the members n = sh, sh + T, ...; the shares' partial sums meet in LDS and are added in share order.
// The O(N^2) pair term is then spread over all 256 threads.  Same definitions as ensemble_metrics_kernel.
__global__ __launch_bounds__(256) void ensemble_metrics_tiled_kernel(const float* preds, const float* targets, int n_members,
                                                                     long long n_points, double* sums, int P) {
    extern __shared__ float xs[];        // [n_members][P] members, then [3][256] partials
    float* red = xs + (size_t)n_members * P;
    const int tid = threadIdx.x, T = 256 / P;
    const int pi = tid % P, sh = tid / P;
    const long long p0 = (long long)blockIdx.x * P, p = p0 + pi;
    const bool live = p < n_points;
    for (int i = tid; i < n_members * P; i += 256) {  // staging: consecutive threads -> consecutive points of one member
        const int n = i / P, q = i - n * P;
        xs[i] = p0 + q < n_points ? preds[(size_t)n * n_points + p0 + q] : 0.0f;
    }
    __syncthreads();
    const float y = live ? targets[p] : 0.0f;
    float sum = 0.0f, sabs = 0.0f;
    for (int n = sh; n < n_members; n += T) {
        const float x = xs[n * P + pi];
        sum += x;
        sabs += fabsf(x - y);
    }
    red[tid] = sum;
    red[256 + tid] = sabs;
    __syncthreads();
    float tsum = 0.0f, tabs = 0.0f;
    for (int k = 0; k < T; ++k) {  // every share adds the T partials in the same order: all hold the same mean
        tsum += red[k * P + pi];
        tabs += red[256 + k * P + pi];
    }
    const float inv = 1.0f / (float)n_members;
    const float mean = tsum * inv;
    __syncthreads();
    float m2 = 0.0f;
    double pair = 0.0;  // up to N^2 / 2 terms per point: rows in fp32, the sum of rows in fp64
    for (int n = sh; n < n_members; n += T) {
        const float x = xs[n * P + pi];
        m2 += (x - mean) * (x - mean);
        float row = 0.0f;
        for (int m = n + 1; m < n_members; ++m) row += fabsf(x - xs[m * P + pi]);
        pair += (double)row;
    }
    red[tid] = m2;
    red[256 + tid] = (float)pair;
    __syncthreads();
    float se = 0.0f, var = 0.0f, crps = 0.0f;
    if (live && sh == 0) {
        float tm2 = 0.0f, tpair = 0.0f;
        for (int k = 0; k < T; ++k) {
            tm2 += red[k * P + pi];
            tpair += red[256 + k * P + pi];
        }
        se = (mean - y) * (mean - y);
        var = tm2 * inv;
        crps = tabs * inv - tpair * inv * inv;
    }
    double d0 = se, d1 = var, d2 = crps;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        d0 += __shfl_xor(d0, off, 64);
        d1 += __shfl_xor(d1, off, 64);
        d2 += __shfl_xor(d2, off, 64);
    }
    if ((tid & 63) == 0) {
        atomicAdd(sums + 0, d0);
        atomicAdd(sums + 1, d1);
        atomicAdd(sums + 2, d2);
    }
}

// Reduction of the training criterion (src/utilities/utils.py:201-212, reduction = "mean"): sum over all elements of
// |p - t| (kind 0), (p - t)^2 (kind 1) or smooth-L1 with beta = 1 (kind 2); fp32 per lane, wave butterfly, one fp64 atomic
// per wave.  HBM-bound: both tensors are read once with 16-byte loads.
__global__ __launch_bounds__(256) void criterion_sum_kernel(const float* p, const float* t, long long count, int kind, double* sum) {
    const long long n4 = count >> 2;
    float acc = 0.0f;
    auto term = [&](float d) {
        const float ad = fabsf(d);
        return kind == 0 ? ad : kind == 1 ? d * d : (ad < 1.0f ? 0.5f * d * d : ad - 0.5f);
    };
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        const float4 a = ((const float4*)p)[i], b = ((const float4*)t)[i];
        acc += term(a.x - b.x) + term(a.y - b.y) + term(a.z - b.z) + term(a.w - b.w);
    }
    if (blockIdx.x == 0 && threadIdx.x < (count & 3)) acc += term(p[n4 * 4 + threadIdx.x] - t[n4 * 4 + threadIdx.x]);
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) acc += __shfl_xor(acc, d, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(sum, (double)acc);
}

hipError_t launch_criterion_sum(const float* p, const float* t, long long count, int kind, double* sum, hipStream_t s) {
    hipError_t e = hipMemsetAsync(sum, 0, sizeof(double), s);
    if (e != hipSuccess) return e;
    const unsigned grid = (unsigned)std::max<long long>(1, std::min<long long>((count / 4 + 1023) / 1024, 2048));
    hipLaunchKernelGGL(criterion_sum_kernel, dim3(grid), dim3(256), 0, s, p, t, count, kind, sum);
    return hipGetLastError();
}

hipError_t launch_ensemble_metrics(const float* preds, const float* targets, int n_members, long long n_points, double* sums,
                                   hipStream_t s) {
    const size_t lds = (size_t)n_members * 256 * sizeof(float);
    hipError_t e = hipMemsetAsync(sums, 0, 3 * sizeof(double), s);
    if (e != hipSuccess) return e;
    if (n_members > 64) {  // points per workgroup: the largest power of two with n_members * P floats <= 60 KB
        int P = 128;
        while (P > 1 && (size_t)n_members * P * sizeof(float) > 60 * 1024) P >>= 1;
        if ((size_t)n_members * P * sizeof(float) > 60 * 1024) return hipErrorInvalidValue;  // > 15 360 members
        const size_t lds2 = ((size_t)n_members * P + 2 * 256) * sizeof(float);
        hipLaunchKernelGGL(ensemble_metrics_tiled_kernel, dim3((unsigned)((n_points + P - 1) / P)), dim3(256), lds2, s, preds, targets,
                           n_members, n_points, sums, P);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(ensemble_metrics_kernel, dim3((unsigned)((n_points + 255) / 256)), dim3(256), lds, s, preds, targets,
                       n_members, n_points, sums);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ exchange unpack
__global__ __launch_bounds__(256) void gather_unpack_kernel(const float4* recv, float4* out, int world, int slots, int nb, int total_rows,
                                                            long long row4) {
    const int g = blockIdx.y;                 // global row
    const int slot = blockIdx.z;
    const int base = total_rows / world, extra = total_rows % world;
    // owner of global row g under the balanced contiguous split: ranks < extra own base + 1 rows
    const int cut = extra * (base + 1);
    const int r = g < cut ? g / (base + 1) : extra + (g - cut) / max(base, 1);
    const int j = g < cut ? g - r * (base + 1) : (g - cut) - (r - extra) * base;
    const float4* src = recv + (((size_t)r * slots + slot) * nb + j) * row4;
    float4* dst = out + ((size_t)slot * total_rows + g) * row4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < row4; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

__global__ __launch_bounds__(256) void gather_unpack_scalar_kernel(const float* recv, float* out, int world, int slots, int nb,
                                                                   int total_rows, long long row) {
    const int g = blockIdx.y, slot = blockIdx.z;
    const int base = total_rows / world, extra = total_rows % world, cut = extra * (base + 1);
    const int r = g < cut ? g / (base + 1) : extra + (g - cut) / max(base, 1);
    const int j = g < cut ? g - r * (base + 1) : (g - cut) - (r - extra) * base;
    const float* src = recv + (((size_t)r * slots + slot) * nb + j) * row;
    float* dst = out + ((size_t)slot * total_rows + g) * row;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < row; i += (long long)gridDim.x * 256) dst[i] = src[i];
}

hipError_t launch_gather_unpack(const float* recv, float* out, int world, int slots, int nb, int total_rows, long long row, hipStream_t s) {
    const bool vec = row % 4 == 0 && ((uintptr_t)recv & 15) == 0 && ((uintptr_t)out & 15) == 0;
    const long long units = vec ? row / 4 : row;
    const dim3 grid((unsigned)std::max<long long>(1, std::min<long long>((units + 1023) / 1024, 64)), (unsigned)total_rows, (unsigned)slots);
    if (vec)
        hipLaunchKernelGGL(gather_unpack_kernel, grid, dim3(256), 0, s, (const float4*)recv, (float4*)out, world, slots, nb, total_rows, units);
    else
        hipLaunchKernelGGL(gather_unpack_scalar_kernel, grid, dim3(256), 0, s, recv, out, world, slots, nb, total_rows, row);
    return hipGetLastError();
}
