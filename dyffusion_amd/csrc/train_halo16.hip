// 3 x 3 / stride 1 / pad 1 convolutions of the training step with 16-bit operands (DYF_TRAIN_OPERANDS=bf16): forward, data
// gradient and weight gradient on fp32 NHWC tensors, the operands rounded to bf16 (in both builds of the library) while they are staged
// (the reference gets these from torch.autograd over src/models/unet_simple.py:29-56 and src/models/unet.py:58-109).
//
// Why a second form next to train_gemm.hip: its implicit GEMM gathers the A operand tap by tap, so a 3 x 3 conv reads every
// fp32 activation nine times through L1 / L2 (the 256 x 256 x 64-channel layers at B = 32: 1.07 GB x 9 per launch, 2.6 ms forward
// and data gradient, 3.7 ms weight gradient -- 84-115 TFLOP/s), and its weight gradient runs one tap per workgroup.  Here a
// workgroup owns an 8 x 16-pixel tile, stages the tile and its one-pixel halo ONCE (fp32 -> 16 bit on the way into LDS) and runs
// all nine taps from LDS: the activation traffic falls to 1.4 x the tensor (the halo overlap), which makes these launches
// HBM-bound on their fp32 operands instead of L2-bound on the re-reads.
//
//   t_halo3x3_16   C[p][n] = bias[n] + sum_{tap, k} A[p + d(tap)][k] * Wb[n][tap][k]        forward (A = x, n = co, k = ci) and
//                  data gradient (A = dz, n = ci, k = co, taps mirrored) are the same kernel; Wb is the weight tensor converted
//                  to 16 bit in that order by t_pack_w16 (per launch: the weights change every step)
//   t_wgrad3x3_16  dW[co][tap][ci] += sum_p dz[p][co] * x[p + d(tap)][ci]                  the contraction runs over pixels, so
//                  both operands are transposed on the way into LDS ([channel][pixel], 8 pixels = one 16-byte fragment); the
//                  tap's column shift of +-1 pixel is a 2-byte shift of the fragment, made in registers (v_alignbyte) from
//                  the aligned fragment and its neighbour; nine accumulators (one per tap) per wave
#include "train_internal.h"

#include <cstdlib>
#include <cstring>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int TH = 8, TW = 16;            // pixel tile: 8 rows x 16 columns = 128 pixels = four 32-pixel MFMA row blocks
constexpr int HW = TW + 2, HP = (TH + 2) * HW;  // halo: 10 x 18 = 180 pixels

// swizzle key of halo pixel hp (XORed into the 16-byte chunk index of its 128-byte row): conflict-free for the fragment reads
// of 2 tile rows x 16 columns at every tap displacement (conv_up_halo.hip HKEY, HALO_W = 18)
__device__ __forceinline__ int hkey(int hp) { return ((hp >> 1) - (int)((unsigned)hp / (unsigned)HW)) & 7; }

// ---------------------------------------------------------------------------------------------- weights -> 16 bit, [n][tap][k]
// mode 0 (forward):  src = wt[tap][ci][co]            -> dst[co][tap][ci]
// mode 1 (dgrad):    src = w[co][tap][ci], mirrored   -> dst[ci][8 - tap][co]
__global__ void t_pack_w16(const float* __restrict__ src, int cin, int cout, int mode, t16_t* __restrict__ dst) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = (long long)cin * cout * 9;
    if (i >= total) return;
    const int CK = mode ? cout : cin;
    const int k = (int)(i % CK);
    const int tap = (int)((i / CK) % 9);
    const int n = (int)(i / ((long long)CK * 9));
    float v;
    if (mode == 0) v = src[((size_t)tap * cin + k) * cout + n];
    else v = src[((size_t)k * 9 + (8 - tap)) * cin + n];
    dst[i] = f32_to_t16(v);
}

// ---------------------------------------------------------------------------------------------- forward / data gradient
// A workgroup walks tiles_per_wg consecutive tiles of one column block as ONE stream of (tile, 64-channel chunk, tap) steps:
//   weights   step g's [BN][64] image is DMA'd (buffer_load ... lds, no registers) into slot g % 4 of a ring, three steps ahead;
//             lane (row r, slot s) of a DMA instruction fetches chunk s ^ key(row), so the linear 1 KB it writes is the swizzled image
//   halo      the NEXT chunk's (or next tile's) halo is requested into registers at the first tap of a chunk and rounded into the
//             other halo buffer after the chunk's ninth tap
//   one barrier per step: it publishes the step's weights (every wave waits for its own DMA pieces first) and frees the slot of
//   the step before.  vmcnt: loads return in order, so "at most 2 stages outstanding" (the two requested after the one awaited)
//   is always enough -- halo loads and the epilogue's stores in between only make the wait longer, never too short.
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int BN>
__global__ __launch_bounds__(256, BN == 64 ? 2 : 1) void t_halo3x3_16(int h, int w, int CK, int NC, const float* __restrict__ A,
                                                                      const t16_t* __restrict__ Wb, const float* __restrict__ bias,
                                                                      float* __restrict__ C, int tiles_x, int tiles_per_img,
                                                                      int tiles_total, int tiles_per_wg, int nblocks) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int NT = BN / 32, RING = 4, BI = BN / 32;  // BI: DMA instructions per wave and step (BN rows x 128 B over 4 waves)
    constexpr int XS_BYTES = HP * 128 + 512, B_BYTES = BN * 128;
    constexpr int HV = (HP * 16 + 255) / 256;  // 12 halo float4 per thread (the last one: wave 0 only)
    extern __shared__ __attribute__((aligned(16))) char smem[];  // 2 halo buffers + the weight ring
    char* const Bring = smem + 2 * XS_BYTES;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // XCD-aware order (workgroup id % 8 = XCD, each with its own L2): the column blocks of one tile range follow one another on the
    // SAME XCD, so all but the first find the fp32 halo in that L2 -- with the blocks on neighbouring ids (= 8 different L2s) the PMC
    // counted the A operand NC / 64 times (16 x 256^2 x 64 -> 128 data gradient: 727 MB fetched for 376 MB of halo)
    const int xq = blockIdx.x >> 3;
    const int nb = xq % nblocks, wg = (xq / nblocks) * 8 + (int)(blockIdx.x & 7), n0 = nb * BN;
    const int t_beg = wg * tiles_per_wg, t_end = min(t_beg + tiles_per_wg, tiles_total);
    if (t_beg >= tiles_total) return;  // (the grid is rounded up to whole groups of 8 tile ranges)
    const int nchunks = CK >> 6;
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)Wb, 0, (int)(unsigned)((size_t)NC * 9 * CK * 2), 0x00020000);

    // weight DMA: instruction j of this wave fills rows (wave * BI + j) * 8 ... + 8 of the step's image
    unsigned w_voff[BI];
#pragma unroll
    for (int j = 0; j < BI; ++j) {
        const int n = (wave * BI + j) * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((n >> 1) & 7);
        w_voff[j] = (unsigned)((size_t)(n0 + n) * 9 * CK + c * 8) * 2u;
    }
    int is_tap = 0, is_chunk = 0, is_slot = 0;  // the step the next issue_b requests (runs past the end: harmless re-fetches)
    auto issue_b = [&]() {
        const unsigned soff = (unsigned)(is_tap * CK + (is_chunk << 6)) * 2u;
#pragma unroll
        for (int j = 0; j < BI; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_w, LDS_PTR(Bring + is_slot * B_BYTES + (wave * BI + j) * 1024), 16, w_voff[j], soff, 0, 0);
        if (++is_tap == 9) { is_tap = 0; if (++is_chunk == nchunks) is_chunk = 0; }
        if (++is_slot == RING) is_slot = 0;
    };

    float4 hv[HV];
    auto halo_load = [&](int tile, int chunk) {
        const int img = tile / tiles_per_img, t_in = tile - img * tiles_per_img;
        const int y0 = (t_in / tiles_x) * TH, x0 = (t_in % tiles_x) * TW;
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int idx = tid + 256 * j;
            const int hp = min(idx >> 4, HP - 1), q = idx & 15;
            const int hy = (int)((unsigned)hp / (unsigned)HW), hx = hp - hy * HW;
            const int yy = y0 - 1 + hy, xx = x0 - 1 + hx;
            hv[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w)
                hv[j] = *(const float4*)(A + (((size_t)img * h + yy) * w + xx) * CK + (chunk << 6) + q * 4);
        }
    };
    auto halo_store = [&](int buf) {
        char* Xs = smem + buf * XS_BYTES;
#pragma unroll
        for (int j = 0; j < HV; ++j) {
            const int idx = tid + 256 * j;
            const int hp = idx >> 4, q = idx & 15;
            if (idx < HP * 16)
                *(uint2*)(Xs + hp * 128 + (((q >> 1) ^ hkey(hp)) << 4) + (q & 1) * 8) =
                    make_uint2(pack_t16x2(hv[j].x, hv[j].y), pack_t16x2(hv[j].z, hv[j].w));
        }
    };

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // this lane's pixel of the wave's 32-pixel block: tile rows 2 * wave + {0, 1}, 16 columns
    const int hp00 = (2 * wave + (l31 >> 4)) * HW + (l31 & 15);  // halo pixel at tap (0, 0)

    halo_load(t_beg, 0);
#pragma unroll
    for (int i = 0; i < RING - 1; ++i) issue_b();
    halo_store(0);
    int xbuf = 0, slot = 0;
    for (int tile = t_beg; tile < t_end; ++tile) {
        for (int chunk = 0; chunk < nchunks; ++chunk) {
            const bool last = chunk + 1 == nchunks;
            const bool has_next = !last || tile + 1 < t_end;
            const char* Xs = smem + xbuf * XS_BYTES;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                // (a raw s_barrier: __syncthreads() is a fence, and the compiler drains vmcnt to 0 in front of it -- the whole ring)
                asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"((RING - 2) * BI) : "memory");
                __builtin_amdgcn_s_barrier();
                __builtin_amdgcn_sched_barrier(0);
                if (tap == 0 && has_next) halo_load(last ? tile + 1 : tile, last ? 0 : chunk + 1);
                issue_b();
                const int hp = hp00 + (tap / 3) * HW + (tap % 3);
                const char* ap = Xs + hp * 128;
                const int akey = hkey(hp);
                const char* bp = Bring + slot * B_BYTES;
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const int c16 = ks * 2 + hi;
                    const t16x8_t a = *(const t16x8_t*)(ap + ((c16 ^ akey) << 4));
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int n = t * 32 + l31;
                        const t16x8_t b = *(const t16x8_t*)(bp + n * 128 + ((c16 ^ ((n >> 1) & 7)) << 4));
                        acc[t] = T16_MFMA_32x32x16(a, b, acc[t], 0, 0, 0);
                    }
                }
                if (++slot == RING) slot = 0;
            }
            // the other halo buffer was last read a whole chunk (nine barriers) ago
            if (has_next) halo_store(xbuf ^ 1);
            xbuf ^= 1;
        }
        // D[i][j]: lane j = l31 holds output channel n0 + 32 t + l31, register r the pixel i = 8 (r >> 2) + 4 hi + (r & 3) of the block
        const int img = tile / tiles_per_img, t_in = tile - img * tiles_per_img;
        const int y0 = (t_in / tiles_x) * TH, x0 = (t_in % tiles_x) * TW;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int n = n0 + t * 32 + l31;
            const float bv = bias ? bias[n] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = 8 * (r >> 2) + 4 * hi + (r & 3);
                const int y = y0 + 2 * wave + (i >> 4), x = x0 + (i & 15);
                if (y < h && x < w) C[(((size_t)img * h + y) * w + x) * NC + n] = acc[t][r] + bv;
                acc[t][r] = 0.0f;
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the look-ahead DMA of steps past the end must land before the LDS is released
#endif
}

// ---------------------------------------------------------------------------------------------- weight gradient
// workgroup = (pixel-tile range, 64 output channels, 64 input channels); wave (wm, wn) owns the 32 x 32 block (co, ci) of all nine taps
constexpr int DZ_PITCH = 256;   // [co][8 rows x 16 columns] 16-bit: 16 chunks of 8 pixels
constexpr int XT_PITCH = 768;   // [ci][10 halo rows x 4 blocks of 8 columns (x0 - 8 ... x0 + 23)]: 40 chunks, pitch 48 (swizzle room)

__global__ __launch_bounds__(256, 2) void t_wgrad3x3_16(int h, int w, int cin, int cout, const float* __restrict__ dz,
                                                        const float* __restrict__ x, float* __restrict__ dw, int tiles_x,
                                                        int tiles_per_img, int tiles_total, int tiles_per_wg) {
#if defined(__HIP_DEVICE_COMPILE__)
    __shared__ __attribute__((aligned(16))) char smem[64 * DZ_PITCH + 64 * XT_PITCH];
    char* Dz = smem;
    char* Xt = smem + 64 * DZ_PITCH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, hi = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int co0 = blockIdx.y * 64, ci0 = blockIdx.z * 64;
    const int t_beg = blockIdx.x * tiles_per_wg, t_end = min(t_beg + tiles_per_wg, tiles_total);

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int cq = tid & 15;  // channel quad of this thread in every staging task
    const int a_row = wm * 32 + l31, b_row = wn * 32 + l31;
    const char* a_base = Dz + a_row * DZ_PITCH;
    const char* b_base = Xt + b_row * XT_PITCH;
    const int a_key = a_row & 15, b_key = b_row & 15;

    for (int tile = t_beg; tile < t_end; ++tile) {
        const int img = tile / tiles_per_img, t_in = tile - img * tiles_per_img;
        const int y0 = (t_in / tiles_x) * TH, x0 = (t_in % tiles_x) * TW;
        __syncthreads();  // every wave is done with the previous tile's images
        {   // dz: chunk pg = (row y, column half xh): 8 pixels x 4 channels per thread, transposed into [co][pixel]
            const int pg = tid >> 4, y = pg >> 1, xh = pg & 1;
            const int yy = y0 + y;
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xx = x0 + xh * 8 + j;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (yy < h && xx < w) v[j] = *(const float4*)(dz + (((size_t)img * h + yy) * w + xx) * cout + co0 + cq * 4);
            }
#define PACK8(F) u32x4{pack_t16x2(v[0].F, v[1].F), pack_t16x2(v[2].F, v[3].F), pack_t16x2(v[4].F, v[5].F), pack_t16x2(v[6].F, v[7].F)}
            const int r0 = cq * 4;
            *(u32x4*)(Dz + (r0 + 0) * DZ_PITCH + ((pg ^ ((r0 + 0) & 15)) << 4)) = PACK8(x);
            *(u32x4*)(Dz + (r0 + 1) * DZ_PITCH + ((pg ^ ((r0 + 1) & 15)) << 4)) = PACK8(y);
            *(u32x4*)(Dz + (r0 + 2) * DZ_PITCH + ((pg ^ ((r0 + 2) & 15)) << 4)) = PACK8(z);
            *(u32x4*)(Dz + (r0 + 3) * DZ_PITCH + ((pg ^ ((r0 + 3) & 15)) << 4)) = PACK8(w);
        }
        // x, the two full blocks of every halo row (columns x0 ... x0 + 15): 20 chunks x 16 channel quads = 320 tasks
#pragma unroll 1
        for (int tk = tid; tk < 320; tk += 256) {
            const int cb = tk >> 4, hy = cb >> 1, blk = (cb & 1) + 1;
            const int yy = y0 - 1 + hy;
            float4 v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int xx = x0 + (blk - 1) * 8 + j;
                v[j] = make_float4(0.f, 0.f, 0.f, 0.f);
                if ((unsigned)yy < (unsigned)h && xx < w) v[j] = *(const float4*)(x + (((size_t)img * h + yy) * w + xx) * cin + ci0 + cq * 4);
            }
            const int r0 = cq * 4, ch = hy * 4 + blk;
            *(u32x4*)(Xt + (r0 + 0) * XT_PITCH + ((ch ^ ((r0 + 0) & 15)) << 4)) = PACK8(x);
            *(u32x4*)(Xt + (r0 + 1) * XT_PITCH + ((ch ^ ((r0 + 1) & 15)) << 4)) = PACK8(y);
            *(u32x4*)(Xt + (r0 + 2) * XT_PITCH + ((ch ^ ((r0 + 2) & 15)) << 4)) = PACK8(z);
            *(u32x4*)(Xt + (r0 + 3) * XT_PITCH + ((ch ^ ((r0 + 3) & 15)) << 4)) = PACK8(w);
        }
#undef PACK8
        // x, the edge columns x0 - 1 (last element of block 0) and x0 + 16 (first element of block 3): 20 x 16 single-pixel tasks
#pragma unroll 1
        for (int tk = tid; tk < 320; tk += 256) {
            const int ce = tk >> 4, hy = ce >> 1, side = ce & 1;
            const int yy = y0 - 1 + hy, xx = side ? x0 + TW : x0 - 1;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if ((unsigned)yy < (unsigned)h && (unsigned)xx < (unsigned)w) v = *(const float4*)(x + (((size_t)img * h + yy) * w + xx) * cin + ci0 + cq * 4);
            const int r0 = cq * 4, ch = hy * 4 + (side ? 3 : 0), eo = side ? 0 : 14;
            *(t16_t*)(Xt + (r0 + 0) * XT_PITCH + ((ch ^ ((r0 + 0) & 15)) << 4) + eo) = f32_to_t16(v.x);
            *(t16_t*)(Xt + (r0 + 1) * XT_PITCH + ((ch ^ ((r0 + 1) & 15)) << 4) + eo) = f32_to_t16(v.y);
            *(t16_t*)(Xt + (r0 + 2) * XT_PITCH + ((ch ^ ((r0 + 2) & 15)) << 4) + eo) = f32_to_t16(v.z);
            *(t16_t*)(Xt + (r0 + 3) * XT_PITCH + ((ch ^ ((r0 + 3) & 15)) << 4) + eo) = f32_to_t16(v.w);
        }
        __syncthreads();
        // 8 k-steps of 16 pixels (tile row y; lanes hi = 0 / 1 take columns 0-7 / 8-15)
#pragma unroll 2
        for (int y = 0; y < TH; ++y) {
            const t16x8_t a = *(const t16x8_t*)(a_base + (((y * 2 + hi) ^ a_key) << 4));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int ch = (y + ky) * 4 + hi;  // block holding columns [8 hi - 8, 8 hi) of halo row y + ky
                const u32x4 pv = *(const u32x4*)(b_base + (((ch + 0) ^ b_key) << 4));
                const u32x4 cu = *(const u32x4*)(b_base + (((ch + 1) ^ b_key) << 4));
                const u32x4 nx = *(const u32x4*)(b_base + (((ch + 2) ^ b_key) << 4));
                const uint32_t s01 = __builtin_amdgcn_alignbyte(cu.y, cu.x, 2), s12 = __builtin_amdgcn_alignbyte(cu.z, cu.y, 2),
                               s23 = __builtin_amdgcn_alignbyte(cu.w, cu.z, 2);
                const u32x4 left = {__builtin_amdgcn_alignbyte(cu.x, pv.w, 2), s01, s12, s23};    // pixels shifted by -1 column
                const u32x4 right = {s01, s12, s23, __builtin_amdgcn_alignbyte(nx.x, cu.w, 2)};   // pixels shifted by +1 column
                acc[ky * 3 + 0] = T16_MFMA_32x32x16(a, __builtin_bit_cast(t16x8_t, left), acc[ky * 3 + 0], 0, 0, 0);
                acc[ky * 3 + 1] = T16_MFMA_32x32x16(a, __builtin_bit_cast(t16x8_t, cu), acc[ky * 3 + 1], 0, 0, 0);
                acc[ky * 3 + 2] = T16_MFMA_32x32x16(a, __builtin_bit_cast(t16x8_t, right), acc[ky * 3 + 2], 0, 0, 0);
            }
        }
    }
    // D[i][j]: lane j = l31 -> ci, register r -> co = 8 (r >> 2) + 4 hi + (r & 3); merged over the pixel ranges with atomics
    // (as the one-tap-per-workgroup kernel this replaces)
    const int ci = ci0 + wn * 32 + l31;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = co0 + wm * 32 + 8 * (r >> 2) + 4 * hi + (r & 3);
            atomicAdd(dw + ((size_t)co * 9 + tap) * cin + ci, acc[tap][r]);
        }
#endif
}

bool halo16_enabled() {
    const char* v = dyf_form("DYF_TRAIN_HALO16");  // =0: the tap-by-tap implicit GEMM of train_gemm.hip for these layers too (A/B)
    return !(v && atoi(v) == 0);
}

}  // namespace

namespace dyf {

// forward (mode 0: A = x, W = wt[tap][ci][co]) / data gradient (mode 1: A = dz, W = w[co][tap][ci]); ws holds the 16-bit weights
bool thalo_conv3x3(const TConv& g, int mode, const float* A, const float* W, const float* bias, float* C, float* ws, size_t ws_floats,
                   hipStream_t st) {
    if (!halo16_enabled() || g.k != 3 || g.s != 1 || g.p != 1 || g.ho != g.h || g.wo != g.w) return false;
    const int CK = mode ? g.cout : g.cin, NC = mode ? g.cin : g.cout;
    if (CK % 64 != 0 || NC % 64 != 0) return false;
    const size_t welems = (size_t)CK * NC * 9;
    if (ws == nullptr || welems > ws_floats * 2) return false;
    const int tiles_x = (g.w + TW - 1) / TW, tiles_per_img = tiles_x * ((g.h + TH - 1) / TH);
    const long long tiles = (long long)g.n * tiles_per_img;
    // 64-channel column blocks (two workgroups per CU) also where 128 divides: NS B=32 step 110 -> 109 ms, ResNet-UNet B=64 144 -> 141 ms;
    // the column blocks of a tile are neighbours in the grid (its halo is read from L2 by all but the first)
    constexpr int bn = 64;
    if (tiles * (NC / bn) < 192 || tiles > 0x7fffffffll) return false;  // small planes: the split-K forms of train_gemm.hip  // small planes: the split-K forms of train_gemm.hip
    t16_t* wb = (t16_t*)ws;
    dyf_form_note(mode ? "t_halo3x3_16:dgrad" : "t_halo3x3_16:forward", g.n);
    hipLaunchKernelGGL(t_pack_w16, dim3((unsigned)((welems + 255) / 256)), dim3(256), 0, st, W, g.cin, g.cout, mode, wb);
    const int nblocks = NC / bn;
    // a workgroup streams several tiles (prefetch runs across tile boundaries)
    // (measured, us per launch with ~512 / 1 024 / 2 048 / 4 096 workgroups: 64 x 60^2 x 64 -> 64 49 / 52 / 55 / 56, 64 x 30^2 x 256 -> 128 data
    // gradient 66 / 69 / 72 / 72, 16 x 256^2 x 64 -> 128 data gradient 385 / 384 / 377 / 379: one resident round for the smaller launches)
    const long long htarget = tiles * nblocks <= 8192 ? 512 : 2048;
    const int per = (int)std::max<long long>(1, std::min<long long>(16, tiles * nblocks / htarget));
    const long long wgs = ((tiles + per - 1) / per + 7) / 8 * 8 * nblocks;  // whole groups of 8 tile ranges (XCD-aware order in the kernel)
    constexpr int XS = HP * 128 + 512;
    if (!train_raise_dynamic_lds(t_halo3x3_16<64>, 2 * XS + 4 * 64 * 128)) return false;  // per device; the tap-by-tap form takes the layer
    hipLaunchKernelGGL(t_halo3x3_16<64>, dim3((unsigned)wgs), dim3(256), 2 * XS + 4 * 64 * 128, st, g.h, g.w, CK, NC, A, wb, bias, C, tiles_x,
                       tiles_per_img, (int)tiles, per, nblocks);
    return true;
}

// dw += the weight gradient (all nine taps per workgroup)
bool thalo_wgrad3x3(const TConv& g, const float* dz, const float* x, float* dw, hipStream_t st) {
    if (!halo16_enabled() || g.k != 3 || g.s != 1 || g.p != 1 || g.ho != g.h || g.wo != g.w) return false;
    if (g.cin % 64 != 0 || g.cout % 64 != 0) return false;
    const int tiles_x = (g.w + TW - 1) / TW, tiles_per_img = tiles_x * ((g.h + TH - 1) / TH);
    const long long tiles = (long long)g.n * tiles_per_img;
    const int pairs = (g.cout / 64) * (g.cin / 64);
    if (tiles < 64 || tiles > 0x7fffffffll || g.cin / 64 > 65535 || g.cout / 64 > 65535) return false;
    // one resident round (2 workgroups x 256 CUs): every workgroup ends with 64 x 64 x 9 fp32 atomics, so more, shorter workgroups cost
    // more than they balance -- us per launch with 256 / 512 / 1 024 / 2 048 workgroups: 16 x 256^2 x 128 -> 64 439 / 314 / 383 / 430,
    // 64 x 60^2 x 128 -> 128 201 / 173 / 234 / 327, 32 x 64^2 x 256 -> 256 361 / 250 / 298 / 379
    long long splits = std::max<long long>(1, std::min<long long>(tiles / 4, (512 + pairs - 1) / pairs));
    const int per = (int)((tiles + splits - 1) / splits);
    splits = (tiles + per - 1) / per;
    dyf_form_note("t_wgrad3x3_16", g.n);
    hipLaunchKernelGGL(t_wgrad3x3_16, dim3((unsigned)splits, g.cout / 64, g.cin / 64), dim3(256), 0, st, g.h, g.w, g.cin, g.cout, dz, x, dw,
                       tiles_x, tiles_per_img, (int)tiles, per);
    return true;
}

}  // namespace dyf
