// Implicit-GEMM Conv2d, second form (gfx950): 256 pixels x 128 output channels per workgroup, weights streamed from
// global memory in MFMA fragment order.  Used for every conv with cout % 128 == 0 that is not a fused-upsample conv
// (encoder 4x4/s2 blocks, decoder 3x3 blocks on small planes, the ResNet-UNet 3x3 convs).
//
// conv_igemm_kernel (conv.hip) stages BOTH operands through LDS with 64x64 wave tiles: at MFMA peak its four waves
// would read 128 B/clk from LDS, i.e. all of the CU's LDS bandwidth, so it tops out near 45 % MFMA utilisation.  Here
//   * a wave owns 128 pixels x 64 channels (4 x 2 accumulator tiles, 128 registers): every pixel fragment read from
//     LDS feeds 2 MFMAs and every weight fragment 4, LDS traffic drops to 64 B/clk/CU;
//   * the weights never touch LDS: pack_conv_frag() lays them out so that one wave-wide buffer_load_dwordx4 fetches one
//     32-channel x 16-k fragment as 1 KB of contiguous memory, four fragment sets in flight 3 k16 sub-steps ahead;
//   * the pixel operand keeps the LDS-DMA gather of conv.hip (one 128-B row per pixel and K step, zero fill for padded
//     taps through the buffer bounds check), double-buffered, one barrier per K step; two workgroups per CU cover each
//     other's barrier and epilogue;
//   * operands are swapped (D^T = W X^T) so the epilogue runs straight out of the accumulators (see conv.hip).
// In-order completion of vector-memory loads makes the per-step wait free: the gather of step s+1 is issued BEFORE the
// weight loads of step s, so by the time the last of those has been consumed the gather has landed.
#include "conv.h"
#include "gn_fused.h"

#include <cstdlib>
#include <mutex>
#include <type_traits>
#include <unordered_map>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {
constexpr int BM = 256;
constexpr int A_BYTES = BM * 128;      // one K step of the pixel operand: 256 rows x 64 channels bf16
constexpr int OROW = 144;              // output staging: [16 pixels][64 channels] 16-bit + 16 B pad per pixel and wave
constexpr int LDS_TOTAL = 2 * A_BYTES + 4 * 16 * OROW; // 64 KB + 9 KB staging: two workgroups per CU
constexpr int LDS_TOTAL_SH3 = 2 * (BM + 3) * 128 + 4 * 16 * OROW;  // SH3 form: + 3 rows per stage (75 520 B; + 4 KB fused GroupNorm)
constexpr int RA = BM / 32;            // rows gathered per lane per K step
constexpr int TH = BM / 16;            // 2-D tile: 16 rows of 16 pixels
}  // namespace

// WN = 2: 256 pixels x 128 channels per workgroup, waves 2 x 2, each 128 pixels x 64 channels (4 x 2 accumulator tiles).
// WN = 1: 256 pixels x 64 channels for layers whose output channels are a multiple of 64 only (the 64-channel level of the
//         ResNet-UNet): waves 4 x 1, each 64 pixels x 64 channels (2 x 2 tiles).  Pixel fragments still feed 2 MFMAs each
//         (64 B/clk/CU of LDS reads at MFMA peak); every weight fragment feeds 2 instead of 4, i.e. the weight stream costs
//         64 B/clk/CU of L1 bandwidth at peak -- its limit, fine at the ~50 % this kernel reaches (conv_igemm_kernel<256, 64>
//         stages both operands through LDS and stops at 19 %).
// GNF (WN = 2): GroupNorm fused into the epilogue (gn_fused.h; ConvArgs::gnf).  A wave's slab of 128 output rows lies in at most
// two samples (the launcher requires planes of >= 128 pixels): it publishes the octet sums of each, sweeps the granules of each
// (slot = slab index inside the sample, in row order) and keeps both (A, C) tables in its OWN 1 KB of LDS -- no workgroup barrier;
// every lane then picks the table of its row's sample.
// SH3 (3x3 / stride 1 / pad 1 on row-major 256-pixel tiles): the three taps of one window ROW share ONE gather.  Tile row r is output
// pixel m0 + r; its centre-column operand of window row dy is input pixel (oy - 1 + dy, ox), and the dx = 0 / 2 operands are the
// centre-column operands of pixels m0 + r -+ 1 -- the neighbouring LDS rows -- unless the pixel sits on the left / right edge of its
// image row, where the tap lies in the zero padding (the lane then reads a row of zeros).  A K step gathered 32 KB for 32 MFMAs per
// wave and the gather's latency (~2 us under load, one step in flight per workgroup) set the pace: 0.18-0.22 of the MFMA peak on
// the 30^2 / 15^2 levels of the ResNet-UNet.  Now one gather (256 + 2 rows) feeds 96 MFMAs per wave, one barrier per window row.
// Stage layout: rows 0..255 as before (16-byte slots swizzled by (row >> 1) & 7), row 256 = pixel m0 - 1, row 257 = pixel m0 + 256
// (both unswizzled), row 258 = zeros.
// BMT = 128 (round 5; WN = 2 only): 128 pixels x 128 channels per workgroup, waves 2 x 2 of 64 pixels x 64 channels (2 x 2 tiles) --
// half the K chain per wave and twice the workgroups.  A 256 x 128 tile with K = 9 x 256 keeps one CU's matrix pipe busy for ~25 us
// whatever the batch; the 15 x 15 level of the ResNet-UNet at 38 rows is 68 such tiles on 256 CUs, 43 us per launch, fourteen
// launches per forward (profiles/r05f_oisst_nb38): the launcher takes this form while the large tiles leave CUs idle.
template <int WN, bool GNF = false, bool SH3 = false, int BMT = 256>
__global__ __launch_bounds__(256, 2) void conv_igemm2_kernel(ConvArgs a, int M, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(BMT == 256 || (BMT == 128 && WN == 2), "the 128-pixel tile exists for 128-channel column blocks");
    constexpr int BM = BMT, A_BYTES = BM * 128, RA = BM / 32, TH = BM / 16;  // (shadow the 256-pixel constants of the namespace)
    constexpr int STAGE_BYTES = SH3 ? (BM + 3) * 128 : A_BYTES;
    constexpr int EPI_OFF = 2 * STAGE_BYTES;             // output staging of the epilogue
    constexpr int LDS_END = EPI_OFF + 4 * 16 * OROW;     // (GNF: the waves' coefficient tables follow)
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int BN = 64 * WN, MT = (WN == 2 ? 4 : 2) * BM / 256;  // channels per workgroup, 32-pixel sub-tiles per wave
    constexpr int SLAB = 32 * MT;                                     // output rows of a wave (its GroupNorm statistics slab)
    constexpr int STEP_BYTES = BN * 128;              // weights of one K step of one column block: BN channels x 64 k bf16
    static_assert(!GNF || WN == 2, "the fused GroupNorm epilogue exists for WN = 2");
    const int wm = WN == 2 ? wave >> 1 : wave, wn = WN == 2 ? wave & 1 : 0;
    const int l31 = lane & 31, hi = lane >> 5;

    // XCD-aware tile id (bijective for any tile count): XCD x = bid % 8 takes a contiguous run of tiles
    const int total = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = tile % tiles_n, tm = tile / tiles_n;

    const int cin = a.c0 + a.c1;
    const int cpt = cin >> 6;
    const int ntaps = a.kh * a.kw;
    const int nk = ntaps * cpt;
    const int plane = a.ho * a.wo;
    const bool tile2d = (a.wo % 16 == 0) && (a.ho % TH == 0);
    const int tiles_x = a.wo >> 4, tiles_per_img = tile2d ? tiles_x * (a.ho / TH) : 1;
    int t_img = 0, t_y0 = 0, t_x0 = 0;
    if (tile2d) {
        t_img = tm / tiles_per_img;
        const int t = tm - t_img * tiles_per_img;
        t_y0 = (t / tiles_x) * TH;
        t_x0 = (t % tiles_x) * 16;
    }

    const size_t npix = (size_t)a.n * a.h * a.w;
    const int pitch0 = a.pix_pitch0 ? a.pix_pitch0 : a.c0;
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * pitch0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_frag, 0, (int)(unsigned)((size_t)a.cout * nk * 128), 0x00020000);

    // ---- per-lane gather descriptors (as conv.hip): DMA row (j*4 + wave)*8 + sub, LDS slot swizzled by (row >> 1) & 7
    const int sub = lane >> 3;
    const int gchunk = (lane & 7) ^ ((((wave & 1) << 2) | (sub >> 1)) & 7);
    int a_pix[SH3 ? 1 : RA];        // pixel index of tap (0,0): may be slightly negative (offsets are formed modulo 2^32)
    unsigned a_mask[SH3 ? 1 : RA];  // tap validity bits
#pragma unroll
    for (int j = 0; j < (SH3 ? 0 : RA); ++j) {
        const int row = (j * 4 + wave) * 8 + sub;
        unsigned mask = 0;
        int pix = 0;
        if (tm * BM + row < M) {
            int n_img, oy, ox;
            if (tile2d) {
                n_img = t_img;
                oy = t_y0 + (row >> 4);
                ox = t_x0 + (row & 15);
            } else {
                const int m = tm * BM + row;
                n_img = m / plane;
                const int rem = m - n_img * plane;
                oy = rem / a.wo;
                ox = rem - oy * a.wo;
            }
            const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
            pix = (n_img * a.h + iy0) * a.w + ix0;  // may be "negative": only ever used together with a valid tap
            const int ylo = max(0, -iy0), yhi = min(a.kh, a.h - iy0);
            const int xlo = max(0, -ix0), xhi = min(a.kw, a.w - ix0);
            const unsigned ym = yhi > ylo ? (1u << yhi) - (1u << ylo) : 0u;
            const unsigned xm = xhi > xlo ? (1u << xhi) - (1u << xlo) : 0u;
            for (int ky = 0; ky < a.kh; ++ky) mask |= (((ym >> ky) & 1u) ? xm : 0u) << (ky * a.kw);
        }
        a_mask[j] = mask;
        a_pix[j] = pix;
    }

    // SH3 (h == ho, w == wo: the input pixel of output pixel m at window row dy, centre column, is m + (dy - 1) w).  Registers are
    // short in this kernel (254 of 256 before SH3): DMA row j of this lane is output pixel m_lane + 32 j, three bits per row say
    // whether it exists (m < M) and whether it lies on the top / bottom image row (window row 0 / 2 in the zero padding).
    const int m_lane = tm * BM + wave * 8 + sub;
    unsigned vbits = 0, e_bits = 0;
    int e_m = 0;
    unsigned fa[2][MT];  // fragment addresses (inside a stage) of dx = 0 / 2; dx = 1 is the row itself (a_row + immediates)
    if constexpr (SH3) {
        auto rowbits = [&](long long m) -> unsigned {
            if (m < 0 || m >= M) return 0u;
            const int oy = (int)(m % plane) / a.wo;
            return 1u | (oy == 0 ? 2u : 0u) | (oy == a.h - 1 ? 4u : 0u);
        };
#pragma unroll
        for (int j = 0; j < RA; ++j) vbits |= rowbits(m_lane + 32 * j) << (3 * j);
        if (wave < 2) {  // extension rows: pixel m0 - 1 (wave 0) / m0 + 256 (wave 1)
            e_m = tm * BM + (wave == 0 ? -1 : BM);
            e_bits = rowbits(e_m);
        }
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int r = wm * (32 * MT) + mt * 32 + l31;
            const int m = min(tm * BM + r, M - 1);
            const int ox = (m % plane) % a.wo;
#pragma unroll
            for (int d = 0; d < 2; ++d) {
                const int rr = r + 2 * d - 1;
                const bool zero = d == 0 ? ox == 0 : ox == a.wo - 1;
                unsigned rowoff, key = 0;
                if (zero) rowoff = (BM + 2) * 128;
                else if (rr < 0) rowoff = BM * 128;
                else if (rr >= BM) rowoff = (BM + 1) * 128;
                else { rowoff = (unsigned)rr * 128u; key = (unsigned)(rr >> 1) & 7u; }
                fa[d][mt] = rowoff + ((((unsigned)hi) ^ key) << 4);
            }
        }
        if (tid < 16) *(uint4*)(smem + (tid >> 3) * STAGE_BYTES + (BM + 2) * 128 + (tid & 7) * 16) = make_uint4(0, 0, 0, 0);
        __syncthreads();
    }

    int is_tap = 0, is_chunk = 0;
    auto issue_a = [&](int stage) {  // LDS-DMA gather of K step (is_tap, is_chunk); taps fastest (L2 reuse of the window)
        char* As = smem + stage * STAGE_BYTES;
        if constexpr (SH3) {  // is_tap = 3 dy: ONE gather per window row -- the centre column of every tile row
            const int dy = is_tap / 3;
            const unsigned bad = dy == 0 ? 2u : dy == 2 ? 4u : 0u;  // window row 0 / 2 of a top / bottom image row: zero padding
            const int cb = is_chunk << 6;
            const bool second = cb >= a.c0;
            const int csrc = second ? a.c1 : pitch0;
            const unsigned cbytes = (unsigned)((second ? cb - a.c0 : cb) * 2);
            const int pitch_b = csrc * 2, shift = (dy - 1) * a.w;
#pragma unroll
            for (int j = 0; j < RA; ++j) {
                const unsigned b = (vbits >> (3 * j)) & 7u;
                const unsigned vo = ((b & 1u) && !(b & bad)) ? (unsigned)((m_lane + 32 * j + shift) * pitch_b) + cbytes + gchunk * 16 : 0xFFFFFFFFu;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
            }
            if (wave < 2 && lane < 8) {
                const unsigned vo = ((e_bits & 1u) && !(e_bits & bad)) ? (unsigned)((e_m + shift) * pitch_b) + cbytes + (unsigned)lane * 16u : 0xFFFFFFFFu;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(As + (BM + wave) * 128), 16, vo, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(As + (BM + wave) * 128), 16, vo, 0, 0, 0);
            }
            is_tap += 3;
            if (is_tap == 9) {
                is_tap = 0;
                ++is_chunk;
            }
            return;
        }
        const int dy = is_tap / a.kw, dx = is_tap - dy * a.kw;
        const unsigned tap_bit = 1u << is_tap;
        const int cb = is_chunk << 6;
        const bool second = cb >= a.c0;
        const int csrc = second ? a.c1 : pitch0;
        const unsigned toff = (unsigned)((dy * a.w + dx) * csrc * 2) + (unsigned)((second ? cb - a.c0 : cb) * 2) + gchunk * 16;
        const unsigned pitch_b = (unsigned)(csrc * 2);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const unsigned vo = (a_mask[j] & tap_bit) ? (unsigned)(a_pix[j] * (int)pitch_b) + toff : 0xFFFFFFFFu;
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
        }
        if (++is_tap == ntaps) {
            is_tap = 0;
            ++is_chunk;
        }
    };

    f32x16 acc[MT][2];  // [pixel sub-tile mt: rows wm*32*MT + mt*32 ..][32-channel half]
#pragma unroll
    for (int mt = 0; mt < MT; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.0f;

    // fragment addresses: row r of the stage at r*128, 16-B slot (k16*2 + hi) ^ ((r >> 1) & 7); sub-tile rows are
    // 32-aligned + l31, so the key is (l31 >> 1) & 7 for every mt and the sub-step only flips bits 5-6
    const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
    const unsigned a_row = (unsigned)((wm * (32 * MT) + l31) * 128);
    const unsigned a_x = (unsigned)((hi ^ ((l31 >> 1) & 7)) << 4);
    const unsigned w_voff = (unsigned)lane * 16u;
    // weight stream of this wave: [tn][K step][wn][ks][half][lane] x 16 B
    unsigned soff_cur = (unsigned)(tn * nk) * STEP_BYTES + (unsigned)wn * (STEP_BYTES / 2);
    const unsigned soff_last = soff_cur + (unsigned)(nk - 1) * STEP_BYTES;

    u32x4 bq[4][2];
    el16x8_t aq[2][MT];
#define ISSUE_B(SET, SOFF, KS)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                         \
        bq[SET][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + nt * 1024, (SOFF) + (KS) * 2048, 0);
#define DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
#define LGKM_WAIT(N)                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");               \
    __builtin_amdgcn_sched_barrier(0);
#define RDA1(SET, KS, MT)                                                                                    \
    {                                                                                                        \
        if constexpr (SH3 && dxi != 1) {                                                                     \
            const unsigned pm = (fa[dxi >> 1][MT] ^ (unsigned)((KS) << 5)) + (As - a_row);                   \
            DSR(aq[SET][MT], pm, 0)                                                                          \
        } else {                                                                                             \
            const unsigned pm = (a_x ^ (unsigned)((KS) << 5)) + As;                                          \
            DSR(aq[SET][MT], pm, (MT) * 4096)                                                                \
        }                                                                                                    \
    }
#define MF(MT, NT, ASET, BSET)                                                                               \
    acc[MT][NT] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, bq[BSET][NT]), aq[ASET][MT], \
                                                          acc[MT][NT], 0, 0, 0);
#define PIN __builtin_amdgcn_sched_barrier(0);
    // one k16 sub-step: 8 MFMAs; the weight fragments of sub-step +3 and the pixel fragments of sub-step +1 are issued
    // between them
#define SLOT(ISET, SOFF, IKS, LOAD, LSET, LKS, USE_A, USE_B)                                                 \
    {                                                                                                        \
        LGKM_WAIT(0)                                                                                         \
        ISSUE_B(ISET, SOFF, IKS)                                                                             \
        MF(0, 0, USE_A, USE_B) PIN                                                                           \
        if (LOAD) RDA1(LSET, LKS, 0)                                                                         \
        MF(1, 0, USE_A, USE_B) PIN                                                                           \
        if (LOAD) RDA1(LSET, LKS, 1)                                                                         \
        if constexpr (MT == 4) {                                                                             \
            MF(2, 0, USE_A, USE_B) PIN                                                                       \
            if (LOAD) RDA1(LSET, LKS, 2)                                                                     \
            MF(3, 0, USE_A, USE_B) PIN                                                                       \
            if (LOAD) RDA1(LSET, LKS, 3)                                                                     \
            MF(0, 1, USE_A, USE_B) PIN                                                                       \
            MF(1, 1, USE_A, USE_B) MF(2, 1, USE_A, USE_B) MF(3, 1, USE_A, USE_B) PIN                         \
        } else {                                                                                             \
            MF(0, 1, USE_A, USE_B) MF(1, 1, USE_A, USE_B) PIN                                                \
        }                                                                                                    \
    }

    issue_a(0);
    ISSUE_B(0, soff_cur, 0)
    ISSUE_B(1, soff_cur, 1)
    ISSUE_B(2, soff_cur, 2)
    if constexpr (SH3) {
        const int ngroups = 3 * cpt;  // (chunk, window row): three K steps (dx = 0, 1, 2) per gather
        for (int g = 0; g < ngroups; ++g) {
            if (g == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // first gather (older than the 6 weight loads)
            __builtin_amdgcn_s_barrier();  // gather of group g visible; every wave is done with the other stage
            __builtin_amdgcn_sched_barrier(0);
            if (g + 1 < ngroups) issue_a((g + 1) & 1);
            const unsigned As = lds_base + (unsigned)((g & 1) * STAGE_BYTES) + a_row;
#define KSTEP3(DX)                                                                                           \
            {                                                                                                \
                constexpr int dxi = DX;                                                                      \
                const unsigned soff_next = soff_cur < soff_last ? soff_cur + STEP_BYTES : soff_cur;          \
                RDA1(0, 0, 0) RDA1(0, 0, 1)                                                                  \
                if constexpr (MT == 4) { RDA1(0, 0, 2) RDA1(0, 0, 3) }                                       \
                SLOT(3, soff_cur, 3, true, 1, 1, 0, 0)                                                       \
                SLOT(0, soff_next, 0, true, 0, 2, 1, 1)                                                      \
                SLOT(1, soff_next, 1, true, 1, 3, 0, 2)                                                      \
                SLOT(2, soff_next, 2, false, 0, 0, 1, 3)                                                     \
                soff_cur = soff_next;                                                                        \
            }
            KSTEP3(0) KSTEP3(1) KSTEP3(2)
#undef KSTEP3
        }
    } else {
        constexpr int dxi = 0;  // (RDA1's SH3 branch is discarded)
        for (int k = 0; k < nk; ++k) {
            const unsigned soff_next = soff_cur < soff_last ? soff_cur + STEP_BYTES : soff_cur;  // tail: harmless re-fetch
            if (k == 0) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");  // first gather (older than the 6 weight loads)
            __builtin_amdgcn_s_barrier();  // gather of step k visible; every wave is done with the other stage
            __builtin_amdgcn_sched_barrier(0);
            if (k + 1 < nk) issue_a((k + 1) & 1);
            const unsigned As = lds_base + (unsigned)((k & 1) * A_BYTES) + a_row;
            RDA1(0, 0, 0) RDA1(0, 0, 1)
            if constexpr (MT == 4) { RDA1(0, 0, 2) RDA1(0, 0, 3) }
            SLOT(3, soff_cur, 3, true, 1, 1, 0, 0)
            SLOT(0, soff_next, 0, true, 0, 2, 1, 1)
            SLOT(1, soff_next, 1, true, 1, 3, 0, 2)
            SLOT(2, soff_next, 2, false, 0, 0, 1, 3)
            soff_cur = soff_next;
        }
    }
#undef SLOT
#undef PIN
#undef MF
#undef RDA1
#undef LGKM_WAIT
#undef DSR
#undef ISSUE_B

    // ---- epilogue straight from the accumulators (same scheme as conv_igemm_kernel): lane (l31, hi) of accumulator
    // (mt, nt) holds pixel l31 of sub-tile mt and channels nt*32 + 8*g + 4*hi + {0..3}
    // Order of memory operations: the coefficient loads of step s + 1 are issued BEFORE the store of step s, and full tiles run a
    // branch-free copy (no exec-masked `if (valid)` around the stores).  A global load issued after a store has to wait for
    // vmcnt(0), i.e. for the store's acknowledgement (gfx9 counts stores in vmcnt; the compiler cannot hoist a load over a store that
    // may alias), and a branch join makes the compiler's s_waitcnt model fall back to vmcnt(0) as well: the 16 steps of a wave's
    // epilogue used to expose 16 store round trips.
    // FAST: 16-bit output only, no residual (every hot launch): those pointer tests are wave-uniform branches, and joins too.
    // ---- GNF phases A and B (gn_fused.h): statistics of y = acc + bias per (sample, octet), exchanged through granules
    float* const gcf = (float*)(smem + LDS_END) + wave * 256;  // this wave's (A, C) tables: [2 samples][A[64] | C[64]]
    int g_bnd = 0x7fffffff;  // first output row (index into M) of the slab's SECOND sample
    if constexpr (GNF) {
        const GnFuse& G = a.gnf;
        const uint32_t tag = (*G.epoch << 8) | G.conv_tag;
        const int slab0 = tm * BM + wm * SLAB;  // first row of this wave's slab
        const int ch0 = tn * BN + wn * 64;     // first channel of this wave's block
        const int oct = a.cout >> 3, cpg = a.cout / G.groups;
        int nA, slotA, nslotsA, nslotsB = 0, slotB = 0;
        bool cross = false;
        if (tile2d) {
            nA = t_img;
            nslotsA = tiles_per_img * 2;
            slotA = (tm - t_img * tiles_per_img) * 2 + wm;
        } else {
            nA = slab0 / plane;
            g_bnd = (nA + 1) * plane;
            cross = slab0 + SLAB > g_bnd && g_bnd < M;
            const int fA = (nA * plane) / SLAB;
            slotA = slab0 / SLAB - fA;
            nslotsA = (g_bnd - 1) / SLAB - fA + 1;
            if (cross) {
                const int fB = g_bnd / SLAB;
                slotB = slab0 / SLAB - fB;  // = 0: the slab that crosses into a sample is that sample's first
                nslotsB = (g_bnd + plane - 1) / SLAB - fB + 1;
            }
        }
        auto publish = [&](int n_img, int slot, bool second) {
            float mval[MT];
#pragma unroll
            for (int mt = 0; mt < MT; ++mt) {
                const int row = slab0 + mt * 32 + l31;
                const bool in = tile2d ? true : (second ? (row >= g_bnd && row < M) : (row < g_bnd && row < M));
                mval[mt] = in ? 1.0f : 0.0f;
            }
            float w[16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *(const float4*)(G.bias + ch0 + nt * 32 + 8 * g + 4 * hi);
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int mt = 0; mt < MT; ++mt) {
                        const float y0 = acc[mt][nt][4 * g + 0] + b4.x, y1 = acc[mt][nt][4 * g + 1] + b4.y;
                        const float y2 = acc[mt][nt][4 * g + 2] + b4.z, y3 = acc[mt][nt][4 * g + 3] + b4.w;
                        s1 = fmaf(mval[mt], (y0 + y1) + (y2 + y3), s1);
                        s2 = fmaf(mval[mt], fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, y3 * y3))), s2);
                    }
                    w[2 * (4 * nt + g)] = s1;
                    w[2 * (4 * nt + g) + 1] = s2;
                }
#pragma unroll
            for (int half = 8, d = 1; half >= 1; half >>= 1, d <<= 1) {  // reduce-scatter butterfly (conv_up_halo_kernel<5>)
                const bool up = (lane & d) != 0;
#pragma unroll
                for (int j = 0; j < half; ++j) {
                    const float send = up ? w[j] : w[j + half];
                    const float keep = up ? w[j + half] : w[j];
                    w[j] = keep + __shfl_xor(send, d, 64);
                }
            }
            float tot = w[0];
            tot += __shfl_xor(tot, 16, 64);
            tot += __shfl_xor(tot, 32, 64);
            if (lane < 16) {
                const int idx = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
                gn_store_granule(G.gran + (((size_t)n_img * G.max_slots + slot) * oct + (ch0 >> 3)) * 2 + idx, tag, tot);
            }
        };
        const bool liveA = slab0 < M;  // slabs beyond the last row publish nothing and need no coefficients
        if (liveA) publish(nA, slotA, false);
        if (cross) publish(nA + 1, slotB, true);
        const double inv_count = 1.0 / ((double)plane * cpg);
        auto coefs = [&](int n_img, int nslots, float* dst) {
            const float2 mr = gn_fuse_sweep<16>(G.gran + ((size_t)n_img * G.max_slots * oct + (ch0 >> 3)) * 2, oct * 2, nslots, tag ^ G.test_tag_xor, cpg,
                                                inv_count, G.err, lane, G.timeout_ticks);
            const float2 ac = gn_fuse_coef(G, ch0 + lane, a.coef_div > 1 ? n_img / a.coef_div : n_img, mr);
            dst[lane] = ac.x;
            dst[64 + lane] = ac.y;
        };
        if (G.slots < 0) nslotsA = nslotsB = 0;  // timing experiment
        if (liveA) coefs(nA, nslotsA, gcf);
        if (cross) coefs(nA + 1, nslotsB, gcf + 128);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // this wave's own table writes, read back below by its other lanes
    }
    auto epilogue = [&](auto act_c, auto mode_c, auto full_c, auto fast_c) {
        constexpr int ACT = decltype(act_c)::value, MODE = decltype(mode_c)::value;
        constexpr bool FULL = decltype(full_c)::value, FAST = decltype(fast_c)::value;
        uint32_t ob_[MT], row0_[MT], cb_[MT];
        RngKey key_[MT];
        bool valid_[MT];
#pragma unroll
        for (int mt = 0; mt < MT; ++mt) {
            const int row = wm * (32 * MT) + mt * 32 + l31;
            valid_[mt] = FULL || tm * BM + row < M;
            int m, n_img;
            if (tile2d) {
                n_img = t_img;
                m = (n_img * a.ho + t_y0 + (row >> 4)) * a.wo + t_x0 + (row & 15);
            } else {
                m = valid_[mt] ? tm * BM + row : 0;
                n_img = m / plane;
            }
            ob_[mt] = (uint32_t)m * (uint32_t)a.cout + (uint32_t)(tn * BN + wn * 64);
            key_[mt] = drop_row_key(a.drop, n_img);  // dropout streams are per batch row
            row0_[mt] = (uint32_t)n_img * (uint32_t)(a.ho * a.wo * a.cout);
            cb_[mt] = (uint32_t)((a.coef_div > 1 ? n_img / a.coef_div : n_img) * a.coef_stride + tn * BN + wn * 64 + 4 * hi);
        }
        const float ps = drop_prescale<ACT, MODE>(a.drop);  // dropout scale folded into the affine
        auto load_coef = [&](int step, float4 (&c)[4]) {
            const int mt = step >> 2, cg0 = ((step >> 1) & 1) * 32 + 16 * (step & 1);
            if constexpr (GNF) {  // the table of the row's sample (first / second sample of the slab), channels + 4 * hi
                const float* t = gcf + ((!tile2d && tm * BM + wm * (32 * MT) + mt * 32 + l31 >= g_bnd) ? 128 : 0) + 4 * hi + cg0;
                c[0] = *(const float4*)(t);
                c[1] = *(const float4*)(t + 8);
                c[2] = *(const float4*)(t + 64);
                c[3] = *(const float4*)(t + 72);
                return;
            }
            c[0] = *(const float4*)(a.coef_a + cb_[mt] + cg0);
            c[1] = *(const float4*)(a.coef_a + cb_[mt] + cg0 + 8);
            c[2] = *(const float4*)(a.coef_c + cb_[mt] + cg0);
            c[3] = *(const float4*)(a.coef_c + cb_[mt] + cg0 + 8);
        };
        float4 cf[2][4];
        load_coef(0, cf[0]);
        // GNF: the ResnetBlock's shortcut, fetched one step ahead of its use, BEFORE the previous step's stores (as the coefficients)
        const bool gnf_res = GNF && a.residual != nullptr;
        uint2 rs[2][2];
        auto load_res = [&](int step, uint2 (&r)[2]) {
            const int mt = step >> 2, cg0 = ((step >> 1) & 1) * 32 + 16 * (step & 1);
            const size_t e0 = (size_t)(ob_[mt] + cg0 + 4 * hi);
            r[0] = valid_[mt] ? *(const uint2*)(a.residual + e0) : make_uint2(0, 0);
            r[1] = valid_[mt] ? *(const uint2*)(a.residual + e0 + 8) : make_uint2(0, 0);
        };
        if (gnf_res) load_res(0, rs[0]);
        // FAST + FULL (every hot launch): a wave's 32 x 64 sub-tile is 32 whole 128-byte lines; its 16-byte rows go through a per-wave
        // LDS staging tile, half at a time, and leave as stores of 8 whole lines (stored from the registers a store instruction
        // writes a 32-byte piece of 32 lines -- worth 20 % of the store-bound conv_enc0_stem_kernel, 1-5 % of the halo kernels)
        constexpr bool STAGED = FAST && FULL;
        unsigned char* ost = (unsigned char*)smem + EPI_OFF + wave * (16 * OROW);
        const int rpx = lane >> 3, rch = lane & 7;  // read-back role: (pixel 8 k + rpx of the half, 16-byte chunk)
        uint4 ostage[4];
#pragma unroll
        for (int step = 0; step < 4 * MT; ++step) {
            const int mt = step >> 2, nt = (step >> 1) & 1, g2 = step & 1;
            if (step + 1 < 4 * MT) load_coef(step + 1, cf[(step + 1) & 1]);
            if (gnf_res && step + 1 < 4 * MT) load_res(step + 1, rs[(step + 1) & 1]);
            const float4 ca0 = cf[step & 1][0], ca1 = cf[step & 1][1], cc0 = cf[step & 1][2], cc1 = cf[step & 1][3];
            const bool valid = valid_[mt];
            const uint32_t ob = ob_[mt], row0 = row0_[mt];
            const RngKey key = key_[mt];
            {
                {
                    const int cg0 = nt * 32 + 16 * g2;  // + 4*hi: own channels of group 2*g2; + 8: group 2*g2+1
                    const float ca[8] = {ca0.x * ps, ca0.y * ps, ca0.z * ps, ca0.w * ps, ca1.x * ps, ca1.y * ps, ca1.z * ps, ca1.w * ps};
                    const float cc[8] = {cc0.x * ps, cc0.y * ps, cc0.z * ps, cc0.w * ps, cc1.x * ps, cc1.y * ps, cc1.z * ps, cc1.w * ps};
                    const uint32_t e0 = ob + cg0 + 4 * hi;
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = fmaf(acc[mt][nt][8 * g2 + t], ca[t], cc[t]);
                    act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                    act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                    if (gnf_res) {
                        const uint32_t rw[4] = {rs[step & 1][0].x, rs[step & 1][0].y, rs[step & 1][1].x, rs[step & 1][1].y};
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            v[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
                    } else if (!FAST && a.residual) {
                        const uint2 r0 = valid ? *(const uint2*)(a.residual + (size_t)e0) : make_uint2(0, 0);
                        const uint2 r1 = valid ? *(const uint2*)(a.residual + (size_t)e0 + 8) : make_uint2(0, 0);
                        const uint32_t rw[4] = {r0.x, r0.y, r1.x, r1.y};
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            v[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
                    }
                    if (!FAST && a.out_f32 && valid) {
                        *(float4*)(a.out_f32 + (size_t)e0) = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(a.out_f32 + (size_t)e0 + 8) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                    if (FAST || a.out_el16) {
                        uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                        uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        if (STAGED) ostage[step & 3] = o;
                        else if (FULL || valid) *(uint4*)(a.out_el16 + (size_t)(ob + cg0 + 8 * hi)) = o;
                    }
                }
            }
            if (STAGED && (step & 3) == 3) {
                const int row0w = wm * (32 * MT) + mt * 32;  // first tile row of this sub-tile
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if ((l31 >> 4) == h) {
#pragma unroll
                        for (int q = 0; q < 4; ++q)  // q = nt * 2 + g2: channels 16 q + 8 hi ..
                            *(uint4*)(ost + (l31 & 15) * OROW + (16 * q + 8 * hi) * 2) = ostage[q];
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int rowp = row0w + 16 * h + 8 * k + rpx;
                        const int mp = tile2d ? (t_img * a.ho + t_y0 + (rowp >> 4)) * a.wo + t_x0 + (rowp & 15) : tm * BM + rowp;
                        const uint4 val = *(const uint4*)(ost + (8 * k + rpx) * OROW + rch * 16);
                        *(uint4*)(a.out_el16 + (size_t)((uint32_t)mp * (uint32_t)a.cout + (uint32_t)(tn * BN + wn * 64) + rch * 8)) = val;
                    }
                }
            }
        }
    };
    const bool tile_full = tile2d || (tm + 1) * BM <= M;  // wave-uniform: every row of the tile is an output pixel
    const bool fast = a.out_el16 != nullptr && a.out_f32 == nullptr && (GNF || a.residual == nullptr);  // GNF adds its residual itself
    auto by_full = [&](auto act_c, auto mode_c) {
        if (tile_full && fast) epilogue(act_c, mode_c, std::true_type{}, std::true_type{});
        else if (tile_full) epilogue(act_c, mode_c, std::true_type{}, std::false_type{});
        else epilogue(act_c, mode_c, std::false_type{}, std::false_type{});
    };
    auto by_mode = [&](auto act_c) {
        if (a.drop.mode == 0) by_full(act_c, std::integral_constant<int, 0>{});
        else if (a.drop.mode == 1) by_full(act_c, std::integral_constant<int, 1>{});
        else by_full(act_c, std::integral_constant<int, 2>{});
    };
    if constexpr (GNF) {  // SiLU, dropout off or from the engine's generator (the launcher checks)
        if (a.drop.mode == 1) by_full(std::integral_constant<int, ACT_SILU>{}, std::integral_constant<int, 1>{});
        else by_full(std::integral_constant<int, ACT_SILU>{}, std::integral_constant<int, 0>{});
        return;
    }
    if (a.act == ACT_RELU) by_mode(std::integral_constant<int, ACT_RELU>{});
    else if (a.act == ACT_LEAKY) by_mode(std::integral_constant<int, ACT_LEAKY>{});
    else if (a.act == ACT_SILU) by_mode(std::integral_constant<int, ACT_SILU>{});
    else by_mode(std::integral_constant<int, ACT_NONE>{});
#endif
}

bool conv_igemm2_tile2d(int ho, int wo) { return wo % 16 == 0 && ho % TH == 0; }

int conv_igemm2_gn_slots(int ho, int wo) {
    const int plane = ho * wo;
    if (conv_igemm2_tile2d(ho, wo)) return (wo / 16) * (ho / TH) * 2;  // 2-D tiles: two 128-row slabs per tile, all inside the sample
    if (plane < 128) return 0;  // a slab would touch more than two samples
    return (plane + 127) / 128 + 1;
}

// ... of the 128-pixel tile form (64-row slabs; its 2-D tiles are 8 rows x 16 pixels)
int conv_igemm2_gn_slots_bm128(int ho, int wo) {
    const int plane = ho * wo;
    if (wo % 16 == 0 && ho % 8 == 0) return (wo / 16) * (ho / 8) * 2;
    if (plane < 64) return 0;
    return (plane + 63) / 64 + 1;
}

// wpk [cout][taps][cin] bf16 -> MFMA fragment order [column block tn (128 ch)][K step = chunk*taps + tap][wn][ks][half]
// [lane][8 k]: lane (l31, hi) of fragment (wn, ks, half) holds channel tn*128 + wn*64 + half*32 + l31,
// k = chunk*64 + ks*16 + hi*8 + {0..7} of tap `tap`
void pack_conv_frag(const el16_t* wpk, int cout, int taps, int cin, el16_t* out) {
    const int cpt = cin / 64;
    const int wns = cout % 128 == 0 ? 2 : 1, bn = 64 * wns;  // column block: 128 channels, or 64 (conv_igemm2_kernel<1>)
    size_t o = 0;
    for (int tn = 0; tn < cout / bn; ++tn)
        for (int chunk = 0; chunk < cpt; ++chunk)
            for (int tap = 0; tap < taps; ++tap)
                for (int wn = 0; wn < wns; ++wn)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int half = 0; half < 2; ++half)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int co = tn * bn + wn * 64 + half * 32 + (lane & 31);
                                const int k0 = chunk * 64 + ks * 16 + (lane >> 5) * 8;
                                const el16_t* s = wpk + ((size_t)co * taps + tap) * cin + k0;
                                for (int e = 0; e < 8; ++e) out[o++] = s[e];
                            }
}

bool conv_igemm2_supported(const ConvArgs& a) {
    if (a.up2x || a.wpk_frag == nullptr) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && a.c1 % 64 == 0 && a.cout % 64 == 0)) return false;
    if (a.kh * a.kw > 32) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    const int pitch0 = a.pix_pitch0 ? a.pix_pitch0 : a.c0;
    return npix * pitch0 * 2 < 0x7F000000ull && npix * (size_t)a.c1 * 2 < 0x7F000000ull &&
           (size_t)a.cout * a.kh * a.kw * (a.c0 + a.c1) * 2 < 0x7F000000ull && (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t conv_igemm2_init() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_igemm2_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<2, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL + 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_igemm2_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<2, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL_SH3);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL_SH3 + 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<1, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL_SH3);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<2, true, false, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL + 4096);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)(conv_igemm2_kernel<2, true, true, 128>), hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL_SH3 + 4096);
    return e;
}

hipError_t launch_conv_igemm2(const ConvArgs& a, hipStream_t stream) {
    const long long M = (long long)a.n * a.ho * a.wo;
    const int tiles_m = (int)((M + BM - 1) / BM);
    // 256 x 128 tiles whenever cout allows.  Tried (round 3): 256 x 64 tiles when the large ones quantise badly on the 512 resident
    // workgroups -- the 15 x 15 level of the ResNet-UNet at 300 rows is 528 tiles = two rounds for 1.03 rounds of work, 1 056 small
    // tiles are three rounds of half the size -- but a small tile takes ~0.75 of a large one's time (every pixel fragment feeds half
    // as many MFMAs): OISST rollout 675 -> 698 ms.  DYF_IGEMM2_BALANCE=1 re-enables the experiment.
    const bool balance = dyf_form("DYF_IGEMM2_BALANCE") && atoi(dyf_form("DYF_IGEMM2_BALANCE")) != 0;
    bool small = a.cout % 128 != 0;
    if (!small && balance) {
        const long long sel = a.n_sel > 0 ? ((long long)a.n_sel * a.ho * a.wo + BM - 1) / BM : tiles_m;
        const long long tb = sel * (a.cout / 128), ts = sel * (a.cout / 64);
        small = 0.56 * (double)((ts + 511) / 512) < 0.92 * (double)((tb + 511) / 512);
    }
    // SH3 (one gather per window row): 3x3 / stride 1 / pad 1 on row-major tiles (the kernel tiles 2-D when wo % 16 == 0 && ho % 16 == 0)
    const bool sh3_on = !(dyf_form("DYF_IGEMM2_SH3") && atoi(dyf_form("DYF_IGEMM2_SH3")) == 0);  // (read per launch: tests compare the two forms)
    const bool sh3 = sh3_on && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.ho == a.h && a.wo == a.w && a.wo >= 2 &&
                     !(a.wo % 16 == 0 && a.ho % TH == 0);
    if (a.gnf.gran != nullptr && a.gnf.bm == 128) {  // the 128-pixel tile form (launch_conv_gn_fused chose it and counted its slots)
        dyf_form_note("conv_igemm2_kernel<2>+gn_fused", a.n);
        dyf_form_note("conv_igemm2_kernel<2,bm128>+gn_fused", a.n);
        if (sh3) dyf_form_note("conv_igemm2_kernel+sh3", a.n);
        const int tiles_n = a.cout / 128, tiles_m128 = (int)((M + 127) / 128);
        const bool sh3_128 = sh3_on && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.ho == a.h && a.wo == a.w && a.wo >= 2 &&
                             !(a.wo % 16 == 0 && a.ho % 8 == 0);
        // (LDS: the 256-pixel sizes are requested -- an upper bound; two workgroups per CU either way)
        if (sh3_128)
            hipLaunchKernelGGL((conv_igemm2_kernel<2, true, true, 128>), dim3(tiles_m128 * tiles_n), dim3(256), 2 * (128 + 3) * 128 + 4 * 16 * OROW + 4096, stream, a, (int)M, tiles_m128, tiles_n);
        else
            hipLaunchKernelGGL((conv_igemm2_kernel<2, true, false, 128>), dim3(tiles_m128 * tiles_n), dim3(256), 2 * 128 * 128 + 4 * 16 * OROW + 4096, stream, a, (int)M, tiles_m128, tiles_n);
        return hipGetLastError();
    }
    if (a.gnf.gran != nullptr) {  // GroupNorm fused (launch_conv_gn_fused checked the shape): + 4 KB of LDS for the waves' (A, C) tables
        dyf_form_note("conv_igemm2_kernel<2>+gn_fused", a.n);
        if (sh3) dyf_form_note("conv_igemm2_kernel+sh3", a.n);  // (a note of its own: the form log's kernel names stay those of the tile shape)
        const int tiles_n = a.cout / 128;
        ConvArgs b = a;
#ifdef DYF_EXPERIMENT_BUILD
        if (dyf_form("DYF_GN_FUSE_NOWAIT")) b.gnf.slots = -1;  // timing experiment (WRONG results): no granule sweep
#endif
        if (sh3)
            hipLaunchKernelGGL((conv_igemm2_kernel<2, true, true>), dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL_SH3 + 4096, stream, b, (int)M, tiles_m, tiles_n);
        else
            hipLaunchKernelGGL((conv_igemm2_kernel<2, true>), dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL + 4096, stream, b, (int)M, tiles_m, tiles_n);
        return hipGetLastError();
    }
    dyf_form_note(small ? "conv_igemm2_kernel<1>" : "conv_igemm2_kernel<2>", a.n);
    if (sh3) dyf_form_note("conv_igemm2_kernel+sh3", a.n);
    if (!small) {
        const int tiles_n = a.cout / 128;
        if (sh3)
            hipLaunchKernelGGL((conv_igemm2_kernel<2, false, true>), dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL_SH3, stream, a, (int)M, tiles_m, tiles_n);
        else
            hipLaunchKernelGGL(conv_igemm2_kernel<2>, dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL, stream, a, (int)M, tiles_m, tiles_n);
    } else {
        const int tiles_n = a.cout / 64;
        if (sh3)
            hipLaunchKernelGGL((conv_igemm2_kernel<1, false, true>), dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL_SH3, stream, a, (int)M, tiles_m, tiles_n);
        else
            hipLaunchKernelGGL(conv_igemm2_kernel<1>, dim3(tiles_m * tiles_n), dim3(256), LDS_TOTAL, stream, a, (int)M, tiles_m, tiles_n);
    }
    return hipGetLastError();
}

namespace {
std::mutex g_frag_mu;
std::unordered_map<const void*, const el16_t*> g_frag;
std::unordered_map<const void*, const el16_t*> g_frag3;  // halo-kernel fragments of plain 3x3 convs
std::unordered_map<const void*, const el16_t*> g_frag64; // pack_halo3_frag64 fragments of 3x3 convs whose g_frag3 entry is the 256-channel-block order
}  // namespace

void conv_register_frag(const el16_t* wpk_dev, const el16_t* frag_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    g_frag[(const void*)wpk_dev] = frag_dev;
}

void conv_unregister_frag(const void* wpk_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    g_frag.erase(wpk_dev);
    g_frag3.erase(wpk_dev);
    g_frag64.erase(wpk_dev);
}

void conv_register_frag64(const el16_t* wpk_dev, const el16_t* frag_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    g_frag64[(const void*)wpk_dev] = frag_dev;
}

const el16_t* conv_lookup_frag64(const el16_t* wpk_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    auto it = g_frag64.find((const void*)wpk_dev);
    return it == g_frag64.end() ? nullptr : it->second;
}

void conv_register_halo3_frag(const el16_t* wpk_dev, const el16_t* frag_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    g_frag3[(const void*)wpk_dev] = frag_dev;
}

const el16_t* conv_lookup_halo3_frag(const el16_t* wpk_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    auto it = g_frag3.find((const void*)wpk_dev);
    return it == g_frag3.end() ? nullptr : it->second;
}

const el16_t* conv_lookup_frag(const el16_t* wpk_dev) {
    std::lock_guard<std::mutex> lk(g_frag_mu);
    auto it = g_frag.find((const void*)wpk_dev);
    return it == g_frag.end() ? nullptr : it->second;
}
