// The "halo" convolution kernel (gfx950): fused x2-bilinear-upsample + 3x3 conv of the decoder blocks
// (src/models/unet_simple.py:40-52; dec3-dec5 = half of a forward's time), and -- same machinery, template SP = 2 / 3 --
// plain 3x3 / stride 1 and 4x4 / stride 2 convolutions with 256-channel output blocks.
//
// Upsample forms: same mathematics as conv_igemm_kernel<.., UP=1> (phase decomposition + border-correction taps, see
// conv.hip), with the correction taps evaluated beforehand by up_border_kernel (below) for the border ring of the output:
// border accumulators start from its result and every tile runs the same 9-tap stencil.  The data movement is designed around what bounded the previous forms, the CU's 128 B/clk of LDS bandwidth:
//   * a workgroup (4 waves, 256 registers each, TWO workgroups per CU) owns an 8x16 LOW-res tile and all four output
//     phases of 64 output channels: GEMM tile M = 128 pixels x N = 256 (4 phases x 64 channels).  A wave owns one phase:
//     128 pixels x 64 channels = 4 x 2 accumulator tiles of v_mfma_f32_32x32x16_bf16 (128 registers); every pixel
//     fragment read from LDS feeds 2 MFMAs, every weight fragment 4;
//   * per 64-channel chunk the 10x18 replicate-padded input window ("halo", 23 KB) is DMA'd into LDS ONCE
//     (buffer_load ... lds, double-buffered) and all 9 stencil taps read their pixel fragments
//     from it at shifted addresses.  This is the ONLY LDS traffic (64 B/clk/CU at MFMA peak);
//   * the weights never touch LDS: they are pre-packed on the host in MFMA FRAGMENT ORDER (pack_up2x_frag), so that one
//     wave-wide buffer_load_dwordx4 fetches 1 KB of contiguous memory = one 32-channel x 16-k fragment straight into
//     registers (L2/L1-resident; 32 B/clk/CU of the vector-memory path).  Four fragment sets are in flight 3 k16
//     sub-steps ahead of their use;
//   * consequently there is NO per-step workgroup barrier: the waves meet only once per 64-channel chunk (halo swap,
//     every 9 steps of 32 MFMAs) and drift apart in between; the second workgroup of the CU covers barrier, halo
//     swap and epilogue of the first;
//   * inside a k16 sub-step the 4 LDS reads (inline asm, hand-counted lgkmcnt; one v_xad_u32 of address arithmetic each,
//     per-tap base/swizzle terms precomputed) and the 2 weight loads of later sub-steps are pinned BETWEEN the 8 MFMAs;
//   * operands are swapped (D^T = W * X^T): an accumulator lane then holds 4 consecutive CHANNELS of one pixel, so the
//     epilogue (affine / activation / dropout / bf16 pack) runs straight out of the accumulators, pairs lanes l and
//     l+32 with v_permlane32_swap and stores 16 B per lane: no LDS round trip, no barrier.
// History (measured): gather form 800 TFLOP/s -> halo + weights through LDS 905 -> weights streamed, 128x128
// wave tiles, 1 workgroup/CU 1105 -> this form 1200 (dense upsample, corrections in-kernel) / 1310 (plain 3x3) -> corrections
// moved to the ring kernel.
#include "conv.h"
#include "gn_fused.h"

#include <algorithm>
#include <type_traits>
#include <vector>
#ifdef HALO_EXP_TIMELINE
#include <cstdio>
#include <string>
#endif
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int TILE_H = 8, TILE_W = 16;                  // low-res tile of a workgroup: 8 rows x 16 columns
constexpr int NWAVES = 4;
constexpr int STEP_BYTES = 32768;                       // weights of one (tap, chunk) step: 256 columns x 64 k bf16

// SP = 0: the 16 columns of a tile are contiguous (halo 10 x 18, double-buffered).
// SP = 2: PLAIN 3x3 / stride 1 / pad 1 convolution on the same machinery: no upsampling, the four waves own four 64-channel
//         blocks of 256 output channels instead of four phases, out-of-image halo pixels are zero-filled by the DMA's bounds
//         check (no correction taps at all).  Weights: pack_halo3_frag.
// SP = 3: 4x4 / stride 2 / pad 1 convolution as SP = 2 on the SPACE-TO-DEPTH view of its input: input pixel (2i+qy, 2j+qx)
//         belongs to parity plane (qy, qx) at half resolution, kernel row ky = 2a + qy + 1 is displacement a of that plane
//         (qy = 0: a in {0, +1}; qy = 1: a in {-1, 0}), so every (plane, 64-channel chunk) contributes a 2 x 2 stencil.
//         The DMA gathers one plane of one chunk per halo (per-lane source offsets: any pixel map is free), the K loop
//         walks 4 planes x c0/64 chunks x 4 taps.  Weights: pack_halo_s2_frag.
// SP = 4: SP = 3 for layers with a multiple of 128 (not 256) output channels: the workgroup's tile is 16 x 16 pixels, waves
//         (wpy, wpx) = (8-row half, 64-channel block); halo 18 x 18, single-buffered (two workgroups per CU still fit).
// SP = 5: plain 3x3 / stride 1 / pad 1 convolution for layers with a multiple of 64 (not 256) output channels -- the 64- and
//         128-channel levels of the ResNet-UNet, whose planes (60x60, 30x30, 512x512 ...) need not tile evenly.  The four waves
//         own four 8 x 16 PIXEL sub-tiles of a 16 x 32 tile and the SAME 64 output channels: one 18 x 34 halo (77 KB, single
//         buffer, two workgroups per CU), every weight fragment is requested by four waves at about the same time (three of
//         the four requests are L1 hits), pixels beyond the right / bottom edge of a ragged plane are zero-filled by the DMA's
//         bounds check and masked at the store.  The halo source offsets are recomputed per chunk instead of being parked in
//         LDS (no room).  Weights: pack_halo3_frag64.
// SP = 1: SPARSE COLUMNS -- the 16 columns of a tile are entries of a per-phase column list (ConvArgs::up_cols): only the
// output columns a later kernel reads are computed.  The NS backbone resamples its 256-wide grid to 42 native columns
// (unet_simple.py:195): the readout touches 104 of the 256 columns of the last decoder block, i.e. 52 of 128 low-res columns
// per horizontal phase.  A list tile spans up to 40 low-res columns (halo 10 x 40 = 51 KB, single-buffered so that two
// workgroups still fit a CU; the other workgroup covers the exposed halo swap).
template <int SP>
struct HaloCfg {
    static constexpr int W = SP == 1 ? 40 : SP == 5 ? 34 : 18;  // halo width in pixels
    static constexpr int TH = (SP == 4 || SP == 5) ? 16 : 8;    // tile rows
    static constexpr int TW = SP == 5 ? 32 : 16;        // tile columns
    static constexpr int REAL = (TH + 2) * W;
    static constexpr int PIX = (REAL + 7) / 8 * 8;      // padded to a multiple of 8 DMA rows
    static constexpr int BYTES = PIX * 128;             // 23 552 / 51 200
    static constexpr int NBUF = (SP == 1 || SP == 4 || SP == 5) ? 1 : 2;
    static constexpr bool PLAIN = SP >= 2;              // one output-channel block per wave, zero-padded window, no corrections
    static constexpr bool S2 = SP == 3 || SP == 4;      // 4x4 / stride 2 on parity planes
    static constexpr int BLK = SP == 4 ? 128 : SP == 5 ? 64 : 256;  // plain forms: output channels of a workgroup
    static constexpr int STEP = BLK * 128;              // weight bytes of one (tap, chunk) step: BLK columns x 64 k bf16
    static constexpr int HOFF_OFF = NBUF * BYTES + 512; // per-thread halo source offsets [PER_WAVE][256]
    static constexpr int INSTR = PIX / 8;               // wave-level DMA instructions per halo: 23 / 50
    static constexpr int PER_WAVE = (INSTR + NWAVES - 1) / NWAVES;
    static constexpr bool HAS_TAB = SP != 5;            // per-thread halo source offsets parked in LDS
    static constexpr int TAB_END = HOFF_OFF + (HAS_TAB ? PER_WAVE * 1024 : 0);    // 53 760 / 65 024 / (SP 5) 79 360 B
    // output staging of the stride-2 forms (-DHALO_NO_STAGE disables): per wave one pixel tile of [32 pixels][64 channels] 16-bit +
    // 16 B pad per pixel -> whole 128-byte lines per store instruction (see conv_halo_rows.hip); the other forms run on the rows
    // kernels (SP 0-2) or have no LDS left (SP 5)
#ifndef HALO_NO_STAGE
    static constexpr bool STAGE = SP == 3 || SP == 4;
#else
    static constexpr bool STAGE = false;
#endif
    static constexpr int OROW = 144;
    static constexpr int LDS_TOTAL = TAB_END + (STAGE ? NWAVES * 32 * OROW : 0);  // SP 3 / 4: 72 192 B, two workgroups per CU
};

}  // namespace

// PLAIN_EPI: only the (no activation, no dropout) epilogue is instantiated -- the launcher picks it for such launches (every 3x3
// conv of the ResNet-UNet: its activation follows the GroupNorm).  With all twelve (activation x dropout mode) epilogues in one
// kernel the SP = 5 form sits at 256 registers with spill slots; alone this one allocates without spilling: level-0 convs of the
// OISST rollout 121.3 -> 114.7 us (300 rows), 51.7 -> 48.7 us (100 rows).
// EPI = 2 (SP = 5 only): GroupNorm fused into this conv -- in-launch statistics exchange between the workgroups of a sample, then
// normalise + FiLM + SiLU + dropout (+ residual) in the epilogue (gn_fused.h; ConvArgs::gnf).
#ifdef HALO_EXP_TIMELINE  // experiment builds only (tools/build_variant.sh): per-wave shader-clock stamps at the phase boundaries of the EPI = 2 form
__device__ unsigned long long g_halo_tl[1 << 18];
#define TL_STAMP(K)                                                                                              \
    if (EPI == 2 && blockIdx.x < 8192 && lane == 0) g_halo_tl[(blockIdx.x * 4 + wave) * 8 + (K)] = __builtin_amdgcn_s_memtime();
#else
#define TL_STAMP(K)
#endif
template <int SP, int EPI = 0>
__global__ __launch_bounds__(256, 2) void conv_up_halo_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m,
                                                              int tiles_n, int xmode) {
#if defined(__HIP_DEVICE_COMPILE__)
    using H = HaloCfg<SP>;
    constexpr bool PLAIN_EPI = EPI == 1;
    static_assert(EPI != 2 || SP == 5, "the fused GroupNorm epilogue exists for SP = 5");
    constexpr int HALO_W = H::W, HALO_REAL = H::REAL, HALO_BYTES = H::BYTES, HOFF_OFF = H::HOFF_OFF;
    constexpr int HALO_INSTR = H::INSTR, HALO_PER_WAVE = H::PER_WAVE;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wpy = wave >> 1, wpx = wave & 1;  // output phase of this wave

    TL_STAMP(0)
    // XCD-aware tile id; the column blocks of one tile are consecutive (they share the halo in L2)
    const int total = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    int tn = tile % tiles_n, tm = tile / tiles_n;
    if (xmode & 1) {
        // column block per XCD: XCD x streams only the weights of block x % tiles_n (they stay in its 4 MB L2), the 8 / tiles_n
        // groups of XCDs split the pixel tiles; the column blocks of one tile no longer share their halo in L2
        const int groups = 8 / tiles_n, chunk = (tiles_m + groups - 1) / groups, g = xcd / tiles_n;
        tn = xcd - g * tiles_n;
        tm = g * chunk + (bid >> 3);
        if ((bid >> 3) >= chunk || tm >= tiles_m) return;
    }
    const int n_img = tm / tiles_per_img;
    const int t_in = tm - n_img * tiles_per_img;
    const int ty0 = (t_in / tiles_x) * H::TH;
    const int lx = t_in % tiles_x;              // column tile: 16 contiguous columns, or 16 entries of the column lists
    const int px_x = l31 & 15, px_r = l31 >> 4;
    // low-res column of this lane's pixels and the column the halo starts at
    int col, cbase, cstore = 0;
    bool lane_valid = true;
    if (SP == 1) {
        cbase = a.up_cbase[lx];
        const int entry = a.up_cols[wpx * a.up_npad + lx * 16 + px_x];  // bit 14: padding entry (computed, not stored)
        col = entry & 0x3FFF;
        lane_valid = (entry & 0x4000) == 0;
        cstore = a.up_cidx[wpx * a.up_npad + lx * 16 + px_x];  // column of output pixel (.., 2*col + px) in the compact tensor
    } else {
        cbase = lx * H::TW - 1;
        col = lx * H::TW + (SP == 5 ? 16 * wpx : 0) + px_x;
        if (SP == 5) lane_valid = col < a.w;  // ragged planes: columns beyond the image are computed on zeros and not stored
    }

    const int cin = a.c0 + a.c1;
    const int cpt = H::S2 ? 4 * (cin >> 6) : cin >> 6;  // K chunks: (SP = 3) 4 parity planes per 64-channel chunk
    const int gh = H::S2 ? a.ho : a.h, gw = H::S2 ? a.wo : a.w;  // the grid the halo / tiles live on
    // The conv's ZERO padding of the upsampled image (which replicate-clamping the stencil cannot express) is repaired by
    // correction taps that only the first / last output row / column see.  They are NOT computed here: up_border_kernel
    // evaluates them for the border ring of the output beforehand and the accumulators of border pixels start from its
    // result, so every tile runs the same 9-tap stencil (in-kernel corrections cost border tiles up to 7 extra taps, with the
    // other waves of the workgroup waiting at the chunk barrier: 19 % of dec4, 24 % of dec3).
    const unsigned long long tap_list = H::S2 ? 0x3210ull : 0x876543210ull;
    const int ntaps = H::S2 ? 4 : 9;

    const size_t npix = (size_t)a.n * a.h * a.w;
    // (SP = 5 with ConvArgs::up_nearest: src0 holds the (h / 2) x (w / 2) planes the gather reads at (y >> 1, x >> 1))
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)((SP == 5 && a.up_nearest ? npix / 4 : npix) * a.c0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_up_frag, 0,
                                                          (int)(unsigned)((size_t)((SP == 2 || SP == 5) ? 1 : 4) * a.cout * 16 * cin * 2), 0x00020000);  // SP = 3: 4 planes

    // LDS swizzle key of halo pixel hp (XORed into its 16-B chunk index).  A ds_read_b128 is serviced in four NON-contiguous
    // 16-lane groups ({0-3,12-15,20-27}, {4-11,16-19,28-31}, and the same +32): with lanes = 2 tile rows x 16 columns a group
    // holds 8 pixels of each row, column sets complementary.  The two rows are HALO_W = 18 pixels apart, so the plain key
    // (hp >> 1) & 7 repeats one key inside every group (2-way conflict: 7.5 instead of 4 LDS cycles per read, measured as
    // SQ_LDS_BANK_CONFLICT = 47 % of SQ_LDS_IDX_ACTIVE); subtracting the halo row makes every group hit 16 distinct slots
    // for every tap displacement (brute-forced over all taps / rows).
    // (the sparse form's columns come from lists: there the plain key conflicts less, 47 M vs 59 M cycles per launch)
#define HKEY(hp) (SP == 1 ? (((hp) >> 1) & 7) : ((((hp) >> 1) - (int)((unsigned)(hp) / (unsigned)HALO_W)) & 7))
    // ---- halo DMA descriptors: instruction i (i % 4 == wave) fills halo pixels [8i, 8i+8); lane -> (pixel, 16-B chunk)
    // (the per-lane source offsets are parked in LDS, not in registers: the K loop needs every VGPR it can get)
    const int sub = lane >> 3;
    unsigned* h_tab = (unsigned*)(smem + HOFF_OFF) + tid;
    auto halo_off = [&](int j) -> unsigned {
        const int i = j * NWAVES + wave;
        int hp = i * 8 + sub;
        if (hp > HALO_REAL - 1) hp = HALO_REAL - 1;  // padding slots re-read the last halo pixel
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int yy = ty0 - 1 + hy, xx = cbase + hx;
        const int y = min(max(yy, 0), gh - 1), x = min(max(xx, 0), gw - 1);  // replicate clamp (upsample forms)
        const int gch = (lane & 7) ^ HKEY(hp);  // swizzled source chunk of this linear LDS slot
        // SP = 3: halo pixel (y, x) of parity plane (0, 0) is input pixel (2y, 2x); the plane offset is added per chunk
        unsigned off = (H::S2 ? (unsigned)((n_img * a.h + 2 * y) * a.w + 2 * x) : (unsigned)((n_img * a.h + y) * a.w + x)) *
                           (unsigned)(a.c0 * 2) + gch * 16;  // c0 == c1 (checked on host)
        if (SP == 5 && a.up_nearest)  // nn.Upsample(scale_factor=2, mode="nearest") folded into the gather: input pixel (y, x) = low-res (y >> 1, x >> 1)
            off = (unsigned)((n_img * (a.h >> 1) + (y >> 1)) * (a.w >> 1) + (x >> 1)) * (unsigned)(a.c0 * 2) + gch * 16;
        if (H::PLAIN && (yy != y || xx != x)) off = 0xFFFFFFFFu;  // plain convs: zero padding = out-of-range DMA offset
        return off;
    };
    // SP = 5 has no LDS room for the table; its two workgroups per CU are set by LDS, not by registers (124 of 256 used), so the
    // offsets of a second chunk (two-source convs, K = 1 152) come from registers instead of being recomputed (~30 VALU each,
    // 20 per wave and chunk: PMC counted 2 560 vector instructions per wave against 384 MFMAs)
    unsigned h_reg[H::HAS_TAB ? 1 : HALO_PER_WAVE];
    if (H::HAS_TAB) {
#pragma unroll
        for (int j = 0; j < HALO_PER_WAVE; ++j) h_tab[j * 256] = halo_off(j);
    } else {
#pragma unroll
        for (int j = 0; j < HALO_PER_WAVE; ++j) h_reg[j] = halo_off(j);
    }

    auto issue_halo = [&](int chunk) {
#ifdef HALO_EXP_NO_HALO
        if (a.n > 0) return;  // timing experiment (wrong results): no halo DMA
#endif
        const int cb = H::S2 ? (chunk >> 2) << 6 : chunk << 6;
        const bool second = !H::S2 && cb >= a.c0;
        unsigned coff = (unsigned)((second ? cb - a.c0 : cb) * 2);
        if (H::S2) coff += (unsigned)((((chunk >> 1) & 1) * a.w + (chunk & 1)) * a.c0 * 2);  // plane (qy, qx) = chunk & 3
        char* dst = smem + (H::NBUF == 2 ? (chunk & 1) * HALO_BYTES : 0);
#pragma unroll
        for (int j = 0; j < HALO_PER_WAVE; ++j) {
            const int i = j * NWAVES + wave;
            if (i < HALO_INSTR) {
                unsigned vo = H::HAS_TAB ? h_tab[j * 256] : h_reg[H::HAS_TAB ? 0 : j];
                if (!H::PLAIN || vo != 0xFFFFFFFFu) vo += coff;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(dst + i * 1024), 16, vo, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(dst + i * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    // ---- weight-fragment stream: step (chunk, tap) of this column block starts at ((tn*cpt + chunk)*16 + tap) * 32 KB;
    // inside, [wn][ks][column tile][lane] x 16 B.  soff_cur / soff_next: this wave's base of the current / next step.
    const unsigned w_voff = (unsigned)lane * 16u;
    int it_pos = 0, it_chunk = 0;
    auto soff_of = [&](int pos, int chunk) {
        const int tap = (int)((tap_list >> (4 * pos)) & 15ull);
        return (unsigned)(((tn * cpt + chunk) * 16 + tap) * H::STEP + (SP == 4 ? 0 : wpy * (STEP_BYTES / 2)) + wpx * 2048);
    };
    unsigned soff_cur = soff_of(0, 0), soff_next = soff_cur;
    auto advance = [&]() {  // the tail re-fetches the last step (harmless, never consumed)
        soff_cur = soff_next;
        if (++it_pos == ntaps) { it_pos = 0; ++it_chunk; }
        if (it_chunk < cpt) soff_next = soff_of(it_pos, it_chunk);
    };
    advance();  // -> soff_cur = step 0, soff_next = step 1

    f32x16 acc[2][4];  // [32-channel half][pixel tile mt: tile rows 2*mt + {0,1}]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;
    if (!H::PLAIN) {
        // border pixels of the OUTPUT start from the correction sums of up_border_kernel (ring index: top row, bottom row,
        // left column, right column); lane (l31, hi) holds channels nt*32 + 8*g + 4*hi + {0..3} of its pixel
        const bool edge_col = col == 0 || col == gw - 1;
        if (ty0 == 0 || ty0 + H::TH == gh || __builtin_amdgcn_ballot_w64(edge_col) != 0ull) {
            const int ring_len = 2 * a.wo + 2 * (a.ho - 2);
            const int X = 2 * col + wpx;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const int Y = 2 * (ty0 + 2 * mt + px_r) + wpy;
                int ring = -1;
                if (Y == 0) ring = X;
                else if (Y == a.ho - 1) ring = a.wo + X;
                else if (X == 0) ring = 2 * a.wo + Y - 1;
                else if (X == a.wo - 1) ring = 2 * a.wo + (a.ho - 2) + Y - 1;
                if (ring >= 0) {
                    const float* cp = a.up_border + ((size_t)n_img * ring_len + ring) * a.cout + tn * 64 + 4 * hi;
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const float4 v = *(const float4*)(cp + nt * 32 + 8 * g);
                            acc[nt][mt][4 * g + 0] = v.x; acc[nt][mt][4 * g + 1] = v.y;
                            acc[nt][mt][4 * g + 2] = v.z; acc[nt][mt][4 * g + 3] = v.w;
                        }
                }
            }
        }
    }

    int hp0 = (px_r + 1 + ((SP == 4 || SP == 5) ? 8 * wpy : 0)) * HALO_W + (col - cbase);  // halo pixel of pixel tile 0 at the un-shifted tap; mt adds 2 rows
    const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);

    u32x4 bq[6][2];   // weight fragments: ring of 6 sets (9-tap modes: 5 sub-steps ahead), 4 sets in the 4-tap modes
    el16x8_t aq[2][4];  // pixel fragments: two sets
    unsigned ab[4], ax[4];  // LDS base / swizzle term of the tap whose pixel fragments are being fetched

    // weights of one (tap, chunk) step: WSTEP bytes, inside [ks][column tile][lane] x 16 B with KSS bytes per k16 sub-step
    constexpr unsigned WSTEP = SP == 5 ? 8192u : (unsigned)STEP_BYTES, KSS = SP == 5 ? 2048u : 4096u;
// timing experiments (wrong results): -DHALO_EXP_W_ALIAS serves the weight stream from 16 KB (L1 hits), -DHALO_EXP_LDS_SKIP
// drops the pixel-fragment reads of odd k16 sub-steps (half the LDS read traffic)
#ifdef HALO_EXP_W_ALIAS
#define W_ALIAS(x) ((x) & 0x3FFFu)
#else
#define W_ALIAS(x) (x)
#endif
#ifdef HALO_EXP_LDS_SKIP
#define LDS_KEEP(KS) (((KS) & 1) == 0)
#else
#define LDS_KEEP(KS) true
#endif
#define ISSUE_B(SET, SOFF, KS)                                                                               \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                         \
        bq[SET][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + nt * 1024, W_ALIAS((SOFF) + (KS) * KSS), 0);
#define DSR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
#define LGKM_WAIT(N)                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");               \
    __builtin_amdgcn_sched_barrier(0);
    // pixel fragments of sub-step KS at halo displacement DISP: 4 pixel tiles, 36 halo pixels (2 rows) apart
    // LDS address of pixel tile mt at tap displacement d, sub-step ks: ab[mt] + (ax[mt] ^ (ks << 5)) (one v_xad_u32);
    // ab / ax are recomputed once per tap (the XOR swizzle key depends on the halo pixel, not linearly on d)
#define TAPADDR(DISP)                                                                                        \
    {                                                                                                        \
        int hpb = hp0;                                                                                       \
        asm volatile("" : "+v"(hpb)); /* opaque: keeps 9 taps x 8 addresses from being hoisted and spilled */ \
        _Pragma("unroll") for (int mt = 0; mt < 4; ++mt) {                                                   \
            const int hpm = hpb + (DISP) + 2 * HALO_W * mt;                                                  \
            ab[mt] = Hs + hpm * 128;                                                                         \
            ax[mt] = (unsigned)((hi ^ HKEY(hpm)) << 4);                                               \
        }                                                                                                    \
    }
#define RDA1(SET, KS, MT)                                                                                    \
    {                                                                                                        \
        const unsigned pm = (ax[MT] ^ (unsigned)((KS) << 5)) + ab[MT];                                       \
        if (LDS_KEEP(KS)) DSR(aq[SET][MT], pm)                                                               \
    }
#define MF(NT, MT, ASET, BSET)                                                                               \
    acc[NT][MT] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, bq[BSET][NT]), aq[ASET][MT], \
                                                          acc[NT][MT], 0, 0, 0);
#define PIN __builtin_amdgcn_sched_barrier(0);
#define NO_PRE
#define SLOT(ISET, SOFF, IKS, LOAD, PRE, LSET, LKS, USE_A, USE_B)                                            \
    {                                                                                                        \
        LGKM_WAIT(0)                                                                                         \
        ISSUE_B(ISET, SOFF, IKS)                                                                             \
        MF(0, 0, USE_A, USE_B) PIN                                                                           \
        if (LOAD) { PRE RDA1(LSET, LKS, 0) }                                                                 \
        MF(0, 1, USE_A, USE_B) PIN                                                                           \
        if (LOAD) RDA1(LSET, LKS, 1)                                                                         \
        MF(0, 2, USE_A, USE_B) PIN                                                                           \
        if (LOAD) RDA1(LSET, LKS, 2)                                                                         \
        MF(0, 3, USE_A, USE_B) PIN                                                                           \
        if (LOAD) RDA1(LSET, LKS, 3)                                                                         \
        MF(1, 0, USE_A, USE_B) PIN                                                                           \
        MF(1, 1, USE_A, USE_B) MF(1, 2, USE_A, USE_B) MF(1, 3, USE_A, USE_B) PIN                             \
    }
#define STEP_CORE(HAS_NEXT, NEXT_ADDR)                                                                       \
    {                                                                                                        \
        SLOT(3, soff_cur, 3, true, NO_PRE, 1, 1, 0, 0)                                                       \
        SLOT(0, soff_next, 0, true, NO_PRE, 0, 2, 1, 1)                                                      \
        SLOT(1, soff_next, 1, true, NO_PRE, 1, 3, 0, 2)                                                      \
        SLOT(2, soff_next, 2, HAS_NEXT, NEXT_ADDR, 0, 0, 1, 3)                                               \
        advance();                                                                                           \
    }
#define STENCIL_STEP(DISP, HAS_NEXT, DNEXT) STEP_CORE(HAS_NEXT, TAPADDR(DNEXT))
    issue_halo(0);
    // 9-tap modes: the (chunk, tap) order is fixed, so the weight stream is addressed directly: sub-step g = 4*tap + ks of a
    // chunk uses ring set g % 6 and requests the fragments of sub-step g + 5 (36 sub-steps per chunk = 6 turns of the ring);
    // the L2 round trip of a fragment is longer than the 3 sub-steps the 4-set ring gave it
    const unsigned soff_w = SP == 5 ? 0u : (unsigned)(wpy * (STEP_BYTES / 2) + wpx * 2048);  // SP 5: all waves, same weights
    unsigned soff_c = (unsigned)((tn * cpt) * 16) * WSTEP + soff_w, soff_n = soff_c;
    // (pinned in program order: the compiler otherwise issues the oldest set LAST, and its own s_waitcnt at the loop head --
    // merged over the entry edge and the back edge -- becomes vmcnt(2) in the first sub-step of EVERY chunk: a full drain of
    // the weight ring and of the halo DMA just requested for the next chunk, i.e. no double buffering at all)
#ifndef HALO_NO_PREPIN
#define PPIN PIN
#else
#define PPIN
#endif
    PPIN
    if (H::S2) {
        ISSUE_B(0, soff_cur, 0) PPIN
        ISSUE_B(1, soff_cur, 1) PPIN
        ISSUE_B(2, soff_cur, 2) PPIN
    } else {
        ISSUE_B(0, soff_c, 0) PPIN
        ISSUE_B(1, soff_c, 1) PPIN
        ISSUE_B(2, soff_c, 2) PPIN
        ISSUE_B(3, soff_c, 3) PPIN
        ISSUE_B(4, soff_c + WSTEP, 0) PPIN
    }
    for (int chunk = 0; chunk < cpt; ++chunk) {
        if (H::NBUF == 2) {
            // halo of this chunk landed (everything older than the 6 / 10 weight loads in flight), every wave is done with the
            // other buffer -> prefetch the next chunk's halo into it
            if (H::S2) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (chunk + 1 < cpt) issue_halo(chunk + 1);
        } else {
            // single buffer: the halo was requested after every wave left the previous chunk (barrier below); it is
            // younger than the weight loads in flight, so wait for everything, then make it visible
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (chunk == 0) { TL_STAMP(1) }
        }
        const unsigned Hs = lds_base + (H::NBUF == 2 ? (chunk & 1) * HALO_BYTES : 0);
        asm volatile("" : "+v"(hp0));  // keep the per-tap LDS addresses from being hoisted out of the chunk loop
        if (H::S2) {
            // 2 x 2 taps of parity plane (qy, qx): displacement a in {0, +1} for parity 0, {-1, 0} for parity 1
            const int qy = (chunk >> 1) & 1, qx = chunk & 1;
            const int d00 = -qy * HALO_W - qx, d01 = d00 + 1, d10 = d00 + HALO_W, d11 = d10 + 1;
            TAPADDR(d00)
            RDA1(0, 0, 0) RDA1(0, 0, 1) RDA1(0, 0, 2) RDA1(0, 0, 3)
            STENCIL_STEP(d00, true, d01)
            STENCIL_STEP(d01, true, d10)
            STENCIL_STEP(d10, true, d11)
            STENCIL_STEP(d11, false, 0)
        } else {
        soff_n = chunk + 1 < cpt ? soff_c + 16u * WSTEP : soff_c;  // tail: harmless re-fetch
        TAPADDR(-HALO_W - 1)
        RDA1(0, 0, 0) RDA1(0, 0, 1) RDA1(0, 0, 2) RDA1(0, 0, 3)
        // ---- 9 stencil taps (a, b) in {-1,0,1}^2: displacement a*18 + b in the halo; 36 sub-steps, weights 5 ahead
#define D_OF(T) (((T) / 3 - 1) * HALO_W + ((T) % 3 - 1))
        SLOT(5, soff_c + 1 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 0)
        SLOT(0, soff_c + 1 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 1)
        SLOT(1, soff_c + 1 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 2)
        SLOT(2, soff_c + 2 * WSTEP, 0, true, TAPADDR(D_OF(1)), 0, 0, 1, 3)
        SLOT(3, soff_c + 2 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 4)
        SLOT(4, soff_c + 2 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 5)
        SLOT(5, soff_c + 2 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 0)
        SLOT(0, soff_c + 3 * WSTEP, 0, true, TAPADDR(D_OF(2)), 0, 0, 1, 1)
        SLOT(1, soff_c + 3 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 2)
        SLOT(2, soff_c + 3 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 3)
        SLOT(3, soff_c + 3 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 4)
        SLOT(4, soff_c + 4 * WSTEP, 0, true, TAPADDR(D_OF(3)), 0, 0, 1, 5)
        SLOT(5, soff_c + 4 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 0)
        SLOT(0, soff_c + 4 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 1)
        SLOT(1, soff_c + 4 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 2)
        SLOT(2, soff_c + 5 * WSTEP, 0, true, TAPADDR(D_OF(4)), 0, 0, 1, 3)
        SLOT(3, soff_c + 5 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 4)
        SLOT(4, soff_c + 5 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 5)
        SLOT(5, soff_c + 5 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 0)
        SLOT(0, soff_c + 6 * WSTEP, 0, true, TAPADDR(D_OF(5)), 0, 0, 1, 1)
        SLOT(1, soff_c + 6 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 2)
        SLOT(2, soff_c + 6 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 3)
        SLOT(3, soff_c + 6 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 4)
        SLOT(4, soff_c + 7 * WSTEP, 0, true, TAPADDR(D_OF(6)), 0, 0, 1, 5)
        SLOT(5, soff_c + 7 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 0)
        SLOT(0, soff_c + 7 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 1)
        SLOT(1, soff_c + 7 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 2)
        SLOT(2, soff_c + 8 * WSTEP, 0, true, TAPADDR(D_OF(7)), 0, 0, 1, 3)
        SLOT(3, soff_c + 8 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 4)
        SLOT(4, soff_c + 8 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 5)
        SLOT(5, soff_c + 8 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 0)
        SLOT(0, soff_n + 0 * WSTEP, 0, true, TAPADDR(D_OF(8)), 0, 0, 1, 1)
        SLOT(1, soff_n + 0 * WSTEP, 1, true, NO_PRE, 1, 1, 0, 2)
        SLOT(2, soff_n + 0 * WSTEP, 2, true, NO_PRE, 0, 2, 1, 3)
        SLOT(3, soff_n + 0 * WSTEP, 3, true, NO_PRE, 1, 3, 0, 4)
        SLOT(4, soff_n + 1 * WSTEP, 0, false, NO_PRE, 0, 0, 1, 5)
        soff_c = soff_n;
#undef D_OF
        }
        if (H::NBUF == 1 && chunk + 1 < cpt) {  // single buffer: every wave is done reading -> request the next chunk's halo
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(chunk + 1);
        }
    }
#undef STENCIL_STEP
#undef STEP_CORE
#undef SLOT
#undef PIN
#undef MF
#undef RDA1
#undef TAPADDR
#undef NO_PRE
#undef LGKM_WAIT
#undef DSR
#undef ISSUE_B

#ifdef HALO_EXP_NO_EPI
    if (a.n > 0) {  // timing experiment (wrong results): no epilogue; the accumulators stay live through an impossible store
        float sacc = 0.0f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) sacc += acc[nt][mt][r];
        if (sacc == 12345.678f) a.out_el16[0] = 1;
        return;
    }
#endif
    // ---- epilogue straight from the accumulators.  Lane (l31, hi) of tile (nt, mt) holds pixel l31 of pixel tile mt and
    // channels half*32 + 8*g + 4*hi + {0..3} (g = register group r >> 2).  Groups 2*g2 and 2*g2+1 are packed to bf16 and
    // exchanged between lanes l and l+32 (v_permlane32_swap), after which every lane owns 8 consecutive channels = 16 B.
    const RngKey key = drop_row_key(a.drop, n_img);
    const uint32_t row0 = (uint32_t)n_img * (uint32_t)(a.ho * a.wo * a.cout);  // dropout streams are per batch row
    // channel block of this wave: the 64 channels of column block tn (upsample forms: one phase per wave), or (plain form)
    // the wave's own 64 of the 256 channels of block tn
    const int ch_blk = H::PLAIN ? tn * H::BLK + (SP == 5 ? 0 : (SP == 4 ? wpx : wave) * 64) : tn * 64;
    const uint32_t ci_base = (uint32_t)((a.coef_div > 1 ? n_img / a.coef_div : n_img) * a.coef_stride + ch_blk + 4 * hi);
    // output pixel (pixel tile 0, px = 0) of this lane, in elements; pixel tile mt adds 4 output rows, px adds one pixel
    const int orow0 = ty0 + px_r + ((SP == 4 || SP == 5) ? 8 * wpy : 0);  // plain forms: output row of pixel tile 0 (mt adds 2)
    const uint32_t m0 = H::PLAIN ? (uint32_t)((n_img * a.ho + orow0) * a.wo + col)
                                : (uint32_t)((n_img * a.ho + 2 * (ty0 + px_r) + wpy) * a.wo + 2 * col + wpx);
    const uint32_t o0 = m0 * (uint32_t)a.cout + (uint32_t)ch_blk;
    const uint32_t mt_stride = (uint32_t)((H::PLAIN ? 2 : 4) * a.wo * a.cout);
    // sparse form: the output tensor keeps only the listed columns, [n][ho][up_wo_store][cout]; the dropout stream stays
    // indexed by the DENSE position (o0), so masks do not depend on the storage layout
    const uint32_t store0 = SP == 1 ? (uint32_t)((n_img * a.ho + 2 * (ty0 + px_r) + wpy) * a.up_wo_store + cstore) * (uint32_t)a.cout +
                                 (uint32_t)(tn * 64)
                           : o0;
    const uint32_t smt_stride = SP == 1 ? (uint32_t)(4 * a.up_wo_store * a.cout) : mt_stride;
    if constexpr (EPI == 2) {
        TL_STAMP(2)
        // ---- GroupNorm fused (gn_fused.h).  Phase A: (sum, sum of squares) of y = acc + bias per 8-channel octet over this wave's
        // 128 pixels (pixels beyond a ragged plane masked by a 0 / 1 factor), reduce-scatter butterfly, 16 granules per wave.
        const GnFuse& G = a.gnf;
        const uint32_t tag = (*G.epoch << 8) | G.conv_tag;
        {
            float mval[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) mval[mt] = (lane_valid && orow0 + 2 * mt < a.ho) ? 1.0f : 0.0f;
            float w[16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 b4 = *(const float4*)(G.bias + ch_blk + nt * 32 + 8 * g + 4 * hi);
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float y0 = acc[nt][mt][4 * g + 0] + b4.x, y1 = acc[nt][mt][4 * g + 1] + b4.y;
                        const float y2 = acc[nt][mt][4 * g + 2] + b4.z, y3 = acc[nt][mt][4 * g + 3] + b4.w;
                        s1 = fmaf(mval[mt], (y0 + y1) + (y2 + y3), s1);
                        s2 = fmaf(mval[mt], fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, y3 * y3))), s2);
                    }
                    w[2 * (4 * nt + g)] = s1;
                    w[2 * (4 * nt + g) + 1] = s2;
                }
#pragma unroll
            for (int half = 8, d = 1; half >= 1; half >>= 1, d <<= 1) {
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
                const int slot = t_in * NWAVES + wave;
                gn_store_granule(G.gran + (((size_t)n_img * G.max_slots + slot) * (a.cout >> 3) + tn * 8) * 2 + idx, tag, tot);
            }
        }
        // residual (the ResnetBlock's shortcut, added last): two 8-byte pieces per (nt, mt, g2) step, fetched one (nt, mt) group AHEAD
        // of its use -- a load issued behind a store waits for that store's acknowledgement (gfx9 counts stores in vmcnt), so the
        // loads of group s + 1 go out before the stores of group s; the first group goes out here, under the sweep
        const bool has_res = a.residual != nullptr;
        uint2 rq[2][4];
        auto load_res = [&](int nt, int mt, uint2 (&r)[4]) {
            const bool st_ok = lane_valid && orow0 + 2 * mt < a.ho;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const size_t e0 = (size_t)(o0 + mt * mt_stride + nt * 32 + 16 * g2 + 4 * hi);
                r[2 * g2] = st_ok ? *(const uint2*)(a.residual + e0) : make_uint2(0, 0);
                r[2 * g2 + 1] = st_ok ? *(const uint2*)(a.residual + e0 + 8) : make_uint2(0, 0);
            }
        };
        if (has_res) load_res(0, 0, rq[0]);
        TL_STAMP(3)
        // Phase B: wave 0 sweeps the sample's granules and parks (A, C) of the block's 64 channels in LDS (its own 512 bytes behind
        // the halo: the other waves may still be in their K loop)
        float* cfA = (float*)(smem + H::LDS_TOTAL);
        float* cfC = cfA + 64;
        if (wave == 0) {
            const int cpg = a.cout / G.groups;
            const float2 mr = gn_fuse_sweep<16>(G.gran + ((size_t)n_img * G.max_slots * (a.cout >> 3) + tn * 8) * 2, (a.cout >> 3) * 2,
                                                G.slots, tag ^ G.test_tag_xor, cpg, 1.0 / ((double)a.ho * a.wo * cpg), G.err, lane,
                                                G.timeout_ticks);
            const float2 ac = gn_fuse_coef(G, ch_blk + lane, a.coef_div > 1 ? n_img / a.coef_div : n_img, mr);
            cfA[lane] = ac.x;
            cfC[lane] = ac.y;
            TL_STAMP(6)
        }
        __syncthreads();
        TL_STAMP(4)
        // Phase C: y * A + C -> SiLU -> dropout -> (+ residual) -> 16-bit, stored as the plain epilogue stores
        auto fused = [&](auto mode_c) {
            constexpr int MODE = decltype(mode_c)::value;
#pragma unroll
            for (int nt = 0; nt < 2; ++nt) {
                float ca[2][8], cc[2][8];
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int cg0 = nt * 32 + 16 * g2 + 4 * hi;
                    const float4 a0 = *(const float4*)(cfA + cg0), a1 = *(const float4*)(cfA + cg0 + 8);
                    const float4 c0 = *(const float4*)(cfC + cg0), c1 = *(const float4*)(cfC + cg0 + 8);
                    ca[g2][0] = a0.x; ca[g2][1] = a0.y; ca[g2][2] = a0.z; ca[g2][3] = a0.w;
                    ca[g2][4] = a1.x; ca[g2][5] = a1.y; ca[g2][6] = a1.z; ca[g2][7] = a1.w;
                    cc[g2][0] = c0.x; cc[g2][1] = c0.y; cc[g2][2] = c0.z; cc[g2][3] = c0.w;
                    cc[g2][4] = c1.x; cc[g2][5] = c1.y; cc[g2][6] = c1.z; cc[g2][7] = c1.w;
                }
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    const int s = nt * 4 + mt;
                    if (has_res && s + 1 < 8) load_res((s + 1) >> 2, (s + 1) & 3, rq[(s + 1) & 1]);
                    const bool st_ok = lane_valid && orow0 + 2 * mt < a.ho;
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int cg0 = nt * 32 + 16 * g2;
                        const uint32_t e0 = o0 + mt * mt_stride + cg0 + 4 * hi;
                        float v[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = fmaf(acc[nt][mt][8 * g2 + t], ca[g2][t], cc[g2][t]);
                        act_drop_fixed<4, ACT_SILU, MODE, true>(v, e0, row0, a.drop, key);
                        act_drop_fixed<4, ACT_SILU, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                        if (has_res) {
                            const uint2 r0 = rq[s & 1][2 * g2], r1 = rq[s & 1][2 * g2 + 1];
                            const uint32_t rw[4] = {r0.x, r0.y, r1.x, r1.y};
#pragma unroll
                            for (int t = 0; t < 8; ++t) v[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
                        }
                        uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                        uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        if (st_ok) *(uint4*)(a.out_el16 + (size_t)(o0 + mt * mt_stride + cg0 + 8 * hi)) = o;
                    }
                }
            }
        };
        if (a.drop.mode == 1) fused(std::integral_constant<int, 1>{});
        else fused(std::integral_constant<int, 0>{});
        TL_STAMP(5)
        return;
    }
    // (activation, dropout mode) are wave-uniform: the whole epilogue is instantiated per pair and dispatched once
    auto epilogue = [&](auto act_c, auto mode_c) {
        constexpr int ACT = decltype(act_c)::value, MODE = decltype(mode_c)::value;
        const float ps = drop_prescale<ACT, MODE>(a.drop);  // dropout scale folded into the affine
        if constexpr (H::STAGE) {
            // stride-2 forms: a pixel tile (2 rows x 16 pixels) of this wave is 32 whole 128-byte lines (its 64 channels of each
            // pixel); through the per-wave LDS staging tile a store instruction writes 8 of them (lane = (pixel, 16-byte chunk))
            // instead of a 32-byte piece of 32 lines
            unsigned char* ost = (unsigned char*)smem + H::TAB_END + wave * (32 * H::OROW);
            float ca[2][2][8], cc[2][2][8];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int cg0 = nt * 32 + 16 * g2;
                    const float4 ca0 = *(const float4*)(a.coef_a + ci_base + cg0), ca1 = *(const float4*)(a.coef_a + ci_base + cg0 + 8);
                    const float4 cc0 = *(const float4*)(a.coef_c + ci_base + cg0), cc1 = *(const float4*)(a.coef_c + ci_base + cg0 + 8);
                    ca[nt][g2][0] = ca0.x * ps; ca[nt][g2][1] = ca0.y * ps; ca[nt][g2][2] = ca0.z * ps; ca[nt][g2][3] = ca0.w * ps;
                    ca[nt][g2][4] = ca1.x * ps; ca[nt][g2][5] = ca1.y * ps; ca[nt][g2][6] = ca1.z * ps; ca[nt][g2][7] = ca1.w * ps;
                    cc[nt][g2][0] = cc0.x * ps; cc[nt][g2][1] = cc0.y * ps; cc[nt][g2][2] = cc0.z * ps; cc[nt][g2][3] = cc0.w * ps;
                    cc[nt][g2][4] = cc1.x * ps; cc[nt][g2][5] = cc1.y * ps; cc[nt][g2][6] = cc1.z * ps; cc[nt][g2][7] = cc1.w * ps;
                }
            const uint32_t tile_base = o0 - (uint32_t)(px_r * a.wo + px_x) * (uint32_t)a.cout;  // pixel (row 0, column 0) of pixel tile 0
            const int rpx = lane >> 3, rch = lane & 7;  // read-back role: (pixel 8 k + rpx, 16-byte chunk)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int cg0 = nt * 32 + 16 * g2;
                        const uint32_t e0 = o0 + mt * mt_stride + cg0 + 4 * hi;
                        float v[8];
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = fmaf(acc[nt][mt][8 * g2 + t], ca[nt][g2][t], cc[nt][g2][t]);
                        act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                        act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                        uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                        uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        *(uint4*)(ost + l31 * H::OROW + (cg0 + 8 * hi) * 2) = o;
                    }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int px = 8 * k + rpx;
                    const uint4 o = *(const uint4*)(ost + px * H::OROW + rch * 16);
#ifdef DYF_NT_STORES  // experiment: non-temporal stores
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), (u32x4*)(a.out_el16 + (size_t)(tile_base + mt * mt_stride + (uint32_t)((px >> 4) * a.wo + (px & 15)) * (uint32_t)a.cout + rch * 8)));
#else
                    *(uint4*)(a.out_el16 + (size_t)(tile_base + mt * mt_stride + (uint32_t)((px >> 4) * a.wo + (px & 15)) * (uint32_t)a.cout + rch * 8)) = o;
#endif
                }
            }
            return;
        }
        // 32-channel half outermost, pixel tiles, then the two 16-channel groups of the half: the two 32-byte pieces of a
        // pixel's 64-byte half block are stored back to back and leave the L2 as whole 64-byte writes (with the channel groups
        // outermost PMC counted 1.2-1.65x the algorithmic write bytes)
        // SP = 5 feeding a GroupNorm (a.gn_part; the launcher asks only with act = none and no dropout: one instantiation carries
        // the code): a pass of its own over the accumulators, BEFORE the store loop -- per lane (sum, sum of squares) of
        // y = acc * A + C for the 8 channel octets of the wave's 64 channels (a lane holds channels 8*o + 4*hi + {0..3} of octet
        // o = 4*nt + g), pixels outside the image masked by a 0 / 1 factor (no branches).  The coefficients pass through an opaque
        // copy so that the compiler does not merge this pass with the store loop below and keep 128 products alive (it spilled).
        constexpr bool STATS = SP == 5 && ACT == ACT_NONE && MODE == 0;
        // SP = 5: the 16 coefficient loads of the wave's 64 channels go out once, together, for the statistics pass AND the store loop
        float4 k4a[SP == 5 ? 8 : 1], k4c[SP == 5 ? 8 : 1];
        if constexpr (SP == 5) {
#pragma unroll
            for (int o = 0; o < 8; ++o) {
                k4a[o] = *(const float4*)(a.coef_a + ci_base + (o >> 2) * 32 + 8 * (o & 3));
                k4c[o] = *(const float4*)(a.coef_c + ci_base + (o >> 2) * 32 + 8 * (o & 3));
            }
        }
#ifdef H5_LEAN  // EXPERIMENT, measured slower and off (see the end of this comment)
        // SP = 5 without activation / dropout (every conv of the ResNet-UNet that feeds a GroupNorm, and its plain 3x3 convs):
        // y = acc * A + C is formed ONCE, in place, with packed fp32 math (a lane's channel pairs are register pairs), octet by octet
        // with the octet's coefficients loaded there (all 16 loads up front spilled); pixels outside a ragged plane are zeroed (they
        // are not stored and then add nothing to the sums); the sums of the octet follow at once, the store loop reads y.
        // PMC had counted 2 461 vector instructions per wave against 384 MFMAs, ~1 100 of them in this epilogue (y formed twice --
        // once per pass, behind an opaque copy that kept the two passes apart --, masked scalar sums, register copies).
        // Measured (level-0 64 -> 64 convs of the OISST rollout, 300 rows): 116.5 us without it, 152 us with it (228 us in its first
        // form with the 16 coefficient loads up front): the kernel already sits at 256 registers with 5 spill slots; in the full
        // kernel (all activation x dropout epilogues instantiated) this path comes out with 52-199 spill slots -- ~230 MB of scratch
        // traffic per launch -- although it allocates 250 registers and no spill when it is the only epilogue (PLAIN_EPI; there: 119.8 us against 114.7 us).  The
        // instruction count does drop (epilogue ~930 -> ~620 VALU per wave); what it needs is register headroom first.
        if constexpr (STATS) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 m2[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float m = (lane_valid && orow0 + 2 * mt < a.ho) ? 1.0f : 0.0f;
                m2[mt] = f32x2{m, m};
            }
            const bool stats = a.gn_part != nullptr;
            float w[16];
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ka = *(const float4*)(a.coef_a + ci_base + nt * 32 + 8 * g), kc = *(const float4*)(a.coef_c + ci_base + nt * 32 + 8 * g);
                    const f32x2 k01 = {ka.x, ka.y}, k23 = {ka.z, ka.w}, c01 = {kc.x, kc.y}, c23 = {kc.z, kc.w};
                    f32x2 s1a = {0.0f, 0.0f}, s1b = {0.0f, 0.0f}, s2a = {0.0f, 0.0f}, s2b = {0.0f, 0.0f};  // two chains each: packed ops have latency
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        f32x2 y01 = {acc[nt][mt][4 * g + 0], acc[nt][mt][4 * g + 1]}, y23 = {acc[nt][mt][4 * g + 2], acc[nt][mt][4 * g + 3]};
                        y01 = (y01 * k01 + c01) * m2[mt];
                        y23 = (y23 * k23 + c23) * m2[mt];
                        acc[nt][mt][4 * g + 0] = y01.x; acc[nt][mt][4 * g + 1] = y01.y;
                        acc[nt][mt][4 * g + 2] = y23.x; acc[nt][mt][4 * g + 3] = y23.y;
                        s1a += y01;
                        s1b += y23;
                        s2a = y01 * y01 + s2a;
                        s2b = y23 * y23 + s2b;
                    }
                    s1a += s1b;
                    s2a += s2b;
                    w[2 * (4 * nt + g)] = s1a.x + s1a.y;
                    w[2 * (4 * nt + g) + 1] = s2a.x + s2a.y;
                    __builtin_amdgcn_sched_barrier(0);  // octet by octet: hoisted together the 64 packed products got fresh registers and spilled
                }
            if (stats) {
                // wave reduction: the reduce-scatter butterfly of the general path below
#pragma unroll
                for (int half = 8, d = 1; half >= 1; half >>= 1, d <<= 1) {
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
                    const int slot = t_in * NWAVES + wave;
                    a.gn_part[((size_t)(n_img * a.gn_slots + slot) * (a.cout >> 3) + tn * 8) * 2 + idx] = tot;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const uint32_t p0 = pack_el16x2(acc[nt][mt][8 * g2 + 0], acc[nt][mt][8 * g2 + 1]), p1 = pack_el16x2(acc[nt][mt][8 * g2 + 2], acc[nt][mt][8 * g2 + 3]);
                        const uint32_t q0 = pack_el16x2(acc[nt][mt][8 * g2 + 4], acc[nt][mt][8 * g2 + 5]), q1 = pack_el16x2(acc[nt][mt][8 * g2 + 6], acc[nt][mt][8 * g2 + 7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        const uint32_t sbase = store0 + mt * smt_stride + nt * 32 + 16 * g2;
#ifdef HALO_EXP_NO_STORE
                        if (a.n < 0)
#endif
                        if (lane_valid && orow0 + 2 * mt < a.ho) *(uint4*)(a.out_el16 + (size_t)(sbase + 8 * hi)) = o;
                    }
            return;
        }
#endif
        if constexpr (STATS) if (a.gn_part != nullptr) {
            float mval[4];
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) mval[mt] = (lane_valid && orow0 + 2 * mt < a.ho) ? 1.0f : 0.0f;
            float w[16];
            // y is formed with an opaque factor 1.0 so that the compiler does not merge this pass with the store loop below and keep
            // 128 products alive across it (it spilled)
            float one = 1.0f;
            asm volatile("" : "+v"(one));
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const float4 ka = k4a[4 * nt + g], kc = k4c[4 * nt + g];
                    const float kx = ka.x * one, ky = ka.y * one, kz = ka.z * one, kw = ka.w * one;
                    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        const float y0 = fmaf(acc[nt][mt][4 * g + 0], kx, kc.x), y1 = fmaf(acc[nt][mt][4 * g + 1], ky, kc.y);
                        const float y2 = fmaf(acc[nt][mt][4 * g + 2], kz, kc.z), y3 = fmaf(acc[nt][mt][4 * g + 3], kw, kc.w);
                        s1 = fmaf(mval[mt], (y0 + y1) + (y2 + y3), s1);
                        s2 = fmaf(mval[mt], fmaf(y0, y0, fmaf(y1, y1, fmaf(y2, y2, y3 * y3))), s2);
                    }
                    w[2 * (4 * nt + g)] = s1;
                    w[2 * (4 * nt + g) + 1] = s2;
                }
            // wave reduction of the 16 values as a reduce-scatter butterfly: at stride d the lane keeps the half of its values its
            // bit selects and adds the partner's copy of that half (8 + 4 + 2 + 1 exchanges), then two plain exchanges over the
            // remaining lane bits: 17 cross-lane moves instead of 96.  Lane L < 16 ends with value index
            // 8*(L&1) + 4*((L>>1)&1) + 2*((L>>2)&1) + ((L>>3)&1), index = 2*octet + {0: sum, 1: sum of squares}.
#pragma unroll
            for (int half = 8, d = 1; half >= 1; half >>= 1, d <<= 1) {
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
                const int slot = t_in * NWAVES + wave;
                a.gn_part[((size_t)(n_img * a.gn_slots + slot) * (a.cout >> 3) + tn * 8) * 2 + idx] = tot;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float ca[2][8], cc[2][8];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int cg0 = nt * 32 + 16 * g2;  // + 4*hi: own channels of group 2*g2; + 8: group 2*g2+1
                float4 ca0, ca1, cc0, cc1;
                if constexpr (SP == 5) {
                    ca0 = k4a[4 * nt + 2 * g2]; ca1 = k4a[4 * nt + 2 * g2 + 1];
                    cc0 = k4c[4 * nt + 2 * g2]; cc1 = k4c[4 * nt + 2 * g2 + 1];
                } else {
                    ca0 = *(const float4*)(a.coef_a + ci_base + cg0); ca1 = *(const float4*)(a.coef_a + ci_base + cg0 + 8);
                    cc0 = *(const float4*)(a.coef_c + ci_base + cg0); cc1 = *(const float4*)(a.coef_c + ci_base + cg0 + 8);
                }
                ca[g2][0] = ca0.x * ps; ca[g2][1] = ca0.y * ps; ca[g2][2] = ca0.z * ps; ca[g2][3] = ca0.w * ps;
                ca[g2][4] = ca1.x * ps; ca[g2][5] = ca1.y * ps; ca[g2][6] = ca1.z * ps; ca[g2][7] = ca1.w * ps;
                cc[g2][0] = cc0.x * ps; cc[g2][1] = cc0.y * ps; cc[g2][2] = cc0.z * ps; cc[g2][3] = cc0.w * ps;
                cc[g2][4] = cc1.x * ps; cc[g2][5] = cc1.y * ps; cc[g2][6] = cc1.z * ps; cc[g2][7] = cc1.w * ps;
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int cg0 = nt * 32 + 16 * g2;
                    const uint32_t obase = o0 + mt * mt_stride + cg0;
                    const uint32_t e0 = obase + 4 * hi;
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = fmaf(acc[nt][mt][8 * g2 + t], ca[g2][t], cc[g2][t]);
                    act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                    act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                    uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                    uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                    // lanes 0-31: {own group 2*g2, partner's group 2*g2} = channels cg0 + 0..7;
                    // lanes 32-63: {partner's group 2*g2+1, own group 2*g2+1} = channels cg0 + 8..15
                    uint4 o;
                    o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                    const uint32_t sbase = store0 + mt * smt_stride + cg0;
#ifdef HALO_EXP_NO_STORE
                    if (a.n < 0)  // timing experiment (wrong results): the epilogue computes, nothing is stored
#endif
                    if ((SP != 1 && SP != 5) || (lane_valid && (SP != 5 || orow0 + 2 * mt < a.ho)))
                        *(uint4*)(a.out_el16 + (size_t)(sbase + 8 * hi)) = o;
                }
            }
        }
    };
    auto by_mode = [&](auto act_c) {
        if (a.drop.mode == 0) epilogue(act_c, std::integral_constant<int, 0>{});
        else if (a.drop.mode == 1) epilogue(act_c, std::integral_constant<int, 1>{});
        else epilogue(act_c, std::integral_constant<int, 2>{});
    };
    if constexpr (PLAIN_EPI) {
        epilogue(std::integral_constant<int, ACT_NONE>{}, std::integral_constant<int, 0>{});
        return;
    }
    if (a.act == ACT_RELU) by_mode(std::integral_constant<int, ACT_RELU>{});
    else if (a.act == ACT_LEAKY) by_mode(std::integral_constant<int, ACT_LEAKY>{});
    else if (a.act == ACT_SILU) by_mode(std::integral_constant<int, ACT_SILU>{});
    else by_mode(std::integral_constant<int, ACT_NONE>{});
#endif
}

// [4 phases][cout][16 taps][cin] (pack_up2x_weights) -> MFMA fragment order:
// [column block tn][chunk][tap][wn = py][ks][column tile nt = px*2 + half][lane][8 k]; lane (l31, hi) of fragment
// (ks, nt) holds channel tn*64 + half*32 + l31, k = chunk*64 + ks*16 + hi*8 + {0..7}
void pack_up2x_frag(const el16_t* wpk_up, int cout, int cin, el16_t* out) {
    const int cpt = cin / 64;
    size_t o = 0;
    for (int tn = 0; tn < cout / 64; ++tn)
        for (int chunk = 0; chunk < cpt; ++chunk)
            for (int tap = 0; tap < 16; ++tap)
                for (int wn = 0; wn < 2; ++wn)
                    for (int ks = 0; ks < 4; ++ks)
                        for (int nt = 0; nt < 4; ++nt)
                            for (int lane = 0; lane < 64; ++lane) {
                                const int phase = wn * 2 + (nt >> 1);
                                const int co = tn * 64 + (nt & 1) * 32 + (lane & 31);
                                const int k0 = chunk * 64 + ks * 16 + (lane >> 5) * 8;
                                const el16_t* s = wpk_up + (((size_t)phase * cout + co) * 16 + tap) * cin + k0;
                                for (int e = 0; e < 8; ++e) out[o++] = s[e];
                            }
}

bool conv_up_halo_supported(const ConvArgs& a) {
    if (!a.up2x || a.wpk_up_frag == nullptr || a.out_el16 == nullptr || a.residual != nullptr) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && (a.c1 == 0 || a.c1 == a.c0) && a.cout % 64 == 0)) return false;
    if (a.h % TILE_H != 0 || a.w % TILE_W != 0 || a.ho != 2 * a.h || a.wo != 2 * a.w) return false;
    const bool mixed = (a.up_mix[0] | a.up_mix[1] | a.up_mix[2]) != 0;
    if (a.up_cols && (a.up_cbase == nullptr || a.up_cidx == nullptr || a.up_wo_store < 1 || a.up_ntiles < 1 ||
                      (!mixed && a.up_npad != a.up_ntiles * 16 && a.up_npad != a.up_ntiles * conv_halo_rows_slots())))
        return false;
    if (a.up_cols && (mixed || a.up_npad != a.up_ntiles * 16) && !conv_halo_rows_up_supported(a)) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && (size_t)4 * a.cout * 16 * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

// Plain 3x3 conv through the halo kernel (SP = 2): wpk [cout][9][cin] -> the fragment order of pack_up2x_frag with the four
// "phases" being the four 64-channel blocks of every 256 output channels (taps 9-15 of the 16-tap axis stay zero).
void pack_halo3_frag(const el16_t* wpk, int cout, int cin, el16_t* out) {
    const int blocks = cout / 256;
    std::vector<el16_t> v((size_t)4 * blocks * 64 * 16 * cin, 0);  // [phase][co' = blk*64 + c][16][cin]
    for (int p = 0; p < 4; ++p)
        for (int b = 0; b < blocks; ++b)
            for (int c = 0; c < 64; ++c)
                for (int t = 0; t < 9; ++t) {
                    const el16_t* src = wpk + ((size_t)(b * 256 + p * 64 + c) * 9 + t) * cin;
                    el16_t* dst = v.data() + ((((size_t)p * blocks * 64 + b * 64 + c) * 16) + t) * cin;
                    std::copy(src, src + cin, dst);
                }
    pack_up2x_frag(v.data(), blocks * 64, cin, out);
}

// 4x4 / stride 2 / pad 1 conv through the halo kernel (SP = 3): wpk [cout][16][cin] -> the fragment order of pack_up2x_frag
// over the virtual K axis [64-channel chunk c][parity plane p = qy*2 + qx][64], "tap" slot t = ty*2 + tx of the 16-tap axis
// holding kernel tap (ky, kx) = (2*a(qy, ty) + qy + 1, 2*a(qx, tx) + qx + 1), a(q, t) = t - q; slots 4-15 stay zero.
void pack_halo_s2_frag(const el16_t* wpk, int cout, int cin, el16_t* out) {
    if (cout % 256 != 0) {
        // SP = 4 (128-channel workgroups): [tn][virtual chunk = c*4 + plane][16 slots][ks][wn][nt][lane][8 k]
        const int cpt = cin / 64;
        size_t o = 0;
        for (int tn = 0; tn < cout / 128; ++tn)
            for (int ch = 0; ch < cpt; ++ch)
                for (int pl = 0; pl < 4; ++pl)
                    for (int t = 0; t < 16; ++t)
                        for (int ks = 0; ks < 4; ++ks)
                            for (int wn = 0; wn < 2; ++wn)
                                for (int nt = 0; nt < 2; ++nt)
                                    for (int lane = 0; lane < 64; ++lane) {
                                        const int qy = pl >> 1, qx = pl & 1;
                                        const int ky = 2 * ((t >> 1) - qy) + qy + 1, kx = 2 * ((t & 1) - qx) + qx + 1;
                                        const int co = tn * 128 + wn * 64 + nt * 32 + (lane & 31);
                                        const int k0 = ch * 64 + ks * 16 + (lane >> 5) * 8;
                                        const el16_t* s = wpk + ((size_t)co * 16 + ky * 4 + kx) * cin + k0;
                                        for (int e = 0; e < 8; ++e) out[o++] = t < 4 ? s[e] : (el16_t)0;
                                    }
        return;
    }
    const int blocks = cout / 256, cpt = cin / 64, cin4 = 4 * cin;
    std::vector<el16_t> v((size_t)4 * blocks * 64 * 16 * cin4, 0);  // [block-of-64 index][co][16][4*cin]
    for (int p4 = 0; p4 < 4; ++p4)
        for (int b = 0; b < blocks; ++b)
            for (int c = 0; c < 64; ++c) {
                const int co = b * 256 + p4 * 64 + c;
                for (int ch = 0; ch < cpt; ++ch)
                    for (int pl = 0; pl < 4; ++pl) {
                        const int qy = pl >> 1, qx = pl & 1;
                        for (int t = 0; t < 4; ++t) {
                            const int ky = 2 * ((t >> 1) - qy) + qy + 1, kx = 2 * ((t & 1) - qx) + qx + 1;
                            const el16_t* src = wpk + ((size_t)co * 16 + ky * 4 + kx) * cin + ch * 64;
                            el16_t* dst = v.data() + ((((size_t)p4 * blocks * 64 + b * 64 + c) * 16) + t) * cin4 + (ch * 4 + pl) * 64;
                            std::copy(src, src + 64, dst);
                        }
                    }
            }
    pack_up2x_frag(v.data(), blocks * 64, cin4, out);
}

// Plain 3x3 conv with a multiple of 64 output channels (SP = 5): wpk [cout][9][cin] ->
// [column block tn][chunk][16 tap slots][ks][column tile nt][lane][8 k]; lane (l31, hi) of fragment (ks, nt) holds channel
// tn*64 + nt*32 + l31, k = chunk*64 + ks*16 + hi*8 + {0..7}; slots 9-15 stay zero.
void pack_halo3_frag64(const el16_t* wpk, int cout, int cin, el16_t* out) {
    const int cpt = cin / 64;
    size_t o = 0;
    for (int tn = 0; tn < cout / 64; ++tn)
        for (int chunk = 0; chunk < cpt; ++chunk)
            for (int tap = 0; tap < 16; ++tap)
                for (int ks = 0; ks < 4; ++ks)
                    for (int nt = 0; nt < 2; ++nt)
                        for (int lane = 0; lane < 64; ++lane) {
                            const int co = tn * 64 + nt * 32 + (lane & 31);
                            const int k0 = chunk * 64 + ks * 16 + (lane >> 5) * 8;
                            const el16_t* sp = wpk + ((size_t)co * 9 + tap) * cin + k0;
                            for (int e2 = 0; e2 < 8; ++e2) out[o++] = tap < 9 ? sp[e2] : (el16_t)0;
                        }
}

bool conv_halo5_supported(const ConvArgs& a) {
    // (a residual is only added by the fused-GroupNorm epilogue, EPI = 2)
    if (a.up2x || a.wpk_up_frag == nullptr || a.out_el16 == nullptr || a.out_f32 != nullptr || (a.residual != nullptr && a.gnf.gran == nullptr)) return false;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.pix_pitch0 != 0) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && (a.c1 == 0 || a.c1 == a.c0) && a.cout % 64 == 0)) return false;
    if (a.ho != a.h || a.wo != a.w) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && (size_t)a.cout * 16 * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

int conv_halo5_gn_slots(int h, int w) {
    using H5 = HaloCfg<5>;
    return ((w + H5::TW - 1) / H5::TW) * ((h + H5::TH - 1) / H5::TH) * NWAVES;
}

hipError_t launch_conv_halo5(const ConvArgs& a, hipStream_t stream) {
    using H5 = HaloCfg<5>;
    const int tiles_x = (a.w + H5::TW - 1) / H5::TW, tiles_per_img = tiles_x * ((a.h + H5::TH - 1) / H5::TH);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 64;
    dyf_form_note(a.gnf.gran ? "conv_up_halo_kernel<5>+gn_fused" : "conv_up_halo_kernel<5>", a.n);
    if (a.up_nearest) dyf_form_note("conv_up_halo_kernel<5>+nearest_up", a.n);
    const bool plain_epi = !(dyf_form("DYF_HALO5_PLAIN_EPI") && atoi(dyf_form("DYF_HALO5_PLAIN_EPI")) == 0);
    if (a.gnf.gran != nullptr) {  // GroupNorm fused: + 1 KB of LDS for the per-channel (A, C) table
        ConvArgs b = a;
#ifdef DYF_EXPERIMENT_BUILD
        if (dyf_form("DYF_GN_FUSE_NOWAIT")) b.gnf.slots = 0;  // timing experiment (WRONG results): no granule sweep
#endif
        hipLaunchKernelGGL((conv_up_halo_kernel<5, 2>), dim3(tiles_m * tiles_n), dim3(256), H5::LDS_TOTAL + 1024, stream, b, tiles_x,
                           tiles_per_img, tiles_m, tiles_n, 0);
#ifdef HALO_EXP_TIMELINE
        if (const char* tl = dyf_form("DYF_TIMELINE_DUMP")) {  // "path:N": the stamps of the N-th fused launch of the process (eager launches only)
            static int count = 0;
            const char* colon = strrchr(tl, ':');
            if (colon && ++count == atoi(colon + 1)) {
                (void)hipStreamSynchronize(stream);
                std::vector<unsigned long long> h((size_t)1 << 18);
                (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_halo_tl), h.size() * 8);
                if (FILE* f = fopen(std::string(tl, colon - tl).c_str(), "wb")) {
                    const int hdr[4] = {tiles_m * tiles_n, a.n, a.residual != nullptr, a.drop.mode};
                    fwrite(hdr, sizeof(int), 4, f);
                    fwrite(h.data(), 8, (size_t)std::min(tiles_m * tiles_n, 8192) * 32, f);
                    fclose(f);
                }
            }
        }
#endif
    }
    else if (plain_epi && a.act == ACT_NONE && a.drop.mode == 0)
        hipLaunchKernelGGL((conv_up_halo_kernel<5, 1>), dim3(tiles_m * tiles_n), dim3(256), H5::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n, 0);
    else
        hipLaunchKernelGGL(conv_up_halo_kernel<5>, dim3(tiles_m * tiles_n), dim3(256), H5::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n, 0);
    return hipGetLastError();
}

bool conv_halo_s2_supported(const ConvArgs& a) {
    if (a.up2x || a.wpk_up_frag == nullptr || a.out_el16 == nullptr || a.out_f32 != nullptr || a.residual != nullptr) return false;
    if (a.kh != 4 || a.kw != 4 || a.stride != 2 || a.pad != 1 || a.pix_pitch0 != 0) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && a.c1 == 0 && a.cout % 128 == 0)) return false;
    const int th = a.cout % 256 == 0 ? TILE_H : HaloCfg<4>::TH;
    if (a.h % 2 != 0 || a.w % 2 != 0 || a.ho != a.h / 2 || a.wo != a.w / 2 || a.ho % th != 0 || a.wo % TILE_W != 0) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && (size_t)a.cout * 16 * 4 * a.c0 * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t launch_conv_halo_s2(const ConvArgs& a, hipStream_t stream) {
    if (a.cout % 256 != 0) {
        const int tiles_x = a.wo / TILE_W, tiles_per_img = tiles_x * (a.ho / HaloCfg<4>::TH);
        const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 128;
        dyf_form_note("conv_up_halo_kernel<4>", a.n);
        hipLaunchKernelGGL(conv_up_halo_kernel<4>, dim3(tiles_m * tiles_n), dim3(256), HaloCfg<4>::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n, 0);
        return hipGetLastError();
    }
    const int tiles_x = a.wo / TILE_W, tiles_per_img = tiles_x * (a.ho / TILE_H);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 256;
    dyf_form_note("conv_up_halo_kernel<3>", a.n);
    hipLaunchKernelGGL(conv_up_halo_kernel<3>, dim3(tiles_m * tiles_n), dim3(256), HaloCfg<3>::LDS_TOTAL, stream, a, tiles_x,
                       tiles_per_img, tiles_m, tiles_n, 0);
    return hipGetLastError();
}

bool conv_halo3_supported(const ConvArgs& a) {
    if (a.up2x || a.wpk_up_frag == nullptr || a.out_el16 == nullptr || a.out_f32 != nullptr || a.residual != nullptr) return false;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.pix_pitch0 != 0) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && (a.c1 == 0 || a.c1 == a.c0) && a.cout % 256 == 0)) return false;
    if (a.h % TILE_H != 0 || a.w % TILE_W != 0 || a.ho != a.h || a.wo != a.w) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && (size_t)a.cout * 16 * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t launch_conv_halo3(const ConvArgs& a, hipStream_t stream) {
    const int tiles_x = a.w / TILE_W, tiles_per_img = tiles_x * (a.h / TILE_H);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 256;
    dyf_form_note("conv_up_halo_kernel<2>", a.n);
    hipLaunchKernelGGL(conv_up_halo_kernel<2>, dim3(tiles_m * tiles_n), dim3(256), HaloCfg<2>::LDS_TOTAL, stream, a, tiles_x,
                       tiles_per_img, tiles_m, tiles_n, 0);
    return hipGetLastError();
}

constexpr int UB_LDS = 4 * 2 * 8192;  // up_border_kernel: per wave two stages of [2 samples][32 pixels][64 channels] bf16
__global__ void up_border_kernel(ConvArgs a, int tr, int tc);
template <int NW, bool KSPLIT>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void up_border_split_kernel(ConvArgs a, int tr, int tc);  // (the bounds must sit on the FIRST declaration: the template is instantiated from here)

hipError_t conv_up_halo_init() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       HaloCfg<0>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<1>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<2>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<3>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<4>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_up_halo_kernel<5>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<5>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_up_halo_kernel<5, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<5>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_up_halo_kernel<5, 2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                HaloCfg<5>::LDS_TOTAL + 1024);
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)up_border_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, UB_LDS);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(up_border_split_kernel<8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 16384);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(up_border_split_kernel<4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * 16384);
    return e;
}

// Border corrections of the fused x2-upsample conv (see conv.hip / pack_up2x_weights: taps 9-11 on the first / last output
// row, 12-14 on the first / last output column, 15 on the corners), evaluated for the ring of border OUTPUT pixels only and
// left in ConvArgs::up_border (fp32) as the starting value of those pixels' accumulators.  A small GEMM (3 taps x cin per
// pixel, 7 on the corners) whose time is memory latency, so it is organised around that:
//   * both MFMA operands come straight from global memory: weight fragments from the halo kernel's own stream (its tap slots
//     9-15: one coalesced 1 KB load per fragment), lane (pixel, hi) 16 B of its NHWC input pixel;
//   * the ring is walked in 8 segments (top / bottom row x px, left / right column x py) so that the 32 pixels of a wave share
//     their phase; a workgroup = one 32-pixel tile x 8 samples (4 waves x 2 samples: the waves stream the same weights at the
//     same time and meet in L1); the loads of iteration i+1 (64 channels of one tap) are issued before the MFMAs of i;
//   * the four corner pixels need 7 taps: they get workgroups of their own with the SAMPLES as MFMA columns (64 per
//     workgroup), the taps dealt to the four waves and summed through LDS -- no workgroup runs a longer chain than the rest.
struct UbStage {
    uint4 wa[4], wb[4];
};
__global__ __launch_bounds__(256) void up_border_kernel(ConvArgs a, int tr, int tc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char ub_smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb_blk = blockIdx.z;
    const int nseg = 4 * tr + 4 * tc;
    const bool cornerwg = (int)blockIdx.x >= nseg;
    if (cornerwg && (blockIdx.y & 7) != 0) return;  // a corner workgroup covers 64 samples
    const int nbase = blockIdx.y * 8;
    int py, px, i0 = 0, j0 = 0, tt = 0;
    bool rowseg = true;
    if (cornerwg) {
        const int c = blockIdx.x - nseg;
        py = c >> 1; px = c & 1;
        i0 = py ? a.h - 1 : 0;
        j0 = px ? a.w - 1 : 0;
    } else {
        int t = blockIdx.x, seg;
        if (t < 4 * tr) { seg = t / tr; tt = t - seg * tr; }
        else { t -= 4 * tr; seg = 4 + t / tc; tt = t - (seg - 4) * tc; }
        rowseg = seg < 4;
        if (rowseg) { py = seg >> 1; px = seg & 1; i0 = py ? a.h - 1 : 0; }           // output row 0 / ho-1, columns 2j + px
        else { px = (seg - 4) >> 1; py = seg & 1; j0 = px ? a.w - 1 : 0; }            // output column 0 / wo-1, rows 2i + py
    }
    // tile pixel p (0..31) -> low-res pixel (i, j) and whether it is a real ring pixel of this segment
    auto tile_pixel = [&](int p, int& ii, int& jj) -> bool {
        if (cornerwg) { ii = i0; jj = j0; return true; }
        if (rowseg) {
            ii = i0; jj = tt * 32 + p;
            const bool ok = jj < a.w;
            jj = min(jj, a.w - 1);
            return ok && !(px ? jj == a.w - 1 : jj == 0);  // corners belong to the corner workgroups
        }
        const int idx = tt * 32 + p;
        jj = j0; ii = min(py ? idx : idx + 1, a.h - 1);
        return idx < a.h - 1;
    };
    const int cin = a.c0 + a.c1, kchunks = cin >> 6;
    // samples: MFMA columns are pixels and a wave owns 2 samples; corner workgroups: columns are 2 x 32 samples
    if (!cornerwg && nbase + wave * 2 >= a.n) return;  // wave-uniform
    auto sample_of = [&](int r, int p) { return min(cornerwg ? nbase + r * 32 + p : nbase + wave * 2 + r, a.n - 1); };
    // slots of this wave: segments run 3 taps; corner workgroups deal their 7 taps to the waves (0,4 / 1,5 / 2,6 / 3)
    const int u0 = cornerwg ? wave : 0, ustride = cornerwg ? 4 : 1, nslots = cornerwg ? (wave < 3 ? 2 : 1) : 3;
    const int niter = nslots * kchunks;
    f32x16 acc[2][2];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[r][nt][q] = 0.0f;
    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                       (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    char* xs = ub_smem + wave * 16384;
    // The input pixels go through LDS: read directly as MFMA fragments every lane would touch its own cache line per load
    // (L1 line throughput, not bytes, was the limit: 51 us per launch); the DMA fetches 8 full 128-B pixel rows per
    // instruction, lane -> (pixel, 16-B piece), pieces swizzled by (pixel >> 1) & 7 for the conflict-free fragment reads.
    const int dsub = lane >> 3, dslot = lane & 7;
    auto load = [&](UbStage& st, int stage, int it) {
        const int u = u0 + (it % nslots) * ustride, kc = it / nslots;  // taps innermost (their pixel rows overlap: L1 / L2 reuse)
        int tap, dy = 0, dx = 0;
        if ((cornerwg || rowseg) && u < 3) { tap = 9 + u; dx = u - 1; }
        else if (cornerwg && u < 6) { tap = 12 + (u - 3); dy = u - 4; }
        else if (cornerwg) { tap = 15; }
        else { tap = 12 + u; dy = u - 1; }
        const bool first = kc * 64 < a.c0;
        const int cs = first ? a.c0 : a.c1;
        const unsigned coff = (unsigned)((first ? kc * 64 : kc * 64 - a.c0) * 2);
        char* dst = xs + stage * 8192;
#pragma unroll
        for (int d = 0; d < 8; ++d) {
            const int r = d >> 2, p = (d & 3) * 8 + dsub;
            int ii, jj;
            (void)tile_pixel(p, ii, jj);
            const int sy = min(max(ii + dy, 0), a.h - 1), sx = min(max(jj + dx, 0), a.w - 1);
            const unsigned off = (unsigned)(((sample_of(r, p) * a.h + sy) * a.w + sx) * cs * 2) + coff +
                                 (unsigned)((dslot ^ ((p >> 1) & 7)) << 4);
            if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, LDS_PTR(dst + d * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, LDS_PTR(dst + d * 1024), 16, off, 0, 0, 0);
        }
        const char* wf = (const char*)a.wpk_up_frag + ((size_t)(cb_blk * kchunks + kc) * 16 + tap) * STEP_BYTES + py * (STEP_BYTES / 2) +
                         px * 2048 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st.wa[q] = *(const uint4*)(wf + q * 4096);
            st.wb[q] = *(const uint4*)(wf + q * 4096 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    auto compute = [&](const UbStage& st, int stage, bool more) {
        // 16 vector-memory operations of the NEXT stage may still be in flight; everything older (this stage's DMA) has landed
        if (more) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const char* src = xs + stage * 8192 + l31 * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = ((q * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const uint4 xv = *(const uint4*)(src + r * 4096 + piece);
                acc[r][0] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, st.wa[q]), __builtin_bit_cast(el16x8_t, xv), acc[r][0], 0, 0, 0);
                acc[r][1] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, st.wb[q]), __builtin_bit_cast(el16x8_t, xv), acc[r][1], 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    UbStage sa, sb;
    load(sa, 0, 0);
    for (int it = 0; it < niter; it += 2) {
        if (it + 1 < niter) load(sb, 1, it + 1);
        compute(sa, 0, it + 1 < niter);
        if (it + 1 < niter) {
            if (it + 2 < niter) load(sa, 0, it + 2);
            compute(sb, 1, it + 2 < niter);
        }
    }
    int ii, jj;
    const bool valid = tile_pixel(l31, ii, jj);
    const int Y = 2 * ii + py, X = 2 * jj + px;
    const int ring = Y == 0 ? X : Y == a.ho - 1 ? a.wo + X : X == 0 ? 2 * a.wo + Y - 1 : 2 * a.wo + (a.ho - 2) + Y - 1;
    const int ring_len = 2 * a.wo + 2 * (a.ho - 2);
    if (cornerwg) {  // sum the four waves' taps (the staging buffers are dead once every wave is past its loop)
        __syncthreads();
        float* red = (float*)ub_smem;  // [3][2][2][16][64]
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int q = 0; q < 16; ++q) red[((((wave - 1) * 2 + r) * 2 + nt) * 16 + q) * 64 + lane] = acc[r][nt][q];
        }
        __syncthreads();
        if (wave > 0) return;
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int q = 0; q < 16; ++q)
#pragma unroll
                    for (int w = 0; w < 3; ++w) acc[r][nt][q] += red[(((w * 2 + r) * 2 + nt) * 16 + q) * 64 + lane];
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int smp = cornerwg ? nbase + r * 32 + l31 : nbase + wave * 2 + r;
        if (smp >= a.n || !valid) continue;
        float* op = a.up_border + ((size_t)smp * ring_len + ring) * a.cout + cb_blk * 64 + 4 * hi;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(op + nt * 32 + 8 * g) =
                    make_float4(acc[r][nt][4 * g], acc[r][nt][4 * g + 1], acc[r][nt][4 * g + 2], acc[r][nt][4 * g + 3]);
    }
#endif
}


// Few-rows form of up_border_kernel (round 5).  The kernel above gives a wave its whole K chain -- 3 taps x cin / 64 iterations of
// one memory round trip each (12 at dec4 / dec5, 24 at dec3; the corner workgroups 8 / 16), whatever the batch: 16 - 27 us per launch,
// three launches per forward, 15.9 % of a 1-row rollout and still 7.9 % at 10 rows (profiles/r04h/i_bench_nb*), all of it in front of
// the main kernel, which starts its border accumulators from these sums.  At a few rows the chip is empty, so here
//   * a segment workgroup owns a (32-pixel tile, ONE sample, 64 channels) unit, a corner workgroup 32 samples of one corner, and
//     deals the unit's (tap, chunk) iterations -- taps innermost -- to its NW = 8 waves: iteration it = wave, wave + NW, ...;
//   * a wave stages an iteration's 32 x 128-byte pixel rows in a 4 KB slot of its own 16 KB of LDS: a ring of FOUR slots, so the
//     operands of up to four iterations (every iteration of a segment wave: ceil(24 / 8) = 3 at dec3, 2 at dec4 / dec5) are in
//     flight together -- one memory round trip instead of one per iteration; the corner workgroups (7 taps: 4 - 7 iterations per
//     wave) take two;
//   * the partial accumulators are parked in the same LDS and added in WAVE ORDER (waves 0 / 1 finish one 32-channel half each): a
//     sum does not depend on timing.
// First version of this kernel (two samples per workgroup, two-slot pipeline): 12 us per launch against 16 - 27; the __launch_bounds__
// must sit on the first declaration (the template is instantiated from conv_up_halo_init, above the definition) -- without them the
// compiler assumed 1 024 threads, capped the kernel at 128 VGPRs and spilled 46 of them: slower than the kernel it replaces.
struct UbSlot {
    uint4 wa[4], wb[4];
};
// KSPLIT = false (many rows): the same 4-slot ring without the K split -- a workgroup's NW waves own NW different samples (corner
// workgroups: NW x 32) and each walks its whole chain four iterations at a time; nothing is summed across waves.
template <int NW, bool KSPLIT>
__global__ __launch_bounds__(NW * 64, (NW + 3) / 4) void up_border_split_kernel(ConvArgs a, int tr, int tc) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char ub_smem[];
    const int lane = threadIdx.x & 63, l31 = lane & 31, hi = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cb_blk = blockIdx.z;
    const int nseg = 4 * tr + 4 * tc;
    const bool cornerwg = (int)blockIdx.x >= nseg;
    // samples: K-split form -- blockIdx.y = one sample (corner workgroups every 32nd y: 32 samples); otherwise blockIdx.y = a group
    // of NW samples, one per wave (corner workgroups every 32nd y: NW x 32 samples, 32 per wave)
    if (cornerwg && (blockIdx.y & 31) != 0) return;
    const int nbase = KSPLIT ? (int)blockIdx.y : (cornerwg ? (int)blockIdx.y * NW + 32 * wave : (int)blockIdx.y * NW + wave);
    if (!KSPLIT && nbase >= a.n) return;  // wave-uniform: this wave's sample(s) do not exist
    int py, px, i0 = 0, j0 = 0, tt = 0;
    bool rowseg = true;
    if (cornerwg) {
        const int c = blockIdx.x - nseg;
        py = c >> 1; px = c & 1;
        i0 = py ? a.h - 1 : 0;
        j0 = px ? a.w - 1 : 0;
    } else {
        int t = blockIdx.x, seg;
        if (t < 4 * tr) { seg = t / tr; tt = t - seg * tr; }
        else { t -= 4 * tr; seg = 4 + t / tc; tt = t - (seg - 4) * tc; }
        rowseg = seg < 4;
        if (rowseg) { py = seg >> 1; px = seg & 1; i0 = py ? a.h - 1 : 0; }
        else { px = (seg - 4) >> 1; py = seg & 1; j0 = px ? a.w - 1 : 0; }
    }
    auto tile_pixel = [&](int p, int& ii, int& jj) -> bool {  // as in up_border_kernel
        if (cornerwg) { ii = i0; jj = j0; return true; }
        if (rowseg) {
            ii = i0; jj = tt * 32 + p;
            const bool ok = jj < a.w;
            jj = min(jj, a.w - 1);
            return ok && !(px ? jj == a.w - 1 : jj == 0);
        }
        const int idx = tt * 32 + p;
        jj = j0; ii = min(py ? idx : idx + 1, a.h - 1);
        return idx < a.h - 1;
    };
    const int cin = a.c0 + a.c1, kchunks = cin >> 6;
    auto sample_of = [&](int p) { return min(cornerwg ? nbase + p : nbase, a.n - 1); };
    const int ntaps = cornerwg ? 7 : 3;
    const int niter = ntaps * kchunks;
    f32x16 acc[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[nt][q] = 0.0f;
    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                       (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    char* xs = ub_smem + wave * 16384;
    const int dsub = lane >> 3, dslot = lane & 7;
    // per-lane source pixel of the four DMA instructions of an iteration at (dy, dx) = (0, 0): they only shift with the tap
    int sii[4], sjj[4], ssm[4];
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const int p = d * 8 + dsub;
        (void)tile_pixel(p, sii[d], sjj[d]);
        ssm[d] = sample_of(p);
    }
    auto load = [&](UbSlot& st, int slot, int it) {
        const int kc = it / ntaps, u = it - kc * ntaps;  // taps innermost (their pixel rows overlap: L1 / L2 reuse)
        int tap, dy = 0, dx = 0;
        if ((cornerwg || rowseg) && u < 3) { tap = 9 + u; dx = u - 1; }
        else if (cornerwg && u < 6) { tap = 12 + (u - 3); dy = u - 4; }
        else if (cornerwg) { tap = 15; }
        else { tap = 12 + u; dy = u - 1; }
        const bool first = kc * 64 < a.c0;
        const int cs = first ? a.c0 : a.c1;
        const unsigned coff = (unsigned)((first ? kc * 64 : kc * 64 - a.c0) * 2);
        char* dst = xs + slot * 4096;
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int p = d * 8 + dsub;
            const int sy = min(max(sii[d] + dy, 0), a.h - 1), sx = min(max(sjj[d] + dx, 0), a.w - 1);
            const unsigned off = (unsigned)(((ssm[d] * a.h + sy) * a.w + sx) * cs * 2) + coff + (unsigned)((dslot ^ ((p >> 1) & 7)) << 4);
            if (first) __builtin_amdgcn_raw_ptr_buffer_load_lds(rs0, LDS_PTR(dst + d * 1024), 16, off, 0, 0, 0);
            else __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, LDS_PTR(dst + d * 1024), 16, off, 0, 0, 0);
        }
        const char* wf = (const char*)a.wpk_up_frag + ((size_t)(cb_blk * kchunks + kc) * 16 + tap) * STEP_BYTES + py * (STEP_BYTES / 2) +
                         px * 2048 + lane * 16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            st.wa[q] = *(const uint4*)(wf + q * 4096);
            st.wb[q] = *(const uint4*)(wf + q * 4096 + 1024);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // `behind` = iterations requested AFTER this one (12 vector-memory operations each) that may still be in flight
    auto compute = [&](const UbSlot& st, int slot, int behind) {
        if (behind >= 3) asm volatile("s_waitcnt vmcnt(36)" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        const char* src = xs + slot * 4096 + l31 * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = ((q * 2 + hi) ^ ((l31 >> 1) & 7)) << 4;
            const uint4 xv = *(const uint4*)(src + piece);
            acc[0] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, st.wa[q]), __builtin_bit_cast(el16x8_t, xv), acc[0], 0, 0, 0);
            acc[1] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, st.wb[q]), __builtin_bit_cast(el16x8_t, xv), acc[1], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
    };
    // this wave's iterations: wave, wave + NW, ... (K-split form) or all of them, in batches of up to four requests
    constexpr int ISTR = KSPLIT ? NW : 1;
    const int ifirst = KSPLIT ? wave : 0;
    const int mine = ifirst < niter ? (niter - ifirst + ISTR - 1) / ISTR : 0;
    for (int j0b = 0; j0b < mine; j0b += 4) {
        const int nb4 = min(4, mine - j0b);  // wave-uniform
        UbSlot s0, s1, s2, s3;
        load(s0, 0, ifirst + (j0b + 0) * ISTR);
        if (nb4 > 1) load(s1, 1, ifirst + (j0b + 1) * ISTR);
        if (nb4 > 2) load(s2, 2, ifirst + (j0b + 2) * ISTR);
        if (nb4 > 3) load(s3, 3, ifirst + (j0b + 3) * ISTR);
        compute(s0, 0, nb4 - 1);
        if (nb4 > 1) compute(s1, 1, nb4 - 2);
        if (nb4 > 2) compute(s2, 2, nb4 - 3);
        if (nb4 > 3) compute(s3, 3, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the slots are re-filled by the next batch
    }
    if constexpr (!KSPLIT) {  // every wave owns its sample(s): store both channel halves straight from the accumulators
        int ii, jj;
        const bool valid = tile_pixel(l31, ii, jj);
        const int Y = 2 * ii + py, X = 2 * jj + px;
        const int ring = Y == 0 ? X : Y == a.ho - 1 ? a.wo + X : X == 0 ? 2 * a.wo + Y - 1 : 2 * a.wo + (a.ho - 2) + Y - 1;
        const int ring_len = 2 * a.wo + 2 * (a.ho - 2);
        const int smp = cornerwg ? nbase + l31 : nbase;
        if (smp >= a.n || !valid) return;
        float* op = a.up_border + ((size_t)smp * ring_len + ring) * a.cout + cb_blk * 64 + 4 * hi;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g)
                *(float4*)(op + nt * 32 + 8 * g) = make_float4(acc[nt][4 * g], acc[nt][4 * g + 1], acc[nt][4 * g + 2], acc[nt][4 * g + 3]);
        return;
    }
    // ---- sum over the waves in wave order: every wave parks its 32 accumulator registers in its OWN 16 KB (8 KB used)
    float* red = (float*)(ub_smem + wave * 16384);  // [nt * 16 + q][64 lanes]
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int q = 0; q < 16; ++q) red[(nt * 16 + q) * 64 + lane] = acc[nt][q];
    __syncthreads();
    if (wave >= 2) return;
    const int nt = wave;
    float sum[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) sum[q] = ((const float*)ub_smem)[(nt * 16 + q) * 64 + lane];
    for (int w = 1; w < NW; ++w) {
        const float* rw = (const float*)(ub_smem + w * 16384);
#pragma unroll
        for (int q = 0; q < 16; ++q) sum[q] += rw[(nt * 16 + q) * 64 + lane];
    }
    int ii, jj;
    const bool valid = tile_pixel(l31, ii, jj);
    const int Y = 2 * ii + py, X = 2 * jj + px;
    const int ring = Y == 0 ? X : Y == a.ho - 1 ? a.wo + X : X == 0 ? 2 * a.wo + Y - 1 : 2 * a.wo + (a.ho - 2) + Y - 1;
    const int ring_len = 2 * a.wo + 2 * (a.ho - 2);
    const int smp = cornerwg ? nbase + l31 : nbase;
    if (smp >= a.n || !valid) return;
    float* op = a.up_border + ((size_t)smp * ring_len + ring) * a.cout + cb_blk * 64 + 4 * hi + nt * 32;
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)(op + 8 * g) = make_float4(sum[4 * g], sum[4 * g + 1], sum[4 * g + 2], sum[4 * g + 3]);
#endif
}
constexpr int UB_SPLIT_NW = 8;

hipError_t launch_conv_up_halo(const ConvArgs& a, hipStream_t stream) {
    if (a.up_border == nullptr || a.wpk_up_frag == nullptr) return hipErrorInvalidValue;
    {
        const int tr = (a.w + 31) / 32, tc = (a.h - 1 + 31) / 32;
        // few rows: the K-split form (one sample per workgroup instead of eight: 8 x as many, 8 waves each -- past ~40 rows the form
        // above, whose workgroups already fill the chip, does the same sums with less LDS traffic).  DYF_UP_BORDER_SPLIT_ROWS
        // moves the switch (0 = never); the form is chosen for ConvArgs::n_sel rows when the engine pins the forms (batch_invariant)
        const char* sre = dyf_form("DYF_UP_BORDER_SPLIT_ROWS");  // read per launch (parity tests)
        const int split_rows = sre ? atoi(sre) : 40;
        // EXPERIMENT, off by default (DYF_UP_BORDER_RING4=1 enables; read per launch): the four-slot ring without the K split for many
        // rows.  Measured SLOWER than the two-slot kernel where it would apply: NS at 80 rows 8 784-8 804 against 8 843-8 851 fields/s
        // (three runs each, same box), dec3 / dec4 / dec5 at 80 rows 306.6 / 568.7 / 527.9 against 293.8 / 560.9 / 523.7 us -- with
        // 480-1 440 workgroups the chip is full either way and the ring's 64 KB of LDS per workgroup halves the resident ones.
        const char* r4e = dyf_form("DYF_UP_BORDER_RING4");
        const bool ring4 = r4e && atoi(r4e) != 0;
        if ((a.n_sel > 0 ? a.n_sel : a.n) <= split_rows) {
            dyf_form_note("up_border_split_kernel", a.n);
            hipLaunchKernelGGL((up_border_split_kernel<UB_SPLIT_NW, true>), dim3(4 * tr + 4 * tc + 4, a.n, a.cout / 64), dim3(UB_SPLIT_NW * 64),
                               UB_SPLIT_NW * 16384, stream, a, tr, tc);
        } else if (ring4) {
            // many rows, opt-in: four samples per workgroup, one per wave, the chain four iterations at a time
            dyf_form_note("up_border_ring4_kernel", a.n);
            hipLaunchKernelGGL((up_border_split_kernel<4, false>), dim3(4 * tr + 4 * tc + 4, (a.n + 3) / 4, a.cout / 64), dim3(256), 4 * 16384, stream, a, tr, tc);
        } else
            hipLaunchKernelGGL(up_border_kernel, dim3(4 * tr + 4 * tc + 4, (a.n + 7) / 8, a.cout / 64), dim3(256), UB_LDS, stream, a, tr, tc);
    }
    const bool sparse = a.up_cols != nullptr;
    // rows form (conv_halo_rows.hip: one-row pixel tiles, half the LDS fragment reads) where the plane tiles by 4 x 32;
    // DYF_HALO_ROWS=0 keeps this file's kernels.  The sparse lists are planned for one form or the other (32 / 16 slots).
    const bool rows = !(dyf_form("DYF_HALO_ROWS") && atoi(dyf_form("DYF_HALO_ROWS")) == 0);
    if (sparse ? ((a.up_mix[0] | a.up_mix[1] | a.up_mix[2]) != 0 || a.up_npad != a.up_ntiles * 16) : (rows && conv_halo_rows_up_supported(a)))
        return launch_conv_halo_rows_up(a, stream);
    const int tiles_x = sparse ? a.up_ntiles : a.w / TILE_W, tiles_per_img = tiles_x * (a.h / TILE_H);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 64;
    dyf_form_note(sparse ? "conv_up_halo_kernel<1>" : "conv_up_halo_kernel<0>", a.n);
    if (sparse)
        hipLaunchKernelGGL(conv_up_halo_kernel<1>, dim3(tiles_m * tiles_n), dim3(256), HaloCfg<1>::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n, 0);
    else {
        // DYF_HALO_TN_XCD=1: one column block per XCD (measured on dec4 at NB=80: HBM-side reads 722 -> 513 MB because the
        // weights stay in L2, but the two column blocks of a tile read their halo on different XCDs; time 649 -> 663 us,
        // whole rollout unchanged -- the extra reads of the default mapping are served by the Infinity Cache).  Off by default.
        const int xenv = dyf_form("DYF_HALO_TN_XCD") ? atoi(dyf_form("DYF_HALO_TN_XCD")) : 0;
        const bool xmode = (xenv & 1) != 0 && (tiles_n == 2 || tiles_n == 4 || tiles_n == 8);
        const int groups = xmode ? 8 / tiles_n : 1, chunk = (tiles_m + groups - 1) / groups;
        const unsigned grid = xmode ? (unsigned)(8 * chunk) : (unsigned)(tiles_m * tiles_n);
        hipLaunchKernelGGL(conv_up_halo_kernel<0>, dim3(grid), dim3(256), HaloCfg<0>::LDS_TOTAL, stream, a, tiles_x, tiles_per_img,
                           tiles_m, tiles_n, xmode ? 1 : 0);
    }
    return hipGetLastError();
}

// Column lists of the sparse form: `needed[x]` (x in [0, 2w)) marks the OUTPUT columns somebody reads.  Per horizontal
// phase px the low-res columns j with needed[2j + px] are listed and dealt evenly to ntiles list tiles of 16 slots (slots
// past a tile's share repeat its last column with bit 14 set: computed, never stored).  cbase[t] = first halo column of
// list tile t.  Returns false (dense form must be used) when a list tile does not fit the 40-column halo or the lists would
// not save at least 20 % of the work.
bool plan_up_sparse_columns(const std::vector<uint8_t>& needed, int w, std::vector<int16_t>& cols, std::vector<int16_t>& cbase,
                            std::vector<int16_t>& cidx, std::vector<int16_t>& col_map, int& ntiles, int& nvalid0, int& nvalid1,
                            int slots) {
    // slots per list tile: 16 (conv_up_halo_kernel<1>, 40-column halo) or 32 (conv_halo_rows_kernel<1>)
    const int halo_w = slots == 16 ? HaloCfg<1>::W : conv_halo_rows_sparse_halo_w();
    // compact storage order: the needed output columns in increasing order
    col_map.assign((size_t)2 * w, -1);
    int nstore = 0;
    for (int x = 0; x < 2 * w; ++x)
        if (needed[x]) col_map[x] = (int16_t)nstore++;
    std::vector<int> l[2];
    for (int j = 0; j < w; ++j)
        for (int px = 0; px < 2; ++px)
            if (needed[2 * j + px]) l[px].push_back(j);
    if (l[0].empty() || l[1].empty()) return false;
    nvalid0 = (int)l[0].size();
    nvalid1 = (int)l[1].size();
    const int nmax = std::max(nvalid0, nvalid1);
    ntiles = (nmax + slots - 1) / slots;
    if (ntiles * slots * 5 > w * 4) return false;
    const int per = (nmax + ntiles - 1) / ntiles;  // entries per tile, balanced
    cols.assign((size_t)2 * ntiles * slots, 0);
    cidx.assign((size_t)2 * ntiles * slots, 0);
    cbase.assign(ntiles, 0);
    for (int t = 0; t < ntiles; ++t) {
        int lo = w, hi = -1;
        for (int px = 0; px < 2; ++px) {
            const int n = (int)l[px].size();
            for (int i = 0; i < slots; ++i) {
                const int k = t * per + i;
                const bool real = i < per && k < n;
                const int kk = std::min(std::min(k, t * per + per - 1), n - 1);
                const int c = l[px][std::max(kk, 0)];
                cols[(size_t)px * ntiles * slots + t * slots + i] = (int16_t)(c | (real ? 0 : 0x4000));
                cidx[(size_t)px * ntiles * slots + t * slots + i] = col_map[2 * c + px];
                lo = std::min(lo, c);
                hi = std::max(hi, c);
            }
        }
        cbase[t] = (int16_t)(lo - 1);
        if (hi - lo + 3 > halo_w) return false;
    }
    return true;
}

bool plan_up_sparse_columns_mixed(const std::vector<uint8_t>& needed, int w, int h, std::vector<int16_t>& cols, std::vector<int16_t>& cbase,
                                  std::vector<int16_t>& cidx, std::vector<int16_t>& col_map, int mix[3], int& nvalid0, int& nvalid1) {
    col_map.assign((size_t)2 * w, -1);
    int nstore = 0;
    for (int x = 0; x < 2 * w; ++x)
        if (needed[x]) col_map[x] = (int16_t)nstore++;
    std::vector<int> l[2];
    for (int j = 0; j < w; ++j)
        for (int px = 0; px < 2; ++px)
            if (needed[2 * j + px]) l[px].push_back(j);
    if (l[0].empty() || l[1].empty()) return false;
    nvalid0 = (int)l[0].size();
    nvalid1 = (int)l[1].size();
    const int nmax = std::max(nvalid0, nvalid1);
    // greedy from the front: 16-entry tiles while at least 16 entries remain, then 4-entry tiles (32-entry tiles only where their
    // 72-pixel halo fits, which the interleaved phase lists of the NS geometry do not allow: a = 0 there)
    struct T { int first, slots, shape; };
    std::vector<T> tiles;
    const int slots_of[3] = {32, 16, 4};
    auto span_ok = [&](int first, int cnt, int shape) {
        int lo = w, hi = -1;
        for (int px = 0; px < 2; ++px) {
            const int n = (int)l[px].size();
            for (int i = 0; i < cnt; ++i) {
                const int c = l[px][std::min(first + i, n - 1)];
                lo = std::min(lo, c);
                hi = std::max(hi, c);
            }
        }
        return hi - lo + 3 <= conv_halo_rows_sparse_halo_w_shape(shape);
    };
    int pos = 0;
    mix[0] = mix[1] = mix[2] = 0;
    std::vector<T> ta, tb, tc;
    while (nmax - pos >= 32 && span_ok(pos, 32, 0)) { ta.push_back({pos, 32, 0}); pos += 32; }
    while (nmax - pos >= 16) {
        if (!span_ok(pos, 16, 1)) return false;
        tb.push_back({pos, 16, 1});
        pos += 16;
    }
    while (pos < nmax) {
        const int cnt = std::min(4, nmax - pos);
        if (!span_ok(pos, cnt, 2)) return false;
        tc.push_back({pos, 4, 2});
        pos += 4;
    }
    // (a 32-entry tile in front of 16-entry tiles keeps list order: a, then b, then c -- the layout the launches index)
    mix[0] = (int)ta.size(); mix[1] = (int)tb.size(); mix[2] = (int)tc.size();
    if (mix[1] > 0 && h % 8 != 0) return false;
    if (mix[2] > 0 && h % 32 != 0) return false;
    if (mix[0] + mix[1] + mix[2] == 0) return false;
    const int npad = 32 * mix[0] + 16 * mix[1] + 4 * mix[2];
    // not worth it unless it beats uniform 32-slot tiles: tile units per 32 rows
    const int units_mixed = 8 * mix[0] + 4 * mix[1] + mix[2], units_uniform = 8 * ((nmax + 31) / 32);
    if (units_mixed >= units_uniform) return false;
    cols.assign((size_t)2 * npad, 0);
    cidx.assign((size_t)2 * npad, 0);
    cbase.clear();
    int off = 0;
    for (const std::vector<T>* grp : {&ta, &tb, &tc})
        for (const T& t : *grp) {
            int lo = w;
            for (int px = 0; px < 2; ++px) {
                const int n = (int)l[px].size();
                for (int i = 0; i < t.slots; ++i) {
                    const int k = t.first + i;
                    const bool real = k < n;
                    const int c = l[px][std::min(k, n - 1)];
                    cols[(size_t)px * npad + off + i] = (int16_t)(c | (real ? 0 : 0x4000));
                    cidx[(size_t)px * npad + off + i] = col_map[2 * c + px];
                    lo = std::min(lo, c);
                }
            }
            cbase.push_back((int16_t)(lo - 1));
            off += t.slots;
            (void)slots_of;
        }
    return true;
}
