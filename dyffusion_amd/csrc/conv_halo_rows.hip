// "Rows" form of the halo convolution (gfx950): the same fused x2-bilinear-upsample + 3x3 conv (SP = 0 dense, SP = 1 sparse
// output columns) and plain 3x3 / stride 1 / pad 1 conv with 256-channel blocks (SP = 2) as conv_up_halo.hip, same weight
// fragment streams (pack_up2x_frag / pack_halo3_frag), same border ring (up_border_kernel), same epilogue -- with the pixel
// tile of an MFMA turned from 2 rows x 16 columns into ONE ROW of 32 columns, a wave owning 4 rows x 32 columns.
//
// Why: conv_up_halo_kernel runs the same instruction stream at 1 290 TFLOP/s on real activations and at 1 750 on all-zero
// operands (tools/scratch experiments): the package is power-limited, and what the kernel can still save is
// operand traffic.  Dropping half of the pixel-fragment LDS reads (timing experiment, wrong results) was worth 18-20 % on
// dec2 / dec3 and 11 % on dec4.  With 2 x 16 pixel tiles every (tile, tap) pair reads its own fragment: 36 reads per k16
// sub-step of a chunk.  With one-row tiles the fragment of halo row r at horizontal shift dx IS the operand of tile row r - dy
// for all three dy: 6 halo rows x 3 shifts = 18 reads feed the same 36 (tile, tap) products -- half the LDS traffic, and one
// v_xad_u32 of address arithmetic per 6 reads (the rows are an immediate offset apart; the swizzle key depends on the halo
// COLUMN only, which is also conflict-free for the four 16-lane groups a ds_read_b128 is serviced in).
//
// K loop per 64-channel chunk: 12 super-slots (dx, ks), each 3 mini-slots (dy) of 8 MFMAs = 2 weight fragments x 4 tile
// rows; the weight ring (6 sets, 5 mini-slots ahead) is the one of conv_up_halo_kernel, addressed (tap, ks) directly; the 6
// row fragments of super-slot s + 1 are read, two per mini-slot, during super-slot s.
#include "conv.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>
#include <vector>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int R_TH = 4, R_TW = 32, R_NWAVES = 4;
constexpr int R_STEP_BYTES = 32768;  // weights of one (tap, chunk) step: 256 columns x 64 k

// SH (sparse form only): shape of the 32 list slots of a tile row set.  0: 32 entries x 4 rows (72-pixel halo: up to 26 entries of
// the NS lists fit).  1: 16 entries x 8 rows (lanes 16..31 own the rows 4..7 of the tile; 48-pixel halo).  2: 4 entries x 32 rows
// (12-pixel halo).  Shapes 1 and 2 let a 52-entry list be tiled as 3 x 16 + 4 with every lane used (13 instead of 16 MFMA tile
// units per 32 rows); they keep the column-only swizzle key -- lanes of different row sets that read the same column collide in
// LDS (2- / 4-way), which the LDS pipe (17 % busy in these kernels) absorbs -- and store straight from the registers.
// TR (dense forms only, round 5): tile rows per wave -- 4 (the form everything above describes), 2 or 1.  A tile is one CU's matrix
// pipe for ~5.3 us per 64-channel chunk whatever the batch; at a few rows the launches are 80 - 320 such tiles on 256 CUs (one tile
// time for a third of the chip, or two for a chip and a quarter).  Tiles of 2 / 1 rows are 2 / 4 x as many workgroups of half / a
// quarter of the K-loop work each (same halo machinery: TR + 2 halo rows, TR + 2 row fragments per super-slot feeding 3 TR x 2
// MFMAs per (dx, ks)); the weight stream per MFMA grows by the same factor, which is why the large-batch form stays at 4.
template <int SP, int SH = 0, int TR = 4>
struct RowsCfg {
    static constexpr int COLS = SP == 1 ? (SH == 1 ? 16 : SH == 2 ? 4 : 32) : 32;  // list entries (columns) per tile row set
    static constexpr int RSETS = 32 / COLS;                                          // row sets of 4 rows each
    static constexpr int ROWS = TR * RSETS;                                          // tile rows of a workgroup
    static_assert(TR == 4 || (SP != 1 && (TR == 2 || TR == 1)), "short tiles exist for the dense forms");
    static constexpr int W = SP == 1 ? (SH == 1 ? 48 : SH == 2 ? 12 : 72) : 34;  // halo width in pixels
    static constexpr int REAL = (ROWS + 2) * W;       // 204 / 432 / 480 / 408
    static constexpr int PIX = (REAL + 7) / 8 * 8;    // padded to whole DMA instructions (8 pixels each)
    static constexpr int BYTES = PIX * 128;           // 26 624 / 55 296
    static constexpr int NBUF = SP == 1 ? 1 : 2;
    static_assert(SP == 1 || SH == 0, "shapes are a property of the sparse form");
    static constexpr bool PLAIN = SP == 2;
    static constexpr int INSTR = PIX / 8;             // 26 / 54
    static constexpr int PER_WAVE = (INSTR + R_NWAVES - 1) / R_NWAVES;
    static constexpr int HOFF_OFF = NBUF * BYTES;     // per-thread halo source offsets [PER_WAVE][256]
    static constexpr int TAB_END = HOFF_OFF + PER_WAVE * 1024;    // 60 416 / 69 632 B
    // output staging (dense forms, -DHALO_NO_STAGE disables): per wave one tile row of [32 pixels][64 channels] 16-bit + 16 B pad
    // per pixel; the sparse form has no room for it next to its 55 KB halo (two workgroups per CU)
#ifndef HALO_NO_STAGE
    static constexpr bool STAGE = SP != 1;
#else
    static constexpr bool STAGE = false;
#endif
    // sparse form: room for HALF a tile row per wave (16 pixels, two passes per row) + the row's compact column indices
#ifndef HALO_NO_STAGE
    static constexpr bool STAGE_HALF = SP == 1 && SH == 0;
#else
    static constexpr bool STAGE_HALF = false;
#endif
    static constexpr int OROW = 144;
    static constexpr int LDS_TOTAL = TAB_END + (STAGE ? R_NWAVES * 32 * OROW : 0) +
                                     (STAGE_HALF ? R_NWAVES * (16 * OROW + 128) : 0);  // 78 848 / 79 360 B: two workgroups per CU
    static constexpr int ROW_BYTES = W * 128;
};

// weight offset of mini-slot g = (dx * 4 + ks) * 3 + dy inside a chunk's 16-tap block: tap (dy, dx) step, k16 sub-step ks
constexpr unsigned rows_woff(int g) { return (unsigned)(((g % 3) * 3 + g / 12) * R_STEP_BYTES + ((g / 3) % 4) * 4096); }

}  // namespace

// KS: split-K instance (conv_halo_rows_splitk_kernel, dense forms): the workgroup contracts the chunks [ks_idx * cpt / ksplit,
// (ks_idx + 1) * cpt / ksplit) only and leaves its raw fp32 accumulators in ConvArgs::splitk_ws[ks_idx][pixel][channel]; the border
// ring goes into split 0; conv_splitk_finish4_kernel adds the splits in index order and runs the epilogue.
template <int SP, int SH, bool KS = false, int TR = 4>
__device__ __forceinline__ void conv_halo_rows_body(const ConvArgs& a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n, int bid,
                                                    int ks_idx = 0, int ksplit = 1) {
#if defined(__HIP_DEVICE_COMPILE__)
    using H = RowsCfg<SP, SH, TR>;
    constexpr int HALO_W = H::W, HALO_REAL = H::REAL, HALO_BYTES = H::BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int wpy = wave >> 1, wpx = wave & 1;  // output phase of this wave (upsample forms)

    // XCD-aware tile id; the column blocks of one tile are consecutive (they share the halo in L2)
    const int total = tiles_m * tiles_n;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int n_img = tm / tiles_per_img;
    const int t_in = tm - n_img * tiles_per_img;
    const int ty0 = (t_in / tiles_x) * H::ROWS;
    const int lx = t_in % tiles_x;  // column tile: 32 contiguous columns, or COLS entries of the column lists
    // sparse shapes: lane -> (list entry ci of the tile, row set rs); the lane's four tile rows are ty0 + 4 rs + {0..3}
    const int ci = l31 & (H::COLS - 1), rs = l31 / H::COLS;
    int col, cbase, cstore = 0;
    bool lane_valid = true;
    if (SP == 1) {
        cbase = a.up_cbase[a.up_sh_cb + lx];
        const int li = wpx * a.up_npad + a.up_sh_off + lx * H::COLS + ci;
        const int entry = a.up_cols[li];  // bit 14: padding entry (computed, not stored)
        col = entry & 0x3FFF;
        lane_valid = (entry & 0x4000) == 0;
        cstore = a.up_cidx[li];
    } else {
        cbase = lx * R_TW - 1;
        col = lx * R_TW + l31;
    }

    const int cin = a.c0 + a.c1;
    const int cpt = cin >> 6;
    const int cbeg = KS ? ks_idx * cpt / ksplit : 0, cend = KS ? (ks_idx + 1) * cpt / ksplit : cpt;  // this workgroup's chunks
    const int gh = a.h, gw = a.w;
    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_up_frag, 0,
                                                          (int)(unsigned)((size_t)(H::PLAIN ? 1 : 4) * a.cout * 16 * cin * 2), 0x00020000);

    // LDS swizzle key of halo column hx (XORed into the 16-B chunk index of the pixel's 128-B row).  A ds_read_b128 of 32
    // consecutive pixels of one halo row is serviced in four 16-lane groups ({0-3,12-15,20-27}, ...): with this key every group
    // touches each of the 64 banks once, for every horizontal shift.
#define HKEY(hx) (((hx) >> 1) & 7)
    // ---- halo DMA: instruction i (i % 4 == wave) fills halo pixels [8i, 8i+8); lane -> (pixel, 16-B chunk); the per-lane
    // source offsets are parked in LDS (the K loop needs every VGPR)
    const int sub = lane >> 3;
    unsigned* h_tab = (unsigned*)(smem + H::HOFF_OFF) + tid;
#pragma unroll
    for (int j = 0; j < H::PER_WAVE; ++j) {
        const int i = j * R_NWAVES + wave;
        int hp = i * 8 + sub;
        if (hp > HALO_REAL - 1) hp = HALO_REAL - 1;  // padding slots re-read the last halo pixel
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int yy = ty0 - 1 + hy, xx = cbase + hx;
        const int y = min(max(yy, 0), gh - 1), x = min(max(xx, 0), gw - 1);  // replicate clamp (upsample forms)
        const int gch = (lane & 7) ^ HKEY(hx);
        unsigned off = (unsigned)((n_img * a.h + y) * a.w + x) * (unsigned)(a.c0 * 2) + gch * 16;  // c0 == c1 (checked on host)
        if (H::PLAIN && (yy != y || xx != x)) off = 0xFFFFFFFFu;  // plain convs: zero padding = out-of-range DMA offset
        h_tab[j * 256] = off;
    }
    auto issue_halo = [&](int chunk) {
        const int cb = chunk << 6;
        const bool second = cb >= a.c0;
        const unsigned coff = (unsigned)((second ? cb - a.c0 : cb) * 2);
        char* dst = smem + (H::NBUF == 2 ? (chunk & 1) * HALO_BYTES : 0);
#pragma unroll
        for (int j = 0; j < H::PER_WAVE; ++j) {
            const int i = j * R_NWAVES + wave;
            if (i < H::INSTR) {
                unsigned vo = h_tab[j * 256];
                if (!H::PLAIN || vo != 0xFFFFFFFFu) vo += coff;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(dst + i * 1024), 16, vo, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(dst + i * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    f32x16 acc[2][4];  // [32-channel half][tile row]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][t][r] = 0.0f;
    if (!H::PLAIN && (!KS || ks_idx == 0)) {
        // border pixels of the OUTPUT start from the correction sums of up_border_kernel (ring index: top row, bottom row,
        // left column, right column); lane (l31, hi) holds channels nt*32 + 8*g + 4*hi + {0..3} of its pixel
        const bool edge_col = col == 0 || col == gw - 1;
        if (ty0 == 0 || ty0 + H::ROWS == gh || __builtin_amdgcn_ballot_w64(edge_col) != 0ull) {
            const int ring_len = 2 * a.wo + 2 * (a.ho - 2);
            const int X = 2 * col + wpx;
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                const int Y = 2 * (ty0 + TR * rs + t) + wpy;
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
                            acc[nt][t][4 * g + 0] = v.x; acc[nt][t][4 * g + 1] = v.y;
                            acc[nt][t][4 * g + 2] = v.z; acc[nt][t][4 * g + 3] = v.w;
                        }
                }
            }
        }
    }

    int cl = col - cbase;  // halo column of this lane's pixel at dx = 0 (>= 1); halo row of tile row t at dy = 0 is 4 rs + t + 1
    const unsigned rs_off = (unsigned)(rs * TR * H::ROW_BYTES);  // the lane's row set inside the halo (0 unless a sparse shape)
    const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
    const unsigned w_voff = (unsigned)lane * 16u;

    u32x4 bq[6][2];     // weight fragments: ring of 6 sets, 5 mini-slots ahead
    el16x8_t pq[2][6];  // pixel fragments: halo rows 0..5 of the current / next super-slot
    unsigned ab = 0, ax = 0, pa = 0;

#ifdef HALO_EXP_W_ALIAS
#define W_ALIAS(x) ((x) & 0x3FFFu)
#else
#define W_ALIAS(x) (x)
#endif
#define ISSUE_B(SET, SOFF)                                                                                   \
    _Pragma("unroll") for (int nt = 0; nt < 2; ++nt)                                                         \
        bq[SET][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + nt * 1024, W_ALIAS(SOFF), 0);
#define DSRO(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
#define LGKM_WAIT0                                                        \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
    __builtin_amdgcn_sched_barrier(0);
#define PIN __builtin_amdgcn_sched_barrier(0);
    // LDS address of the row-0 fragment of super-slot SN = dx * 4 + ks: pixel column cl + dx - 1, k16 sub-step ks
#define PADDR(SN)                                                                                            \
    {                                                                                                        \
        if (((SN) & 3) == 0) {                                                                               \
            int c = cl + ((SN) >> 2) - 1;                                                                    \
            asm volatile("" : "+v"(c)); /* opaque: keeps the addresses of all shifts from being hoisted */   \
            ab = Hs + rs_off + (unsigned)c * 128u;                                                           \
            ax = (unsigned)((hi ^ HKEY(c)) << 4);                                                            \
        }                                                                                                    \
        pa = (ax ^ (unsigned)(((SN) & 3) << 5)) + ab;                                                        \
    }
#define RD(SET, R) if constexpr ((R) < TR + 2) { DSRO(pq[SET][R], pa, (R) * H::ROW_BYTES) }
#define MF(NT, T, ASET, BSET, DY)                                                                            \
    if constexpr ((T) < TR) {                                                                                \
        acc[NT][T] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, bq[BSET][NT]), pq[ASET][(T) + (DY)], acc[NT][T], 0, 0, 0); \
    }
    // mini-slot G of the chunk (dy = G % 3): 8 MFMAs on ring set G % 6; requests the fragments of mini-slot G + 5
#define WSRC(G) ((G) < 36 ? soff_c + rows_woff((G) % 36) : soff_n + rows_woff((G) % 36))
#define MS(G, ASET, LSET, R0, R1, LOAD, WAITL)                                                               \
    {                                                                                                        \
        if (WAITL) { LGKM_WAIT0 }                                                                            \
        ISSUE_B(((G) + 5) % 6, WSRC((G) + 5))                                                                \
        MF(0, 0, ASET, (G) % 6, (G) % 3) PIN                                                                 \
        if (LOAD) { RD(LSET, R0) }                                                                           \
        MF(0, 1, ASET, (G) % 6, (G) % 3) PIN                                                                 \
        if (LOAD) { RD(LSET, R1) }                                                                           \
        MF(0, 2, ASET, (G) % 6, (G) % 3) PIN                                                                 \
        MF(0, 3, ASET, (G) % 6, (G) % 3) PIN                                                                 \
        MF(1, 0, ASET, (G) % 6, (G) % 3) MF(1, 1, ASET, (G) % 6, (G) % 3)                                    \
        MF(1, 2, ASET, (G) % 6, (G) % 3) MF(1, 3, ASET, (G) % 6, (G) % 3) PIN                                \
    }
    // super-slot S: uses pq[S & 1], fetches the row fragments of super-slot S + 1 into pq[(S + 1) & 1]
#define SS(S, LOAD)                                                                                          \
    {                                                                                                        \
        LGKM_WAIT0                                                                                           \
        if (LOAD) PADDR((S) + 1)                                                                             \
        MS(3 * (S) + 0, (S) & 1, ((S) + 1) & 1, 0, 1, LOAD, false)                                           \
        MS(3 * (S) + 1, (S) & 1, ((S) + 1) & 1, 2, 3, LOAD, false)                                           \
        MS(3 * (S) + 2, (S) & 1, ((S) + 1) & 1, 4, 5, LOAD, false)                                           \
    }

    issue_halo(cbeg);
    // timing experiments (wrong results): -DHALO_EXP_W_SHARE makes the four waves stream the SAME fragments (do simultaneous
    // requests meet in L1?), -DHALO_EXP_W_ALIAS serves the stream from 16 KB
#ifdef HALO_EXP_W_SHARE
    const unsigned soff_w = 0u;
#else
    const unsigned soff_w = (unsigned)(wpy * (R_STEP_BYTES / 2) + wpx * 2048);
#endif
    unsigned soff_c = (unsigned)((tn * cpt + cbeg) * 16) * (unsigned)R_STEP_BYTES + soff_w, soff_n = soff_c;
    // (pinned in program order: the compiler's own vmcnt at the loop head is merged over the entry and the back edge)
    PIN
    ISSUE_B(0, soff_c + rows_woff(0)) PIN
    ISSUE_B(1, soff_c + rows_woff(1)) PIN
    ISSUE_B(2, soff_c + rows_woff(2)) PIN
    ISSUE_B(3, soff_c + rows_woff(3)) PIN
    ISSUE_B(4, soff_c + rows_woff(4)) PIN
    for (int chunk = cbeg; chunk < cend; ++chunk) {
        if (H::NBUF == 2) {
            // halo of this chunk landed (everything older than the 10 weight loads in flight), every wave is done with the
            // other buffer -> prefetch the next chunk's halo into it
            asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            if (chunk + 1 < cend) issue_halo(chunk + 1);
        } else {
#ifdef HALO_EXP_NO_DMA_WAIT  // timing experiment (wrong results): the single-buffered halo is not waited for
            asm volatile("s_waitcnt vmcnt(24)" ::: "memory");
#else
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
        }
        const unsigned Hs = lds_base + (H::NBUF == 2 ? (chunk & 1) * HALO_BYTES : 0);
        asm volatile("" : "+v"(cl));  // keep the LDS addresses from being hoisted out of the chunk loop
        soff_n = chunk + 1 < cend ? soff_c + 16u * (unsigned)R_STEP_BYTES : soff_c;  // tail: harmless re-fetch
        PADDR(0)
        RD(0, 0) RD(0, 1) RD(0, 2) RD(0, 3) RD(0, 4) RD(0, 5)
        SS(0, true) SS(1, true) SS(2, true) SS(3, true)
        SS(4, true) SS(5, true) SS(6, true) SS(7, true)
        SS(8, true) SS(9, true) SS(10, true) SS(11, false)
        soff_c = soff_n;
        if (H::NBUF == 1 && chunk + 1 < cend) {  // single buffer: every wave is done reading -> request the next chunk's halo
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            issue_halo(chunk + 1);
        }
    }
#undef SS
#undef MS
#undef WSRC
#undef MF
#undef RD
#undef PADDR
#undef PIN
#undef LGKM_WAIT0
#undef DSRO
#undef ISSUE_B

    // ---- epilogue straight from the accumulators (conv_up_halo_kernel's, with tile rows instead of row pairs).  Lane (l31, hi)
    // of tile (nt, t) holds pixel (row t, column l31) and channels nt*32 + 8*g + 4*hi + {0..3}.
    const RngKey key = drop_row_key(a.drop, n_img);
    const uint32_t row0 = (uint32_t)n_img * (uint32_t)(a.ho * a.wo * a.cout);
    const int ch_blk = H::PLAIN ? tn * 256 + wave * 64 : tn * 64;
    const uint32_t ci_base = (uint32_t)((a.coef_div > 1 ? n_img / a.coef_div : n_img) * a.coef_stride + ch_blk + 4 * hi);
    const int ry0 = ty0 + TR * rs;  // first of the lane's tile rows
    const uint32_t m0 = H::PLAIN ? (uint32_t)((n_img * a.ho + ty0) * a.wo + col)
                                : (uint32_t)((n_img * a.ho + 2 * ry0 + wpy) * a.wo + 2 * col + wpx);
    const uint32_t o0 = m0 * (uint32_t)a.cout + (uint32_t)ch_blk;
    const uint32_t t_stride = (uint32_t)((H::PLAIN ? 1 : 2) * a.wo * a.cout);
    const uint32_t store0 = SP == 1 ? (uint32_t)((n_img * a.ho + 2 * ry0 + wpy) * a.up_wo_store + cstore) * (uint32_t)a.cout +
                                          (uint32_t)(tn * 64)
                                    : o0;
    const uint32_t st_stride = SP == 1 ? (uint32_t)(2 * a.up_wo_store * a.cout) : t_stride;
    if constexpr (KS) {
        static_assert(SP != 1, "split-K serves the dense forms");
        // raw fp32 partial sums, [split][output element] in the layout of the (dense NHWC) output tensor
        float* ws = a.splitk_ws + (size_t)ks_idx * ((size_t)a.n * a.ho * a.wo * a.cout);
#pragma unroll
        for (int t = 0; t < TR; ++t)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(ws + (size_t)(o0 + t * t_stride + nt * 32 + 8 * g + 4 * hi)) =
                        make_float4(acc[nt][t][4 * g], acc[nt][t][4 * g + 1], acc[nt][t][4 * g + 2], acc[nt][t][4 * g + 3]);
        return;
    }
    auto epilogue = [&](auto act_c, auto mode_c) {
        constexpr int ACT = decltype(act_c)::value, MODE = decltype(mode_c)::value;
        const float ps = drop_prescale<ACT, MODE>(a.drop);  // dropout scale folded into the affine
        if constexpr (H::STAGE) {
            // Dense forms: a tile row of this wave is 32 pixels x 128 bytes (its 64 channels), one whole 128-byte line per pixel.
            // Stored from the registers, an instruction writes a 32-byte piece of 32 different lines; through a per-wave LDS
            // staging row it writes 8 whole lines (lane = (pixel, 16-byte chunk)): a quarter of the line requests, no partial
            // lines.  (conv_enc0_stem.hip: the same change was worth 20 % of a store-bound kernel.)
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
            const uint32_t pstride = (uint32_t)((H::PLAIN ? 1 : 2) * a.cout);     // elements between the row's consecutive pixels
            const uint32_t row_base = store0 - (uint32_t)l31 * pstride;             // pixel 0 of this wave's tile row 0
            const int rpx = lane >> 3, rch = lane & 7;                              // read-back role: (pixel 8 k + rpx, 16-byte chunk)
#pragma unroll
            for (int t = 0; t < TR; ++t) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int cg0 = nt * 32 + 16 * g2;
                        const uint32_t e0 = o0 + t * t_stride + cg0 + 4 * hi;
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaf(acc[nt][t][8 * g2 + q], ca[nt][g2][q], cc[nt][g2][q]);
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
#ifdef HALO_EXP_NO_STORE  // timing experiment (wrong results): the staged epilogue without its global stores
                    if (o.x == 0x12345678u && o.y == 0x9abcdef0u)
#endif
#ifndef HALO_NO_NT_STORE  // non-temporal stores (the output is not read again by this kernel, and the weights / halos keep their
                         // place in L2): dec4 558 -> 545 us; -DHALO_NO_NT_STORE restores plain stores
                    __builtin_nontemporal_store(__builtin_bit_cast(u32x4, o), (u32x4*)(a.out_el16 + (size_t)(row_base + t * st_stride + (uint32_t)px * pstride + rch * 8)));
#else
                    *(uint4*)(a.out_el16 + (size_t)(row_base + t * st_stride + (uint32_t)px * pstride + rch * 8)) = o;
#endif
                }
            }
            return;
        }
        if constexpr (H::STAGE_HALF) {
            // Sparse form: the same whole-line stores with half a tile row (16 list entries) staged at a time; the compact column
            // of every entry (or -1 for a padding entry: computed, not stored) travels through LDS to the lane that stores it.
            unsigned char* ost = (unsigned char*)smem + H::TAB_END + wave * (16 * H::OROW + 128);
            int* cst = (int*)(ost + 16 * H::OROW);
            if (hi == 0) cst[l31] = lane_valid ? cstore : -1;
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
            // compact output row of this wave's tile row 0, column 0, channel block tn
            const uint32_t row_base = (uint32_t)((n_img * a.ho + 2 * ty0 + wpy) * a.up_wo_store) * (uint32_t)a.cout + (uint32_t)(tn * 64);
            const int rpx = lane >> 3, rch = lane & 7;
#pragma unroll
            for (int t = 0; t < TR; ++t) {
                uint4 o[2][2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int cg0 = nt * 32 + 16 * g2;
                        const uint32_t e0 = o0 + t * t_stride + cg0 + 4 * hi;  // dropout stream: dense position
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaf(acc[nt][t][8 * g2 + q], ca[nt][g2][q], cc[nt][g2][q]);
                        act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                        act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                        uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                        uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        o[nt][g2].x = s0[0]; o[nt][g2].y = s1[0]; o[nt][g2].z = s0[1]; o[nt][g2].w = s1[1];
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if ((l31 >> 4) == h) {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int g2 = 0; g2 < 2; ++g2)
                                *(uint4*)(ost + (l31 & 15) * H::OROW + (nt * 32 + 16 * g2 + 8 * hi) * 2) = o[nt][g2];
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k) {
                        const int px16 = 8 * k + rpx;
                        const int c = cst[16 * h + px16];
                        const uint4 val = *(const uint4*)(ost + px16 * H::OROW + rch * 16);
                        if (c >= 0) *(uint4*)(a.out_el16 + (size_t)(row_base + t * st_stride + (uint32_t)c * (uint32_t)a.cout + rch * 8)) = val;
                    }
                }
            }
            return;
        }
        if constexpr (SP == 1 && SH != 0) {
            // Sparse lane shapes (16 entries x 2 row sets, 4 entries x 8 row sets): whole-line stores through a per-wave staging row of
            // all 32 list slots, laid OVER the halo buffer -- the K loop is over, a barrier makes sure every wave is done reading it
            // (their halo leaves no LDS for a staging area of its own; stored as 32-byte pieces straight from the registers these
            // tiles wrote 1.24x the output and the stores cost 9 % of the launch: 537 vs 489 us with the stores removed).
#ifndef HALO_NO_STAGE_SHAPES
            __syncthreads();
            unsigned char* ost = (unsigned char*)smem + wave * (32 * H::OROW + 128);
            int* cst = (int*)(ost + 32 * H::OROW);
            if (hi == 0) cst[l31] = lane_valid ? cstore : -1;
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
            const int rpx = lane >> 3, rch = lane & 7;  // read-back role: (slot 8 k + rpx, 16-byte chunk)
#pragma unroll
            for (int t = 0; t < TR; ++t) {
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int g2 = 0; g2 < 2; ++g2) {
                        const int cg0 = nt * 32 + 16 * g2;
                        const uint32_t e0 = o0 + t * t_stride + cg0 + 4 * hi;  // dropout stream: dense position
                        float v[8];
#pragma unroll
                        for (int q = 0; q < 8; ++q) v[q] = fmaf(acc[nt][t][8 * g2 + q], ca[nt][g2][q], cc[nt][g2][q]);
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
                    const int c = cst[px];
                    const int orow = 2 * (ty0 + R_TH * (px / H::COLS) + t) + wpy;  // output row of the slot's row set
                    const uint4 val = *(const uint4*)(ost + px * H::OROW + rch * 16);
                    if (c >= 0) {
#ifndef HALO_NO_NT_STORE
                        __builtin_nontemporal_store(__builtin_bit_cast(u32x4, val),
                                                    (u32x4*)(a.out_el16 + ((size_t)((n_img * a.ho + orow) * a.up_wo_store + c) * a.cout + tn * 64 + rch * 8)));
#else
                        *(uint4*)(a.out_el16 + ((size_t)((n_img * a.ho + orow) * a.up_wo_store + c) * a.cout + tn * 64 + rch * 8)) = val;
#endif
                    }
                }
            }
            return;
#endif
        }
        // 32-channel half outermost, tile rows, then the two 16-channel groups of the half: the two 32-byte pieces of a pixel's
        // 64-byte half block are stored back to back and leave the L2 as whole 64-byte writes (with the channel groups
        // outermost PMC counted 1.65x the algorithmic write bytes)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            float ca[2][8], cc[2][8];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int cg0 = nt * 32 + 16 * g2;
                const float4 ca0 = *(const float4*)(a.coef_a + ci_base + cg0), ca1 = *(const float4*)(a.coef_a + ci_base + cg0 + 8);
                const float4 cc0 = *(const float4*)(a.coef_c + ci_base + cg0), cc1 = *(const float4*)(a.coef_c + ci_base + cg0 + 8);
                ca[g2][0] = ca0.x * ps; ca[g2][1] = ca0.y * ps; ca[g2][2] = ca0.z * ps; ca[g2][3] = ca0.w * ps;
                ca[g2][4] = ca1.x * ps; ca[g2][5] = ca1.y * ps; ca[g2][6] = ca1.z * ps; ca[g2][7] = ca1.w * ps;
                cc[g2][0] = cc0.x * ps; cc[g2][1] = cc0.y * ps; cc[g2][2] = cc0.z * ps; cc[g2][3] = cc0.w * ps;
                cc[g2][4] = cc1.x * ps; cc[g2][5] = cc1.y * ps; cc[g2][6] = cc1.z * ps; cc[g2][7] = cc1.w * ps;
            }
#pragma unroll
            for (int t = 0; t < TR; ++t) {
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int cg0 = nt * 32 + 16 * g2;
                    const uint32_t obase = o0 + t * t_stride + cg0;
                    const uint32_t e0 = obase + 4 * hi;
                    float v[8];
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = fmaf(acc[nt][t][8 * g2 + q], ca[g2][q], cc[g2][q]);
                    act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                    act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                    uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                    uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                    const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                    const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                    uint4 o;
                    o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                    const uint32_t sbase = store0 + t * st_stride + cg0;
#ifdef HALO_EXP_NO_STORE  // timing experiment (wrong results): the register-store epilogue without its stores
                    if (o.x == 0x12345678u && o.y == 0x9abcdef0u) *(uint4*)(a.out_el16 + (size_t)(sbase + 8 * hi)) = o;
#else
                    if (SP != 1 || lane_valid) *(uint4*)(a.out_el16 + (size_t)(sbase + 8 * hi)) = o;
#endif
                }
            }
        }
    };
    auto by_mode = [&](auto act_c) {
        if (a.drop.mode == 0) epilogue(act_c, std::integral_constant<int, 0>{});
        else if (a.drop.mode == 1) epilogue(act_c, std::integral_constant<int, 1>{});
        else epilogue(act_c, std::integral_constant<int, 2>{});
    };
    if (a.act == ACT_RELU) by_mode(std::integral_constant<int, ACT_RELU>{});
    else if (a.act == ACT_LEAKY) by_mode(std::integral_constant<int, ACT_LEAKY>{});
    else if (a.act == ACT_SILU) by_mode(std::integral_constant<int, ACT_SILU>{});
    else by_mode(std::integral_constant<int, ACT_NONE>{});
#undef HKEY
#endif
}

template <int SP, int SH = 0>
__global__ __launch_bounds__(256, 2) void conv_halo_rows_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n) {
    conv_halo_rows_body<SP, SH>(a, tiles_x, tiles_per_img, tiles_m, tiles_n, (int)blockIdx.x);
}

// Split-K instance for FEW ROWS (round 5).  A tile of these kernels is 24 us (dec4, 4 chunks) to 85 us (dec2, 16 chunks) of one
// workgroup's serial instruction stream, whatever the batch: at one or two rows a launch is 16 - 128 workgroups on 256 CUs and takes
// exactly that long (profiles/r04i_bench_nb1: dec3 + dec4 21 % of the rollout), at 10 - 20 rows dec2 / dec3 are 80 - 320 workgroups
// -- one tile time for a quarter of the chip, or two for a chip and a quarter.  Here blockIdx.y deals a tile's 64-channel chunks to
// `splitk` workgroups (raw fp32 partials in ConvArgs::splitk_ws, the split-K scratch conv_igemm_kernel uses) and
// conv_splitk_finish4_kernel adds them in split order and runs the epilogue: the K chain of a workgroup and the idle part of the
// chip shrink by the factor.  The summation order of an output element depends on the factor only, which the launcher derives from
// the tile count of ConvArgs::n_sel rows when the engine pins the forms.
// short tiles (RowsCfg TR = 2 / 1): dense upsample form (SP = 0) and plain 3x3 form (SP = 2)
template <int SP, int TR>
__global__ __launch_bounds__(256, 2) void conv_halo_rows_tr_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n) {
    conv_halo_rows_body<SP, 0, false, TR>(a, tiles_x, tiles_per_img, tiles_m, tiles_n, (int)blockIdx.x);
}

// Tile rows per wave of a dense launch of `tiles4` four-row tiles (counted for ConvArgs::n_sel rows when the engine pins the forms).
// A CU works through its workgroups at the rate of its matrix pipe whether one or two are resident, so what short tiles buy is
// GRANULARITY: launches that leave CUs idle (<= 128 tiles: halves / quarters spread over twice / four times the CUs) or end in a
// ragged round (256 < tiles <= 384: 2.5 half-tiles per CU instead of 2 whole ones on a quarter of the chip).  DYF_ROWS_TR = 4 / 2 /
// 1 forces a shape (read per launch); measured in DESIGN.md 5 (round 5).
static int rows_tile_rows(long long tiles4, int h) {
    if (const char* f = dyf_form("DYF_ROWS_TR")) {
        const int t = atoi(f);
        if ((t == 1 || t == 2 || t == 4) && h % t == 0) return t;
    }
    // measured (NS decoder, us per launch with 4- / 2- / 1-row tiles; dec3 = 16 tiles per row, dec4 = 64, dec2 = 8): dec3 at 4 rows
    // 43.3 / 36.8 / 36.8, at 7 rows 51.2 / 39.7 / 55.7, at 10 rows 52.8 / 60.3 / 76.9, at 20 rows 101.7 / 93.8 / 128.5; dec4 at 1 row
    // 30.5 / 22.0 / 21.8, at 2 rows 31.4 / 24.7 / 32.5, at 4 rows 36.6 / 40.1 / 54.4; dec2 at 10 rows 53.3 (split-K) / 48.5 / 82.9, at 14
    // rows 61.8 / 54.1 / 86.8, at 40 rows 166.6 / 157.9 / 230.8 -- one-row tiles never win; at <= 32 tiles the four-row tiles with
    // split-K (rows_splitk_factor) are as good or better (dec3 at one row 31.4 against 34.8)
    if (tiles4 > 32 && tiles4 <= 128) return 2;
    if (tiles4 > 256 && tiles4 <= 384) return 2;
    return 4;
}

template <int SP>
__global__ __launch_bounds__(256, 2) void conv_halo_rows_splitk_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n) {
    conv_halo_rows_body<SP, 0, true>(a, tiles_x, tiles_per_img, tiles_m, tiles_n, (int)blockIdx.x, (int)blockIdx.y, a.splitk);
}

// Split factor of a dense rows launch, 1 = no split.  What the split buys is the K chain of a tile (one workgroup keeps one CU's matrix
// pipe busy for ~5.3 us per 64-channel chunk: dec2 85 us, dec3 42 us, dec4 / dec5 21 us per tile) divided by the factor -- as long as
// the workgroups still fit the chip side by side; what it costs is writing the fp32 partials and reading them back (2 s + 1 passes
// over the output at ~4 TB/s, the partial stores are 16-byte pieces) and one more launch.  Measured on the NS decoder (round 5,
// tools/bench_small_rows.py): dec3 at one row 42 -> 21 us with 8 splits, dec4 at two rows 22 -> 42 us with 4 (its 128 x 128 output is
// 17 MB per partial set) -- so the factor comes from this cost model, not from a fill target: a power of two that divides the chunk
// count, at most 8, that minimises the modelled time.  m_cout = output elements of the launch.  DYF_HALO_SPLITK=0 disables,
// DYF_HALO_SPLITK_FORCE=s forces a factor (tests); both read per launch.
static int rows_splitk_factor(const ConvArgs& a, long long tiles_sel, long long tiles, int cpt, long long m_cout) {
    if (a.splitk_ws == nullptr || a.out_f32 != nullptr || a.out_el16 == nullptr || (a.cout & 3) != 0) return 1;
    const char* on = dyf_form("DYF_HALO_SPLITK");
    if (on && atoi(on) == 0) return 1;
    int best = 1;
    if (const char* f = dyf_form("DYF_HALO_SPLITK_FORCE")) {
        best = atoi(f);
        if (best < 1 || best > 8 || (best & (best - 1)) != 0 || cpt % best != 0) best = 1;
    } else {
        // the model is evaluated for the rows the form is pinned to (n_sel) so that a batch_invariant engine keeps one factor
        const double bytes = (double)m_cout * ((double)tiles_sel / (double)std::max<long long>(tiles, 1)) * 4.0;
        double best_us = 1e30;
        for (int s = 1; s <= 8 && cpt % s == 0; s *= 2) {
            const double rounds = std::ceil((double)tiles_sel * s / 256.0);
            double us = rounds * (5.3 * cpt / s + 4.0);
            if (s > 1) us += 3.0 + (2.0 * s + 1.0) * bytes / 4.0e6;
            if (us < best_us * (s > 1 ? 0.9 : 1.0)) { best_us = us; best = s; }  // a split must win by 10 %
        }
    }
    while (best > 1 && (long long)best * m_cout > a.splitk_cap) best >>= 1;
    return best;
}

// EXPERIMENT (DYF_ROWS_PERSISTENT=512; off by default: measured SLOWER, dec4 541 -> 555 us, dec3 283 -> 287 us -- the hardware already
// starts a new one-tile workgroup the moment one retires, so nothing is gained, and the tile boundary adds a barrier).
// Persistent form: 512 resident workgroups walk the tiles (stride = grid, a multiple of 8: a workgroup stays on its XCD).  A wave
// that ends waits for its stores to be acknowledged (s_endpgm implies s_waitcnt 0) and holds its registers and LDS meanwhile: in
// the one-tile form that store phase is 12-17 % of dec3 / dec4 (565 vs 470 us with the stores removed).  Here the next tile's halo
// DMA and weight stream are issued right behind the stores; the boundary between tiles is a bare s_barrier behind an LDS wait (not
// __syncthreads, whose fence would wait for the stores).
template <int SP, int SH = 0>
__global__ __launch_bounds__(256, 2) void conv_halo_rows_persistent_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n) {
    const int total = tiles_m * tiles_n;
    for (int bid = (int)blockIdx.x; bid < total; bid += (int)gridDim.x) {
        conv_halo_rows_body<SP, SH>(a, tiles_x, tiles_per_img, tiles_m, tiles_n, bid);
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
#endif
    }
}

// The 16-entry and the 4-entry list tiles of the mixed sparse tiling in ONE grid (blocks [0, n2): 4-entry tiles, the rest: 16-entry
// tiles): launched separately, the 4-entry tiles -- 1/13 of the work, 320 workgroups at 80 rows -- cost a whole round of the chip.
struct RowsMixGeom {
    int off, cb, tiles_x, tiles_per_img, tiles_m;
};
__global__ __launch_bounds__(256, 2) void conv_halo_rows_mixed_kernel(ConvArgs a, RowsMixGeom g1, RowsMixGeom g2, int tiles_n) {
    const int n2 = g2.tiles_m * tiles_n;
    ConvArgs b = a;
    if ((int)blockIdx.x < n2) {
        b.up_sh_off = g2.off;
        b.up_sh_cb = g2.cb;
        conv_halo_rows_body<1, 2>(b, g2.tiles_x, g2.tiles_per_img, g2.tiles_m, tiles_n, (int)blockIdx.x);
    } else {
        b.up_sh_off = g1.off;
        b.up_sh_cb = g1.cb;
        conv_halo_rows_body<1, 1>(b, g1.tiles_x, g1.tiles_per_img, g1.tiles_m, tiles_n, (int)blockIdx.x - n2);
    }
}

int conv_halo_rows_slots() { return R_TW; }
int conv_halo_rows_sparse_halo_w() { return RowsCfg<1>::W; }

// dense / sparse upsample forms: the geometry the rows kernels tile (4 x 32 low-res pixels); everything else as
// conv_up_halo_supported (checked by the caller)
bool conv_halo_rows_up_supported(const ConvArgs& a) {
    if (a.h % R_TH != 0) return false;
    if (a.up_cols && (a.up_mix[0] | a.up_mix[1] | a.up_mix[2]) != 0) {  // mixed shapes
        if (a.up_mix[0] < 0 || a.up_mix[1] < 0 || a.up_mix[2] < 0) return false;
        if (a.up_mix[1] > 0 && a.h % (R_TH * 2) != 0) return false;
        if (a.up_mix[2] > 0 && a.h % (R_TH * 8) != 0) return false;
        return a.up_ntiles == a.up_mix[0] + a.up_mix[1] + a.up_mix[2] && a.up_npad == 32 * a.up_mix[0] + 16 * a.up_mix[1] + 4 * a.up_mix[2];
    }
    if (a.up_cols) return a.up_ntiles >= 1 && a.up_npad == a.up_ntiles * R_TW;
    return a.w % R_TW == 0;
}

int conv_halo_rows_sparse_halo_w_shape(int shape) { return shape == 1 ? RowsCfg<1, 1>::W : shape == 2 ? RowsCfg<1, 2>::W : RowsCfg<1, 0>::W; }

bool conv_halo_rows3_supported(const ConvArgs& a) { return a.h % R_TH == 0 && a.w % R_TW == 0; }

hipError_t conv_halo_rows_init() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_halo_rows_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       RowsCfg<0>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RowsCfg<1>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RowsCfg<2>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_kernel<1, 1>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (RowsCfg<1, 1>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_kernel<1, 2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                (RowsCfg<1, 2>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_persistent_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RowsCfg<0>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_halo_rows_tr_kernel<0, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (RowsCfg<0, 0, 2>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_halo_rows_tr_kernel<0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (RowsCfg<0, 0, 1>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_halo_rows_tr_kernel<2, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (RowsCfg<2, 0, 2>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)(conv_halo_rows_tr_kernel<2, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, (RowsCfg<2, 0, 1>::LDS_TOTAL));
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_splitk_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RowsCfg<0>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_splitk_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                RowsCfg<2>::LDS_TOTAL);
    if (e == hipSuccess)
        e = hipFuncSetAttribute((const void*)conv_halo_rows_mixed_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                std::max((int)RowsCfg<1, 1>::LDS_TOTAL, (int)RowsCfg<1, 2>::LDS_TOTAL));
    return e;
}

// the conv_up_halo_kernel<0 / 1> part of launch_conv_up_halo (the border ring has been launched by the caller)
hipError_t launch_conv_halo_rows_up(const ConvArgs& a, hipStream_t stream) {
    const bool sparse = a.up_cols != nullptr;
    const int tiles_n = a.cout / 64;
    if (sparse && (a.up_mix[0] | a.up_mix[1] | a.up_mix[2]) != 0) {
        // mixed shapes: one launch per shape over its own row blocks; the smallest launch (the 4-entry tiles) goes first
        dyf_form_note("conv_halo_rows_kernel<1>", a.n);
        constexpr int LDS_S0 = RowsCfg<1, 0>::LDS_TOTAL, LDS_S1 = RowsCfg<1, 1>::LDS_TOTAL, LDS_S2 = RowsCfg<1, 2>::LDS_TOTAL;
        const int cnt[3] = {a.up_mix[0], a.up_mix[1], a.up_mix[2]};
        const int off[3] = {0, 32 * cnt[0], 32 * cnt[0] + 16 * cnt[1]}, cb[3] = {0, cnt[0], cnt[0] + cnt[1]};
        const bool one_grid = !(dyf_form("DYF_SPARSE_MIXED_ONE_GRID") && atoi(dyf_form("DYF_SPARSE_MIXED_ONE_GRID")) == 0);
        if (one_grid && cnt[1] > 0 && cnt[2] > 0) {  // the 16- and 4-entry tiles in one grid; 32-entry tiles (if any) on their own below
            RowsMixGeom g1{off[1], cb[1], cnt[1], cnt[1] * (a.h / (2 * R_TH)), a.n * cnt[1] * (a.h / (2 * R_TH))};
            RowsMixGeom g2{off[2], cb[2], cnt[2], cnt[2] * (a.h / (8 * R_TH)), a.n * cnt[2] * (a.h / (8 * R_TH))};
            hipLaunchKernelGGL(conv_halo_rows_mixed_kernel, dim3((g1.tiles_m + g2.tiles_m) * tiles_n), dim3(256), std::max(LDS_S1, LDS_S2), stream, a,
                               g1, g2, tiles_n);
        }
        for (int sh = 2; sh >= 0; --sh) {
            if (cnt[sh] == 0 || (one_grid && sh != 0 && cnt[1] > 0 && cnt[2] > 0)) continue;
            ConvArgs b = a;
            b.up_sh_off = off[sh];
            b.up_sh_cb = cb[sh];
            const int rows = sh == 0 ? R_TH : sh == 1 ? 2 * R_TH : 8 * R_TH;
            const int tiles_x = cnt[sh], tiles_per_img = tiles_x * (a.h / rows), tiles_m = a.n * tiles_per_img;
            if (sh == 0)
                hipLaunchKernelGGL((conv_halo_rows_kernel<1, 0>), dim3(tiles_m * tiles_n), dim3(256), LDS_S0, stream, b, tiles_x,
                                   tiles_per_img, tiles_m, tiles_n);
            else if (sh == 1)
                hipLaunchKernelGGL((conv_halo_rows_kernel<1, 1>), dim3(tiles_m * tiles_n), dim3(256), LDS_S1, stream, b, tiles_x,
                                   tiles_per_img, tiles_m, tiles_n);
            else
                hipLaunchKernelGGL((conv_halo_rows_kernel<1, 2>), dim3(tiles_m * tiles_n), dim3(256), LDS_S2, stream, b, tiles_x,
                                   tiles_per_img, tiles_m, tiles_n);
        }
        return hipGetLastError();
    }
    const int tiles_x = sparse ? a.up_ntiles : a.w / R_TW, tiles_per_img = tiles_x * (a.h / R_TH);
    int tiles_m = a.n * tiles_per_img;
#ifdef DYF_EXPERIMENT_BUILD
    // timing experiment (WRONG results): 13 of 16 sparse tiles -- what packing the 52-column lists without padded slots would save
    const bool exp1316 = dyf_form("DYF_EXP_DEC5_1316") && atoi(dyf_form("DYF_EXP_DEC5_1316")) != 0;
    if (sparse && exp1316) tiles_m = tiles_m * 13 / 16;
#endif
    if (!sparse) {
        const long long sel4 = (long long)(a.n_sel > 0 ? a.n_sel : a.n) * tiles_per_img * tiles_n;
        const int tr = rows_tile_rows(sel4, a.h);
        if (tr != 4) {
            const int tpi = tiles_x * (a.h / tr), tm = a.n * tpi;
            dyf_form_note(tr == 2 ? "conv_halo_rows_kernel<0>+tr2" : "conv_halo_rows_kernel<0>+tr1", a.n);
            if (tr == 2)
                hipLaunchKernelGGL((conv_halo_rows_tr_kernel<0, 2>), dim3(tm * tiles_n), dim3(256), (RowsCfg<0, 0, 2>::LDS_TOTAL), stream, a, tiles_x, tpi, tm, tiles_n);
            else
                hipLaunchKernelGGL((conv_halo_rows_tr_kernel<0, 1>), dim3(tm * tiles_n), dim3(256), (RowsCfg<0, 0, 1>::LDS_TOTAL), stream, a, tiles_x, tpi, tm, tiles_n);
            return hipGetLastError();
        }
        const long long sel = sel4;
        const long long m_cout = (long long)a.n * a.ho * a.wo * a.cout;
        const int sk = rows_splitk_factor(a, sel, (long long)tiles_m * tiles_n, (a.c0 + a.c1) >> 6, m_cout);
        if (sk > 1) {
            ConvArgs b = a;
            b.splitk = sk;
            dyf_form_note("conv_halo_rows_kernel<0>+splitk", a.n);
            hipLaunchKernelGGL(conv_halo_rows_splitk_kernel<0>, dim3(tiles_m * tiles_n, sk), dim3(256), RowsCfg<0>::LDS_TOTAL, stream, b, tiles_x,
                               tiles_per_img, tiles_m, tiles_n);
            return launch_conv_splitk_finish4(b, (long long)a.n * a.ho * a.wo, stream);
        }
    }
    dyf_form_note(sparse ? "conv_halo_rows_kernel<1>" : "conv_halo_rows_kernel<0>", a.n);
    const int persist = dyf_form("DYF_ROWS_PERSISTENT") ? atoi(dyf_form("DYF_ROWS_PERSISTENT")) : 0;
    if (sparse)
        hipLaunchKernelGGL((conv_halo_rows_kernel<1, 0>), dim3(tiles_m * tiles_n), dim3(256), RowsCfg<1>::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n);
    else if (persist > 0 && tiles_m * tiles_n > persist)
        hipLaunchKernelGGL(conv_halo_rows_persistent_kernel<0>, dim3(persist), dim3(256), RowsCfg<0>::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n);
    else
        hipLaunchKernelGGL(conv_halo_rows_kernel<0>, dim3(tiles_m * tiles_n), dim3(256), RowsCfg<0>::LDS_TOTAL, stream, a, tiles_x,
                           tiles_per_img, tiles_m, tiles_n);
    return hipGetLastError();
}

hipError_t launch_conv_halo_rows3(const ConvArgs& a, hipStream_t stream) {
    const int tiles_x = a.w / R_TW, tiles_per_img = tiles_x * (a.h / R_TH);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 256;
    {
        const long long sel = (long long)(a.n_sel > 0 ? a.n_sel : a.n) * tiles_per_img * tiles_n;
        // The plain form keeps its four-row tiles (+ split-K) by default: isolated, its two-row tiles look better (dec2 at 10 rows 48.5
        // against 53.3 + 8 us), INSIDE the rollout they are worse (80.7 us against 53.5 + 7.9: the 16-chunk weight stream of dec2, twice as
        // long per MFMA with half tiles, comes from the Infinity Cache there, not from a warm L2) -- whole rollouts, same box, two runs each,
        // ms at 1 / 4 / 7 / 10 rows: short tiles for the upsample form only 8.90 / 14.69 / 20.20 / 26.37, for both forms 8.90 / 14.79 / 20.39 /
        // 26.68, four-row tiles only 9.09 / 14.85 / 20.28 / 26.46.  DYF_ROWS_TR_PLAIN=1 (or DYF_ROWS_TR) applies the rule here too.
        const char* tpe = dyf_form("DYF_ROWS_TR_PLAIN");
        const int tr = ((tpe && atoi(tpe) != 0) || dyf_form("DYF_ROWS_TR")) ? rows_tile_rows(sel, a.h) : 4;
        if (tr != 4) {
            const int tpi = tiles_x * (a.h / tr), tm = a.n * tpi;
            dyf_form_note(tr == 2 ? "conv_halo_rows_kernel<2>+tr2" : "conv_halo_rows_kernel<2>+tr1", a.n);
            if (tr == 2)
                hipLaunchKernelGGL((conv_halo_rows_tr_kernel<2, 2>), dim3(tm * tiles_n), dim3(256), (RowsCfg<2, 0, 2>::LDS_TOTAL), stream, a, tiles_x, tpi, tm, tiles_n);
            else
                hipLaunchKernelGGL((conv_halo_rows_tr_kernel<2, 1>), dim3(tm * tiles_n), dim3(256), (RowsCfg<2, 0, 1>::LDS_TOTAL), stream, a, tiles_x, tpi, tm, tiles_n);
            return hipGetLastError();
        }
        const int sk = rows_splitk_factor(a, sel, (long long)tiles_m * tiles_n, (a.c0 + a.c1) >> 6, (long long)a.n * a.ho * a.wo * a.cout);
        if (sk > 1) {
            ConvArgs b = a;
            b.splitk = sk;
            dyf_form_note("conv_halo_rows_kernel<2>+splitk", a.n);
            hipLaunchKernelGGL(conv_halo_rows_splitk_kernel<2>, dim3(tiles_m * tiles_n, sk), dim3(256), RowsCfg<2>::LDS_TOTAL, stream, b, tiles_x,
                               tiles_per_img, tiles_m, tiles_n);
            return launch_conv_splitk_finish4(b, (long long)a.n * a.ho * a.wo, stream);
        }
    }
    dyf_form_note("conv_halo_rows_kernel<2>", a.n);
    hipLaunchKernelGGL(conv_halo_rows_kernel<2>, dim3(tiles_m * tiles_n), dim3(256), RowsCfg<2>::LDS_TOTAL, stream, a, tiles_x,
                       tiles_per_img, tiles_m, tiles_n);
    return hipGetLastError();
}
