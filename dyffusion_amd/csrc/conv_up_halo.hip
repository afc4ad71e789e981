// Fused x2-bilinear-upsample + 3x3 conv, "halo" form (gfx950): the dominant kernel of the Navier-Stokes backbone
// (decoder blocks dec3-dec5 of src/models/unet_simple.py:40-52, 69 % of a forward's FLOPs).
//
// Same mathematics as conv_igemm_kernel<.., UP=1> (phase decomposition + border-correction taps, see conv.hip), but the
// data movement is re-designed around what bounds that kernel: it re-gathers every input pixel once per tap and phase
// (36x) through the CU's texture path (TA ~64 B/clk) and synchronises its 4 waves every 16 MFMAs.  Here
//   * a workgroup owns a 16x16 LOW-res tile and ALL FOUR output phases of 64 output channels: the GEMM tile is
//     M = 256 pixels x N = 256 (4 phases x 64 channels); each wave owns 64 pixels x 256 columns = 2 x 8 accumulator
//     tiles of v_mfma_f32_32x32x16_bf16 (256 accumulator registers; one wave per SIMD owns the whole 512-entry file);
//   * per 64-channel chunk the 18x18 replicate-padded input window ("halo", 41 KB) is DMA'd into LDS ONCE and all 9
//     stencil taps (and the correction taps) read their A fragments from it at shifted addresses: input traffic per
//     MFMA drops ~30x, and A fragments are shared by the 4 phases (LDS reads per MFMA: 0.625 KB vs 1 KB);
//   * a K step is one tap of one chunk: 64 MFMAs per wave between barriers (4x fewer barriers), B tile = 32 KB
//     (4 phases x 64 channels x 64 k) double-buffered, next step's tile in flight for a whole 2 048-cycle step;
//   * halo double-buffered across chunks; LDS total 147 KB -> one workgroup per CU, latency hidden by ILP inside the wave
//     (~2 non-MFMA instructions per MFMA).
#include "conv.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {

constexpr int HALO_W = 18, HALO_PIX = 328;              // 18*18 = 324, padded to a multiple of 8 DMA rows
constexpr int HALO_BYTES = HALO_PIX * 128;              // 41 984
constexpr int ZERO_OFF = 2 * HALO_BYTES;                // 128 B of zeros (A rows masked out of a correction tap)
constexpr int B_OFF = ZERO_OFF + 512;                   // two 32 KB weight stages
constexpr int B_BYTES = 256 * 128;
constexpr int LDS_TOTAL = B_OFF + 2 * B_BYTES;          // 150 016 B
constexpr int HALO_INSTR = HALO_PIX / 8;                // 41 wave-level DMA instructions per halo
constexpr int NWAVES = 8;                               // 512 threads: two waves per SIMD cover each other's issue gaps
constexpr int HALO_PER_WAVE = (HALO_INSTR + NWAVES - 1) / NWAVES;  // 6

}  // namespace

__global__ __launch_bounds__(512, 2) void conv_up_halo_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m,
                                                              int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    // wave -> (pixel group wm: tile rows 4*wm .. 4*wm+3 = 64 pixels, column half wn: phases py = wn, 128 columns)
    const int wm = wave >> 1, wn = wave & 1;

    // XCD-aware tile id; the column blocks of one tile are consecutive (they share the halo in L2)
    const int total = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int n_img = tm / tiles_per_img;
    const int t_in = tm - n_img * tiles_per_img;
    const int ty0 = (t_in / tiles_x) * 16, tx0 = (t_in % tiles_x) * 16;

    const int cin = a.c0 + a.c1;
    const int cpt = cin >> 6;
    const bool has_top = ty0 == 0, has_bot = ty0 + 16 == a.h, has_left = tx0 == 0, has_right = tx0 + 16 == a.w;
    const bool has_row = has_top || has_bot, has_col = has_left || has_right;
    // tap list of this tile, 4 bits per entry: 0-8 stencil, 9-11 row correction, 12-14 column correction, 15 corner
    unsigned long long tap_list = 0x876543210ull;
    int ntaps = 9;
    if (has_row) { tap_list |= 0xBA9ull << (4 * ntaps); ntaps += 3; }
    if (has_col) { tap_list |= 0xEDCull << (4 * ntaps); ntaps += 3; }
    if (has_row && has_col) { tap_list |= 0xFull << (4 * ntaps); ntaps += 1; }
    const int nsteps = ntaps * cpt;

    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_up, 0, (int)(unsigned)((size_t)4 * a.cout * 16 * cin * 2),
                                                          0x00020000);

    // ---- halo DMA descriptors: instruction i (i % 4 == wave) fills halo pixels [8i, 8i+8); lane -> (pixel, 16-B chunk)
    const int sub = lane >> 3;
    unsigned h_off[HALO_PER_WAVE];
#pragma unroll
    for (int j = 0; j < HALO_PER_WAVE; ++j) {
        const int i = j * NWAVES + wave;
        int hp = i * 8 + sub;
        if (hp > 323) hp = 323;  // padding slots re-read the last halo pixel
        const int hy = hp / HALO_W, hx = hp - hy * HALO_W;
        const int y = min(max(ty0 - 1 + hy, 0), a.h - 1), x = min(max(tx0 - 1 + hx, 0), a.w - 1);  // replicate clamp
        const int gch = (lane & 7) ^ ((hp >> 1) & 7);  // swizzled source chunk of this linear LDS slot
        h_off[j] = (unsigned)((n_img * a.h + y) * a.w + x) * (unsigned)(a.c0 * 2) + gch * 16;  // c0 == c1 (checked on host)
    }
    // ---- weight DMA descriptors: B row r in [0,256): phase r>>6, channel tn*64 + (r & 63); instruction j*4+wave
    unsigned b_off[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = (j * NWAVES + wave) * 8 + sub;
        const int gch = (lane & 7) ^ ((r >> 1) & 7);
        b_off[j] = (unsigned)((r >> 6) * a.cout + tn * 64 + (r & 63)) * (unsigned)(16 * cin * 2) + gch * 16;
    }
    if (tid < 32) ((uint4*)(smem + ZERO_OFF))[tid] = make_uint4(0, 0, 0, 0);

    int is_step = 0, is_pos = 0, is_chunk = 0;  // issue-side iterator over (chunk, tap-list position)
    auto issue_halo = [&](int chunk) {
        const int cb = chunk << 6;
        const bool second = cb >= a.c0;
        const unsigned coff = (unsigned)((second ? cb - a.c0 : cb) * 2);
        char* dst = smem + (chunk & 1) * HALO_BYTES;
#pragma unroll
        for (int j = 0; j < HALO_PER_WAVE; ++j) {
            const int i = j * NWAVES + wave;
            if (i < HALO_INSTR) {
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(dst + i * 1024), 16, h_off[j] + coff, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(dst + i * 1024), 16, h_off[j] + coff, 0, 0, 0);
            }
        }
    };
    auto issue_b = [&]() {  // weights of step is_step into stage (is_step & 1); advances the iterator
        if (is_step >= nsteps) return;
        const int tap = (int)((tap_list >> (4 * is_pos)) & 15ull);
        char* dst = smem + B_OFF + (is_step & 1) * B_BYTES;
        const int soff = (tap * cin + (is_chunk << 6)) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, LDS_PTR(dst + (j * NWAVES + wave) * 1024), 16, b_off[j], soff, 0, 0);
        ++is_step;
        if (++is_pos == ntaps) {
            is_pos = 0;
            ++is_chunk;
        }
    };

    f32x16 acc[4][2];  // [local column tile; global = wn*4 + nt = phase*2 + half][pixel tile]
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;

    // tile pixel of this lane's A rows: pixel tile mt covers tile rows 4*wm + 2*mt + {0,1}
    int hp0[2];      // halo pixel index of the un-shifted tap
    bool m_top[2], m_bot[2], m_left, m_right;
    {
        const int x = l31 & 15;
        m_left = has_left && x == 0;
        m_right = has_right && x == 15;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int y = 4 * wm + 2 * mt + (l31 >> 4);
            hp0[mt] = (y + 1) * HALO_W + (x + 1);
            m_top[mt] = has_top && y == 0;
            m_bot[mt] = has_bot && y == 15;
        }
    }
    const int b_row_off = l31 * 128;
    const unsigned lds_base = (unsigned)(uintptr_t)LDS_PTR(smem);
    const int bkey = (l31 >> 1) & 7;

    int cs_step = 0, cs_chunk = 0;  // compute-side step counter / chunk

    // One K step: A fragments of (tap displacement d, row masks) for both pixel tiles, then MFMAs into the column tiles
    // selected by NT_MASK (bit nt).  MASKED: rows whose `keep` flag is false read the zero page.
#define DSR(dst, addr, OFF) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "i"(OFF));
    // A fragments are read with inline asm as well: every LDS read inside the K loop is hand-counted, so the compiler
    // inserts no (over-conservative) lgkmcnt waits between the MFMAs.
#define LOAD_A(DISP, KEEP0, KEEP1, MASKED)                                                                   \
    bf16x8 af[2][4];                                                                                         \
    {                                                                                                        \
        const unsigned Hs = lds_base + (cs_chunk & 1) * HALO_BYTES;                                          \
        _Pragma("unroll") for (int mt = 0; mt < 2; ++mt) {                                                   \
            const int hp = hp0[mt] + (DISP);                                                                 \
            const int key = (hp >> 1) & 7;                                                                   \
            const bool keep = mt == 0 ? (KEEP0) : (KEEP1);                                                   \
            _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                               \
                const unsigned addr = ((MASKED) && !keep) ? lds_base + ZERO_OFF                              \
                                                           : Hs + hp * 128 + (((ks * 2 + hi) ^ key) << 4);   \
                DSR(af[mt][ks], addr, 0)                                                                     \
            }                                                                                                \
        }                                                                                                    \
    }
#define MMA_COLS(NT_MASK)                                                                                    \
    {                                                                                                        \
        const unsigned Bb = lds_base + B_OFF + (cs_step & 1) * B_BYTES + b_row_off + wn * 16384;             \
        _Pragma("unroll") for (int ks = 0; ks < 4; ++ks) {                                                   \
            const unsigned ad = Bb + (((ks * 2 + hi) ^ bkey) << 4);                                          \
            _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                               \
                if (((NT_MASK) >> (wn * 4 + nt)) & 1) {                                                      \
                    bf16x8 bf;                                                                               \
                    const unsigned adn = ad + nt * 4096;                                                     \
                    DSR(bf, adn, 0)                                                                          \
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                       \
                    __builtin_amdgcn_sched_barrier(0);                                                       \
                    acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][ks], bf, acc[nt][0], 0, 0, 0); \
                    acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][ks], bf, acc[nt][1], 0, 0, 0); \
                }                                                                                            \
            }                                                                                                \
        }                                                                                                    \
    }
    // All 8 column tiles (the stencil taps, > 90 % of the MFMAs).  hipcc sinks every ds_read next to its first use (one
    // wave per SIMD then eats the ~100-cycle LDS latency before every pair of MFMAs), so the B-fragment reads are issued
    // as inline asm with hand-counted waits (cdna_hip_programming.md 5.7): two fragment sets, the reads of k16 sub-step
    // ks+1 are in flight under the 16 MFMAs of sub-step ks.
#define DSR4(Q, addr) DSR(Q[0], addr, 0) DSR(Q[1], addr, 4096) DSR(Q[2], addr, 8192) DSR(Q[3], addr, 12288)
#define LGKM_WAIT(N)                                                      \
    asm volatile("s_waitcnt lgkmcnt(" #N ")" ::: "memory");               \
    __builtin_amdgcn_sched_barrier(0);
#define MFMA8(KS, Q)                                                                                         \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                       \
        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[0][KS], Q[nt], acc[nt][0], 0, 0, 0);         \
        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[1][KS], Q[nt], acc[nt][1], 0, 0, 0);         \
    }
    // stencil step: A (2 pixel tiles) and B (4 column tiles) fragments of k16 sub-step ks+1 stream in while the 8 MFMAs of
    // sub-step ks execute; two register sets of 6 fragments
#define RD6(AQ, Q, KS)                                                                                       \
    {                                                                                                        \
        const unsigned ka0 = pa0 + ((((KS) * 2 + hi) ^ key0) << 4), ka1 = pa1 + ((((KS) * 2 + hi) ^ key1) << 4); \
        const unsigned kb = ba + ((((KS) * 2 + hi) ^ bkey) << 4);                                            \
        DSR(AQ[0], ka0, 0) DSR(AQ[1], ka1, 0) DSR4(Q, kb)                                                    \
    }
#define MFMA8X(AQ, Q)                                                                                        \
    _Pragma("unroll") for (int nt = 0; nt < 4; ++nt) {                                                       \
        acc[nt][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ[0], Q[nt], acc[nt][0], 0, 0, 0);             \
        acc[nt][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(AQ[1], Q[nt], acc[nt][1], 0, 0, 0);             \
    }
#define STENCIL_MMA(DISP)                                                                                    \
    {                                                                                                        \
        const unsigned Hs = lds_base + (cs_chunk & 1) * HALO_BYTES;                                          \
        const int hpa = hp0[0] + (DISP), hpb = hp0[1] + (DISP);                                              \
        const int key0 = (hpa >> 1) & 7, key1 = (hpb >> 1) & 7;                                              \
        const unsigned pa0 = Hs + hpa * 128, pa1 = Hs + hpb * 128;                                           \
        const unsigned ba = lds_base + B_OFF + (cs_step & 1) * B_BYTES + b_row_off + wn * 16384;             \
        bf16x8 aq0[2], aq1[2], q0[4], q1[4];                                                                 \
        RD6(aq0, q0, 0)                                                                                      \
        RD6(aq1, q1, 1)                                                                                      \
        LGKM_WAIT(6)                                                                                         \
        MFMA8X(aq0, q0)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        RD6(aq0, q0, 2)                                                                                      \
        LGKM_WAIT(6)                                                                                         \
        MFMA8X(aq1, q1)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
        RD6(aq1, q1, 3)                                                                                      \
        LGKM_WAIT(6)                                                                                         \
        MFMA8X(aq0, q0)                                                                                      \
        LGKM_WAIT(0)                                                                                         \
        MFMA8X(aq1, q1)                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
    }
    // every step: wait for this wave's DMAs, workgroup barrier (data visible + previous stage free), prefetch the next
    // step's weights (and, at the first step of a chunk, the next chunk's halo)
#define STEP_BEGIN()                                                          \
    {                                                                         \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                      \
        __syncthreads();                                                      \
        issue_b();                                                            \
    }
#define STEP_END() { ++cs_step; }

    issue_halo(0);
    issue_b();
    for (cs_chunk = 0; cs_chunk < cpt; ++cs_chunk) {
        // keep the per-tap LDS addresses from being hoisted out of the chunk loop (9 taps x 8 addresses would eat the
        // registers the B-fragment double buffer needs)
        asm volatile("" : "+v"(hp0[0]), "+v"(hp0[1]));
        // ---- 9 stencil taps (a, b) in {-1,0,1}^2: displacement a*18 + b in the halo, every phase
#define STENCIL(T)                                                                           \
        {                                                                                    \
            STEP_BEGIN()                                                                     \
            if ((T) == 0 && cs_chunk + 1 < cpt) issue_halo(cs_chunk + 1);                     \
            STENCIL_MMA(((T) / 3 - 1) * HALO_W + ((T) % 3 - 1))                               \
                                                                                             \
            STEP_END()                                                                       \
        }
        STENCIL(0) STENCIL(1) STENCIL(2) STENCIL(3) STENCIL(4) STENCIL(5) STENCIL(6) STENCIL(7) STENCIL(8)
#undef STENCIL
        if (has_row) {  // taps 9-11: b = -1,0,+1 on the border row; top -> phases py=0 (cols 0-3), bottom -> py=1 (cols 4-7)
#define ROWCORR(B_)                                                                          \
            {                                                                                \
                STEP_BEGIN()                                                                 \
                if (has_top) { LOAD_A((B_), m_top[0], m_top[1], true) MMA_COLS(0x0F) }       \
                if (has_bot) { LOAD_A((B_), m_bot[0], m_bot[1], true) MMA_COLS(0xF0) }       \
                STEP_END()                                                                   \
            }
            ROWCORR(-1) ROWCORR(0) ROWCORR(1)
#undef ROWCORR
        }
        if (has_col) {  // taps 12-14: a = -1,0,+1 on the border column; left -> px=0 (cols 0,1,4,5), right -> px=1 (2,3,6,7)
#define COLCORR(A_)                                                                          \
            {                                                                                \
                STEP_BEGIN()                                                                 \
                if (has_left) { LOAD_A((A_) * HALO_W, m_left, m_left, true) MMA_COLS(0x33) } \
                if (has_right) { LOAD_A((A_) * HALO_W, m_right, m_right, true) MMA_COLS(0xCC) } \
                STEP_END()                                                                   \
            }
            COLCORR(-1) COLCORR(0) COLCORR(1)
#undef COLCORR
        }
        if (has_row && has_col) {  // tap 15: the corner pixel, one phase per corner
            STEP_BEGIN()
            if (has_top && has_left) { LOAD_A(0, m_top[0] && m_left, m_top[1] && m_left, true) MMA_COLS(0x03) }
            if (has_top && has_right) { LOAD_A(0, m_top[0] && m_right, m_top[1] && m_right, true) MMA_COLS(0x0C) }
            if (has_bot && has_left) { LOAD_A(0, m_bot[0] && m_left, m_bot[1] && m_left, true) MMA_COLS(0x30) }
            if (has_bot && has_right) { LOAD_A(0, m_bot[0] && m_right, m_bot[1] && m_right, true) MMA_COLS(0xC0) }
            STEP_END()
        }
    }
#undef LOAD_A
#undef MMA_COLS
#undef STENCIL_MMA
#undef MFMA8X
#undef RD6
#undef MFMA8
#undef LGKM_WAIT
#undef DSR4
#undef DSR
#undef STEP_BEGIN
#undef STEP_END

    // ---- epilogue, two rounds: waves wn = 0 / 1 park phase (py = wn, px = round) in LDS tile wn (fp32 [256 px][64 ch],
    // the whole LDS is free now), then all 512 threads apply affine/act/dropout and store bf16 NHWC
    float* Ct = (float*)smem;
    const uint32_t key = drop_key(a.drop);
#pragma unroll
    for (int px = 0; px < 2; ++px) {
        __syncthreads();
#pragma unroll
        for (int half = 0; half < 2; ++half)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ml = wm * 64 + mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi;
                    Ct[wn * 16384 + ml * 64 + half * 32 + l31] = acc[px * 2 + half][mt][r];
                }
        __syncthreads();
#pragma unroll 2
        for (int it = 0; it < 8; ++it) {
            const int id = it * 512 + tid;
            const int py = id >> 11, row = (id >> 3) & 255, cg = id & 7;
            const int y = ty0 + (row >> 4), x = tx0 + (row & 15);
            const size_t m = ((size_t)n_img * a.ho + 2 * y + py) * a.wo + 2 * x + px;
            const int co = tn * 64 + cg * 8;
            const float* cp = Ct + py * 16384 + row * 64 + cg * 8;
            const float4 v0 = *(const float4*)cp;
            const float4 v1 = *(const float4*)(cp + 4);
            const size_t ci = (size_t)n_img * a.coef_stride + co;
            const float4 a0 = *(const float4*)(a.coef_a + ci), a1 = *(const float4*)(a.coef_a + ci + 4);
            const float4 c0 = *(const float4*)(a.coef_c + ci), c1 = *(const float4*)(a.coef_c + ci + 4);
            float v[8] = {fmaf(v0.x, a0.x, c0.x), fmaf(v0.y, a0.y, c0.y), fmaf(v0.z, a0.z, c0.z), fmaf(v0.w, a0.w, c0.w),
                          fmaf(v1.x, a1.x, c1.x), fmaf(v1.y, a1.y, c1.y), fmaf(v1.z, a1.z, c1.z), fmaf(v1.w, a1.w, c1.w)};
            const uint32_t e0 = (uint32_t)(m * a.cout + co);
#pragma unroll
            for (int t = 0; t < 8; ++t) {
                v[t] = apply_act(v[t], a.act);
                v[t] = drop_apply(v[t], e0 + t, a.drop, key);
            }
            uint4 o;
            o.x = pack_bf16x2(v[0], v[1]);
            o.y = pack_bf16x2(v[2], v[3]);
            o.z = pack_bf16x2(v[4], v[5]);
            o.w = pack_bf16x2(v[6], v[7]);
            *(uint4*)(a.out_bf16 + m * a.cout + co) = o;
        }
    }
#endif
}

bool conv_up_halo_supported(const ConvArgs& a) {
    if (!a.up2x || a.wpk_up == nullptr || a.out_bf16 == nullptr || a.residual != nullptr) return false;
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && (a.c1 == 0 || a.c1 == a.c0) && a.cout % 64 == 0)) return false;
    if (a.h % 16 != 0 || a.w % 16 != 0 || a.ho != 2 * a.h || a.wo != 2 * a.w) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && (size_t)4 * a.cout * 16 * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t conv_up_halo_init() {
    return hipFuncSetAttribute((const void*)conv_up_halo_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TOTAL);
}

hipError_t launch_conv_up_halo(const ConvArgs& a, hipStream_t stream) {
    const int tiles_x = a.w / 16, tiles_per_img = tiles_x * (a.h / 16);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 64;
    hipLaunchKernelGGL(conv_up_halo_kernel, dim3(tiles_m * tiles_n), dim3(512), LDS_TOTAL, stream, a, tiles_x, tiles_per_img,
                       tiles_m, tiles_n);
    return hipGetLastError();
}
