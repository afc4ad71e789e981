// conv_gn16_kernel: 3 x 3 / stride 1 / pad 1 convolution WITH the GroupNorm + FiLM + SiLU + Dropout (+ residual) of its Block fused
// into the epilogue (ResnetBlock of src/models/unet.py:58-109; gn_fused.h) on SMALL tiles -- 16 x 16 pixels x 64 output channels per
// workgroup, THREE workgroups per CU.  Round 6; serves every fused-GroupNorm conv of the ResNet-UNet at the OISST shapes: the 64- / 128-
// channel levels (60 x 60, 30 x 30), the 256-channel 15 x 15 level (four 64-channel blocks) and the up path's two-source inputs
// cat([x, skip]) of unequal channel counts (conv.hip launch_conv_gn_fused holds the policy).
//
// Why a second form beside conv_up_halo_kernel<5, 2> (16 x 32 tiles, 128 accumulator registers per wave, 80 KB of LDS, two workgroups
// per CU): a phase timeline of that kernel (tools/timeline_oisst.py, profiles/r06_halo5_timeline.txt) shows a tile's life as a SERIAL
// chain -- 10-15 k cycles until its 78 KB halo has landed, 11-14 k of K loop (9.2 k of matrix work), 4-5 k statistics, 4-10 k waiting
// for the sample's other tiles, 9-20 k epilogue -- i.e. the matrix pipe works for 15-18 % of a wave's life and NOTHING on the CU
// overlaps it except the one other workgroup (which started at the same time and is in the same phase; delaying it changed nothing:
// the launch is bound by the chain's latency, not by a shared pipe).  With two workgroups per CU set by LDS and by the 128-register
// accumulator, the only lever is a smaller tile:
//   * 16 x 16 pixels: halo 18 x 18 x 128 B = 41.5 KB -> three workgroups (12 waves) per CU, 64 accumulator registers per wave
//     (<= 168 registers: three waves per SIMD), every phase of the chain half as long, four times the tiles (608 instead of 304 at
//     38 OISST rows: the few-rows regime fills the chip);
//   * a wave owns 4 rows x 16 columns = two 32-pixel MFMA tiles x 64 channels: per k16 sub-step 2 pixel fragments from LDS, 2
//     weight fragments straight from the L2-resident fragment stream (pack_halo3_frag64: the SP = 5 layout) and 4 MFMAs;
//   * ONE statistics slot per workgroup: the four waves' (sum, sum of squares) meet in LDS, wave 0 adds them in wave order and
//     publishes 16 granules; 16 slots per sample at 60 x 60 (the 16 x 32 form: 32);
//   * operands swapped (D^T = W X^T) as in the halo kernels: an accumulator lane holds 4 consecutive channels of one pixel and the
//     epilogue runs straight out of the accumulators.
// The K loop is plain HIP (no hand-placed asm): with 12 waves per CU the scheduler has other waves to issue while one waits.
#include "conv.h"
#include "gn_fused.h"

#include <algorithm>
#include <type_traits>

typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

namespace {
constexpr int T16 = 16;                       // tile rows = tile columns
constexpr int HW16 = 18;                      // halo width / height in pixels
constexpr int REAL16 = HW16 * HW16;           // 324 halo pixels
constexpr int PIX16 = (REAL16 + 7) / 8 * 8;   // 328: padded to whole 8-pixel DMA instructions
constexpr int HALO16_BYTES = PIX16 * 128;     // 41 984 B
constexpr int INSTR16 = PIX16 / 8;            // 41 wave-level DMA instructions per halo
constexpr int PER_WAVE16 = (INSTR16 + 3) / 4; // 11
constexpr int RED16_OFF = HALO16_BYTES;       // [4 waves][16] floats: the waves' statistics
constexpr int COEF16_OFF = RED16_OFF + 256;   // [64] A + [64] C
constexpr int LDS16_TOTAL = COEF16_OFF + 512; // 42 752 B: three workgroups per CU
constexpr unsigned WSTEP16 = 8192u;           // weight bytes of one (tap, chunk) step of a 64-channel block: [ks][nt][lane] x 16 B
}  // namespace

#ifdef HALO_EXP_TIMELINE  // experiment builds only (tools/build_variant.sh, tools/timeline_oisst.py): per-wave shader-clock stamps at the phase boundaries
__device__ unsigned long long g_gn16_tl[1 << 18];
#define TL16(K) if (blockIdx.x < 8192 && lane == 0) g_gn16_tl[(blockIdx.x * 4 + wave) * 8 + (K)] = __builtin_amdgcn_s_memtime();
#include <cstdio>
#include <string>
#include <vector>
#else
#define TL16(K)
#endif
__global__ __launch_bounds__(256, 3) void conv_gn16_kernel(ConvArgs a, int tiles_x, int tiles_per_img, int tiles_m, int tiles_n) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    TL16(0)

    // XCD-aware tile id (block b runs on XCD b % 8): every XCD walks a contiguous range of tiles, the column blocks of one tile are
    // consecutive (they share the halo in L2)
    const int total = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = tile % tiles_n, tm = tile / tiles_n;
    const int n_img = tm / tiles_per_img;
    const int t_in = tm - n_img * tiles_per_img;
    const int ty0 = (t_in / tiles_x) * T16, tx0 = (t_in % tiles_x) * T16;
    const int px_x = l31 & 15, px_r = l31 >> 4;
    const int col = tx0 + px_x;
    const bool lane_valid = col < a.w;  // ragged planes: columns beyond the image are computed on zeros and neither counted nor stored

    const int cin = a.c0 + a.c1;
    const int cpt = cin >> 6;
    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_up_frag, 0, (int)(unsigned)((size_t)a.cout * 16 * cin * 2), 0x00020000);

    // LDS swizzle key of halo pixel hp (XORed into its 16-B chunk index): the key of the 18-wide halos of conv_up_halo.hip -- a
    // ds_read_b128 of 2 tile rows x 16 columns then hits 16 distinct slots in each of its 16-lane groups for every tap displacement
#define HKEY16(hp) ((((hp) >> 1) - (int)((unsigned)(hp) / (unsigned)HW16)) & 7)
    // ---- halo DMA: instruction i (i % 4 == wave) fills halo pixels [8 i, 8 i + 8); lane -> (pixel i * 8 + lane / 8, 16-B chunk lane % 8).
    // Out-of-image pixels (the conv's zero padding, ragged edges) get an out-of-range offset: the DMA's bounds check writes zeros.
    unsigned h_reg[PER_WAVE16];
    {
        const int sub = lane >> 3;
#pragma unroll
        for (int j = 0; j < PER_WAVE16; ++j) {
            int hp = (j * 4 + wave) * 8 + sub;
            if (hp > REAL16 - 1) hp = REAL16 - 1;  // padding slots re-read the last halo pixel
            const int hy = hp / HW16, hx = hp - hy * HW16;
            const int yy = ty0 - 1 + hy, xx = tx0 - 1 + hx;
            const int gch = (lane & 7) ^ HKEY16(hp);
            const bool inside = yy >= 0 && yy < a.h && xx >= 0 && xx < a.w;
            // (pixel index | swizzled chunk << 28: the byte offset is formed per chunk -- the two sources of a concatenated input have
            // different pixel strides, c0 and c1 channels)
            h_reg[j] = inside ? (unsigned)((n_img * a.h + yy) * a.w + xx) | ((unsigned)gch << 28) : 0xFFFFFFFFu;
        }
    }
    auto issue_halo = [&](int chunk) {
        const int cb = chunk << 6;
        const bool second = cb >= a.c0;  // torch.cat([src0, src1], dim=channels): chunks [0, c0 / 64) from src0, the rest from src1
        const unsigned coff = (unsigned)((second ? cb - a.c0 : cb) * 2);
        const unsigned pstride = (unsigned)((second ? a.c1 : a.c0) * 2);
#pragma unroll
        for (int j = 0; j < PER_WAVE16; ++j) {
            const int i = j * 4 + wave;
            if (i < INSTR16) {
                unsigned vo = h_reg[j];
                if (vo != 0xFFFFFFFFu) vo = (vo & 0x0FFFFFFFu) * pstride + (vo >> 28) * 16u + coff;
                if (second)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(smem + i * 1024), 16, vo, 0, 0, 0);
                else
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(smem + i * 1024), 16, vo, 0, 0, 0);
            }
        }
    };

    f32x16 acc[2][2];  // [32-channel half nt][pixel tile mt: rows 4 wave + 2 mt + {0, 1}]
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][mt][r] = 0.0f;

    const int hp0 = (px_r + 1 + 4 * wave) * HW16 + (px_x + 1);  // halo pixel of pixel tile 0 at the centre tap; mt adds 2 rows
    const unsigned w_voff = (unsigned)lane * 16u;
    for (int chunk = 0; chunk < cpt; ++chunk) {
        if (chunk > 0) __syncthreads();  // every wave is done reading the previous chunk's halo
        issue_halo(chunk);
        const unsigned wbase = (unsigned)((tn * cpt + chunk) * 16) * WSTEP16;
        // weight fragments: ring of WAHEAD + 1 sets, requested WAHEAD sub-steps ahead (an L2 round trip is longer than one sub-step's 4 MFMAs)
#ifndef G16_WAHEAD
#define G16_WAHEAD 5
#endif
        constexpr int WAHEAD = G16_WAHEAD, WR = WAHEAD + 1;
        u32x4 wq[WR][2];
#pragma unroll
        for (int p = 0; p < WAHEAD; ++p)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
                wq[p][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + nt * 1024, wbase + (unsigned)(p >> 2) * WSTEP16 + (unsigned)(p & 3) * 2048u, 0);
        // this wave's part of the halo has landed (the 2 x WAHEAD weight loads behind it may still fly) ...
        if constexpr (WAHEAD == 3) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else if constexpr (WAHEAD == 5) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if constexpr (WAHEAD == 7) asm volatile("s_waitcnt vmcnt(14)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // ... and so has everybody else's
        if (chunk == 0) { TL16(1) }
        el16x8_t pq[2][2];
        unsigned ab[2], ax[2];
        auto tap_addr = [&](int tap) {
            const int d = (tap / 3 - 1) * HW16 + (tap % 3 - 1);
            int hpb = hp0;
            asm volatile("" : "+v"(hpb));  // opaque: keeps the 9 taps' addresses from being computed up front (and spilled)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int hpm = hpb + d + 2 * HW16 * mt;
                ab[mt] = (unsigned)hpm * 128u;
                ax[mt] = (unsigned)((hi ^ HKEY16(hpm)) << 4);
            }
        };
        tap_addr(0);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) pq[0][mt] = *(const el16x8_t*)(smem + ab[mt] + ax[mt]);
        // 36 k16 sub-steps (9 taps x 4): pixel fragments of sub-step s + 1 and weight fragments of s + 3 are requested before the MFMAs of s
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            const int cur = s & 1, nxt = cur ^ 1;
            if (s + WAHEAD < 36) {
                const int tw = (s + WAHEAD) >> 2, kw = (s + WAHEAD) & 3;
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
                    wq[(s + WAHEAD) % WR][nt] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_voff + nt * 1024, wbase + tw * WSTEP16 + kw * 2048u, 0);
            }
            if (s + 1 < 36) {
                const int tap1 = (s + 1) >> 2, ks1 = (s + 1) & 3;
                if (ks1 == 0) tap_addr(tap1);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) pq[nxt][mt] = *(const el16x8_t*)(smem + ab[mt] + (ax[mt] ^ (unsigned)(ks1 << 5)));
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, wq[s % WR][nt]), pq[cur][mt], acc[nt][mt], 0, 0, 0);
        }
    }

    // ---- epilogue straight from the accumulators: lane (l31, hi) of tile (nt, mt) holds pixel l31 of pixel tile mt and channels
    // nt * 32 + 8 g + 4 hi + {0..3} (g = register group r >> 2)
    TL16(2)
    const GnFuse& G = a.gnf;
    const RngKey key = drop_row_key(a.drop, n_img);
    const uint32_t row0 = (uint32_t)n_img * (uint32_t)(a.ho * a.wo * a.cout);  // dropout streams are per batch row
    const int ch_blk = tn * 64;
    const int orow0 = ty0 + px_r + 4 * wave;  // output row of pixel tile 0 (mt adds 2)
    const uint32_t m0 = (uint32_t)((n_img * a.ho + orow0) * a.wo + col);
    const uint32_t o0 = m0 * (uint32_t)a.cout + (uint32_t)ch_blk;
    const uint32_t mt_stride = (uint32_t)(2 * a.wo * a.cout);
    const uint32_t tag = (*G.epoch << 8) | G.conv_tag;
    float* red = (float*)(smem + RED16_OFF);
    // Phase A: (sum, sum of squares) of y = acc + bias per 8-channel octet over this wave's 64 pixels (pixels beyond a ragged plane
    // masked by a 0 / 1 factor), reduce-scatter butterfly over the lanes, 16 values per wave into LDS
    {
        float mval[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) mval[mt] = (lane_valid && orow0 + 2 * mt < a.ho) ? 1.0f : 0.0f;
        float w[16];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const float4 b4 = *(const float4*)(G.bias + ch_blk + nt * 32 + 8 * g + 4 * hi);
                float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
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
            red[wave * 16 + idx] = tot;
        }
    }
    // residual (the ResnetBlock's shortcut, added last): two 8-byte pieces per (nt, mt, g2) step; ALL 32 registers of it are requested
    // here, in front of the statistics exchange -- the loads (one group ahead they cost ~3 k cycles of exposed latency per tile) land
    // while the workgroup waits for the sample's other tiles
    const bool has_res = a.residual != nullptr;
    uint2 rq[4][4];
    if (has_res) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int nt = s >> 1, mt = s & 1;
            const bool st_ok = lane_valid && orow0 + 2 * mt < a.ho;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const size_t e0 = (size_t)(o0 + mt * mt_stride + nt * 32 + 16 * g2 + 4 * hi);
                rq[s][2 * g2] = st_ok ? *(const uint2*)(a.residual + e0) : make_uint2(0, 0);
                rq[s][2 * g2 + 1] = st_ok ? *(const uint2*)(a.residual + e0 + 8) : make_uint2(0, 0);
            }
        }
    }
    TL16(3)
    __syncthreads();  // the four waves' statistics are in LDS (and every wave has left the K loop)
    // Phase B: wave 0 adds the waves' values in wave order, publishes the workgroup's slot (16 granules), sweeps the sample's slots and
    // parks (A, C) of the block's 64 channels in LDS
    float* cfA = (float*)(smem + COEF16_OFF);
    float* cfC = cfA + 64;
    // dropout keep bits of this lane's 64 elements (engine generator: one hash per pair of consecutive channels, 16-bit thresholds --
    // the stream of act_drop_fixed), bit ((nt * 2 + mt) * 2 + g2) * 8 + t: computed inside the wait for the sample's statistics, which
    // they do not depend on (~3 k cycles of the epilogue of a dropout launch)
    uint32_t keep[2] = {0u, 0u};
    auto prehash = [&]() {
        if (a.drop.mode != 1) return;
        const uint32_t th = a.drop.thresh16;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int nt = s >> 1, mt = s & 1;
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const uint32_t e0 = o0 + mt * mt_stride + nt * 32 + 16 * g2 + 4 * hi;
                uint32_t bits = 0u;
#pragma unroll
                for (int h2 = 0; h2 < 2; ++h2)  // channels e0 + {0..3} and e0 + 8 + {0..3}
#pragma unroll
                    for (int p = 0; p < 2; ++p) {
                        const uint32_t w = rng_pair_word(((e0 + 8 * h2 - row0) >> 1) + p, key);
                        bits |= ((w & 0xffffu) < th ? 1u : 0u) << (4 * h2 + 2 * p);
                        bits |= ((w >> 16) < th ? 1u : 0u) << (4 * h2 + 2 * p + 1);
                    }
                keep[s >> 1] |= bits << (((s & 1) * 2 + g2) * 8);
            }
        }
    };
    if (wave != 0) prehash();
    if (wave == 0) {
        if (lane < 16) {
            const float tot = ((red[lane] + red[16 + lane]) + red[32 + lane]) + red[48 + lane];
            gn_store_granule(G.gran + (((size_t)n_img * G.max_slots + t_in) * (a.cout >> 3) + tn * 8) * 2 + lane, tag, tot);
        }
        prehash();  // (after the publication -- the other workgroups wait for it -- and before the sweep: the granules arrive meanwhile)
        const int cpg = a.cout / G.groups;
        const float2 mr = gn_fuse_sweep<16>(G.gran + ((size_t)n_img * G.max_slots * (a.cout >> 3) + tn * 8) * 2, (a.cout >> 3) * 2, G.slots,
                                            tag ^ G.test_tag_xor, cpg, 1.0 / ((double)a.ho * a.wo * cpg), G.err, lane, G.timeout_ticks);
        const float2 ac = gn_fuse_coef(G, ch_blk + lane, a.coef_div > 1 ? n_img / a.coef_div : n_img, mr);
        cfA[lane] = ac.x;
        cfC[lane] = ac.y;
        TL16(6)
    }
    __syncthreads();
    TL16(4)
    // Phase C: y * A + C -> SiLU -> dropout -> (+ residual) -> 16-bit; groups 2 g2 and 2 g2 + 1 are packed and exchanged between lanes
    // l and l + 32 (v_permlane32_swap), after which every lane owns 8 consecutive channels = one 16-byte store
    auto fused = [&](auto mode_c) {
        constexpr int MODE = decltype(mode_c)::value;
        // 32-channel half outermost, pixel tiles, then the two 16-channel groups of the half: the two 32-byte pieces of a pixel's 64-byte
        // half block are stored back to back (they leave the L2 as one 64-byte write)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int s = nt * 2 + mt;
                const bool st_ok = lane_valid && orow0 + 2 * mt < a.ho;
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    // (A, C) of this lane's 8 channels of the 16-channel group, from LDS per use: 16 live registers instead of 64 beside
                    // the accumulators and the prefetched residual
                    const int cgl = nt * 32 + 16 * g2 + 4 * hi;
                    const float4 a0 = *(const float4*)(cfA + cgl), a1 = *(const float4*)(cfA + cgl + 8);
                    const float4 c0 = *(const float4*)(cfC + cgl), c1 = *(const float4*)(cfC + cgl + 8);
                    const float ca[8] = {a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
                    const float cc[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
                    const int cg0 = nt * 32 + 16 * g2;
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = act_fixed<ACT_SILU>(fmaf(acc[nt][mt][8 * g2 + t], ca[t], cc[t]));
                    if constexpr (MODE == 1) {
                        const uint32_t kb = keep[nt] >> ((mt * 2 + g2) * 8);
#pragma unroll
                        for (int t = 0; t < 8; ++t) v[t] = ((kb >> t) & 1u) ? v[t] * a.drop.scale : 0.0f;
                    }
                    if (has_res) {
                        const uint2 r0 = rq[s][2 * g2], r1 = rq[s][2 * g2 + 1];
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
    };
    if (a.drop.mode == 1) fused(std::integral_constant<int, 1>{});
    else fused(std::integral_constant<int, 0>{});
    TL16(5)
#undef HKEY16
#endif
}

// plain 3 x 3 / stride 1 / pad 1, channels in whole 64-blocks, fragments of pack_halo3_frag64 in ConvArgs::wpk_up_frag, the fused
// GroupNorm requested
bool conv_gn16_supported(const ConvArgs& a) {
    if (a.gnf.gran == nullptr || a.up2x || a.up_nearest || a.wpk_up_frag == nullptr || a.out_el16 == nullptr || a.out_f32 != nullptr) return false;
    if (a.kh != 3 || a.kw != 3 || a.stride != 1 || a.pad != 1 || a.pix_pitch0 != 0 || a.ho != a.h || a.wo != a.w || a.h < 1 || a.w < 1) return false;
    // two sources of ANY 64-multiples (the up path's cat([x, skip]): 256 + 128, 128 + 64, 64 + 64); conv_up_halo_kernel<5> needs c1 == c0
    if (!(a.c0 > 0 && a.c0 % 64 == 0 && a.c1 % 64 == 0 && a.cout % 64 == 0)) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix < (1u << 28) && npix * (size_t)std::max(a.c0, a.c1) * 2 < 0x7F000000ull && (size_t)a.cout * 16 * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

int conv_gn16_slots(int h, int w) { return ((w + T16 - 1) / T16) * ((h + T16 - 1) / T16); }

hipError_t launch_conv_gn16(const ConvArgs& a, hipStream_t stream) {
    const int tiles_x = (a.w + T16 - 1) / T16, tiles_per_img = tiles_x * ((a.h + T16 - 1) / T16);
    const int tiles_m = a.n * tiles_per_img, tiles_n = a.cout / 64;
    dyf_form_note("conv_gn16_kernel+gn_fused", a.n);
    hipLaunchKernelGGL(conv_gn16_kernel, dim3(tiles_m * tiles_n), dim3(256), LDS16_TOTAL, stream, a, tiles_x, tiles_per_img, tiles_m, tiles_n);
#ifdef HALO_EXP_TIMELINE
    if (const char* tl = dyf_form("DYF_TIMELINE_DUMP")) {  // "path:N": the stamps of the N-th launch of the process (eager launches only)
        static int count = 0;
        const char* colon = strrchr(tl, ':');
        if (colon && ++count == atoi(colon + 1)) {
            (void)hipStreamSynchronize(stream);
            std::vector<unsigned long long> h((size_t)1 << 18);
            (void)hipMemcpyFromSymbol(h.data(), HIP_SYMBOL(g_gn16_tl), h.size() * 8);
            if (FILE* f = fopen(std::string(tl, colon - tl).c_str(), "wb")) {
                const int hdr[4] = {tiles_m * tiles_n, a.n, a.residual != nullptr, a.drop.mode};
                fwrite(hdr, sizeof(int), 4, f);
                fwrite(h.data(), 8, (size_t)std::min(tiles_m * tiles_n, 8192) * 32, f);
                fclose(f);
            }
        }
    }
#endif
    return hipGetLastError();
}
