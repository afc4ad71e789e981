// enc0 on the fused stem (gfx950): the first encoder block's 4x4 / stride-2 conv over the zero-bordered 16-channel stem16
// tensor (kernels.hip stem16_kernel; init_conv is composed into the weights, engine.hip compose_stem_enc0), K = 16 taps x 16
// channels = 256, cout = 64 or 128.  Reference: unet_simple.py:184-195 (cat -> Upsample -> init_conv) + :121,166 (first UNetBlock).
//
// The layer is HBM-bound: per output pixel 128 B of unique input and 2 * cout B of output against 2 * 256 * cout flop
// (170 flop/byte at cout = 128, under the 312 of the machine).  The general implicit-GEMM kernel (conv_igemm2.hip) spends its time
// in the prologue / epilogue of a K loop of only four 64-wide steps: 447 us for 1.0 GB at NB = 160 (2.25 TB/s).  Here
//   * the pixel operand never touches LDS: with the channel-contiguous 16-channel pixels, the MFMA fragment of k-step s = (kh, kw)
//     is ONE 16-byte global load per lane (lane = output pixel, hi = channel half); the 16 loads of a tile are issued a whole tile
//     ahead of their use, neighbouring taps / rows hit the L1 / L2;
//   * the weights (cout x 256 x 2 B <= 64 KB) stay in LDS as MFMA A fragments for the lifetime of the workgroup, which is
//     persistent: one 8-wave workgroup per CU walks a contiguous range of 32-pixel row segments (a wave keeps its column segment and moves
//     down the rows, so half of its input rows were fetched by its previous tile);
//   * operands are swapped (D^T = W X^T) so the epilogue runs from the accumulators exactly as in conv_igemm2.hip: lane (pixel, hi)
//     holds channels 32 nt + 8 g + 4 hi + {0..3}, exchanged pairwise (v_permlane32_swap) into 16-byte channel rows;
//   * a tile's output (32 pixels x cout channels) is one contiguous run of memory: the 16-byte rows go through a per-wave LDS
//     staging tile and leave as whole 1 KB wave stores: 327 -> 260 us at 160 rows against storing the 32-byte pieces of the
//     128-byte lines straight from the registers, ~100 instructions apart (8 times the store instructions' address work in the
//     texture path; the WRITE_SIZE counter reads 1.42-1.44x the algorithmic bytes for BOTH forms and 1.00x for conv_igemm2, so
//     it is not what separated them).
#include "conv.h"

#include <type_traits>
typedef __attribute__((ext_vector_type(16))) float f32x16;

namespace {
constexpr int E0_KSTEPS = 16;
constexpr int E0_WAVES = 8, E0_THREADS = E0_WAVES * 64;  // one workgroup per CU: the weights are fetched once per CU
constexpr int e0_lds_bytes(int nt) { return E0_KSTEPS * nt * 1024 + E0_WAVES * 2 * nt * 32 * 4 + E0_WAVES * 32 * (nt * 64 + 16); }
}

template <int NT>  // cout / 32
__global__ __launch_bounds__(512) void conv_enc0_stem_kernel(ConvArgs a, const el16_t* __restrict__ wfrag, int tiles, int tiles_per_wg) {
#if defined(__gfx950__)
    extern __shared__ __attribute__((aligned(16))) unsigned char e0_smem[];
    uint4* wl = (uint4*)e0_smem;  // [16 k-steps][NT][64 lanes]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
    for (int i = tid; i < E0_KSTEPS * NT * 64; i += E0_THREADS) wl[i] = ((const uint4*)wfrag)[i];
    __syncthreads();
    // Epilogue coefficients live in a per-wave LDS row, refreshed when the wave's tile moves to another coefficient row (sample):
    // loaded from global memory inside the epilogue they would make every (nt, g2) step wait for vmcnt(0), i.e. for the
    // acknowledgement of the stores of the step before (gfx9 counts stores in vmcnt) -- 8 exposed store round trips per tile.
    float* coefl = (float*)(e0_smem + E0_KSTEPS * NT * 1024) + wave * (2 * NT * 32);  // [a | c][cout]
    int coef_row = -1;
    // output staging tile of this wave: [32 pixels][cout * 2 B + 16 B pad] (the pad spreads the pixels' rows over the banks)
    constexpr int OROW = NT * 64 + 16;
    unsigned char* ostage = e0_smem + E0_KSTEPS * NT * 1024 + E0_WAVES * 2 * NT * 32 * 4 + wave * (32 * OROW);
    const int segs = a.wo >> 5;               // 32-pixel segments per output row
    const int plane = a.ho * a.wo;
    const int t0 = blockIdx.x * tiles_per_wg, t1 = min(t0 + tiles_per_wg, tiles);
    const int cout = NT * 32;

    // the 16 pixel fragments of tile t: k-step s = kh * 4 + kw reads padded input pixel (2 oy + kh, 2 ox + kw), channels 8 hi ..
    auto load = [&](int t, uint4 (&xf)[E0_KSTEPS]) {
        const int row = t / segs, seg = t - row * segs;     // row = n * ho + oy
        const int n = row / a.ho, oy = row - n * a.ho;
        const el16_t* p = a.src0 + ((size_t)(n * a.h + 2 * oy) * a.w + 2 * (seg * 32 + l31)) * 16 + 8 * hi;
#pragma unroll
        for (int kh = 0; kh < 4; ++kh)
#pragma unroll
            for (int kw = 0; kw < 4; ++kw) xf[kh * 4 + kw] = *(const uint4*)(p + ((size_t)kh * a.w + kw) * 16);
        __builtin_amdgcn_sched_barrier(0);  // issue them HERE, a tile ahead: sunk to their uses (the scheduler's choice) every k-step waits for its own load
    };

    auto tile = [&](int t, const uint4 (&xf)[E0_KSTEPS], auto act_c, auto mode_c) {
        constexpr int ACT = decltype(act_c)::value, MODE = decltype(mode_c)::value;
        f32x16 acc[NT];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[nt][r] = 0.0f;
#pragma unroll
        for (int s = 0; s < E0_KSTEPS; ++s) {
            int wlane = lane;
            asm volatile("" : "+v"(wlane));  // keep the (loop-invariant) weight reads inside the tile loop: hoisted they are 256 registers
            const el16x8_t xb = __builtin_bit_cast(el16x8_t, xf[s]);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
                acc[nt] = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, wl[(s * NT + nt) * 64 + wlane]), xb, acc[nt], 0, 0, 0);
        }
        // ---- epilogue (the scheme of conv_igemm2.hip)
        const int m = t * 32 + l31;             // output pixel: tiles are ordered (n, oy, segment)
        const int n_img = m / plane;
        const uint32_t ob = (uint32_t)m * (uint32_t)cout;
        const RngKey key = drop_row_key(a.drop, n_img);
        const uint32_t row0 = (uint32_t)n_img * (uint32_t)(plane * cout);
        const int crow = __builtin_amdgcn_readfirstlane(a.coef_div > 1 ? n_img / a.coef_div : n_img);  // one sample per tile
        if (crow != coef_row) {
            coef_row = crow;
            const float ps = drop_prescale<ACT, MODE>(a.drop);  // dropout scale folded into the affine, once per row
            if (lane < NT * 8) {
                float4 ca = *(const float4*)(a.coef_a + (size_t)crow * a.coef_stride + lane * 4);
                float4 cc = *(const float4*)(a.coef_c + (size_t)crow * a.coef_stride + lane * 4);
                ca.x *= ps; ca.y *= ps; ca.z *= ps; ca.w *= ps;
                cc.x *= ps; cc.y *= ps; cc.z *= ps; cc.w *= ps;
                *(float4*)(coefl + lane * 4) = ca;
                *(float4*)(coefl + NT * 32 + lane * 4) = cc;
            }
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            uint4 o2[2];
#pragma unroll
            for (int g2 = 0; g2 < 2; ++g2) {
                const int cg0 = nt * 32 + 16 * g2;
                const float4 ca0 = *(const float4*)(coefl + cg0 + 4 * hi), ca1 = *(const float4*)(coefl + cg0 + 4 * hi + 8);
                const float4 cc0 = *(const float4*)(coefl + NT * 32 + cg0 + 4 * hi), cc1 = *(const float4*)(coefl + NT * 32 + cg0 + 4 * hi + 8);
                const float ca[8] = {ca0.x, ca0.y, ca0.z, ca0.w, ca1.x, ca1.y, ca1.z, ca1.w};
                const float cc[8] = {cc0.x, cc0.y, cc0.z, cc0.w, cc1.x, cc1.y, cc1.z, cc1.w};
                const uint32_t e0 = ob + cg0 + 4 * hi;
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = fmaf(acc[nt][8 * g2 + i], ca[i], cc[i]);
                act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                const uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                const uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                o2[g2].x = s0[0]; o2[g2].y = s1[0]; o2[g2].z = s0[1]; o2[g2].w = s1[1];
            }
            *(uint4*)(ostage + l31 * OROW + (nt * 32 + 8 * hi) * 2) = o2[0];
            *(uint4*)(ostage + l31 * OROW + (nt * 32 + 16 + 8 * hi) * 2) = o2[1];
        }
        // the tile is cout * 64 contiguous bytes of the output: store instruction k writes bytes [1024 k, 1024 k + 1024)
        el16_t* tbase = a.out_el16 + (size_t)t * 32 * cout;
#pragma unroll
        for (int k = 0; k < NT * 2; ++k) {
            const int off = k * 1024 + lane * 16;                 // byte offset inside the tile
            const int px = off / (NT * 64), within = off - px * (NT * 64);
            const uint4 ov = *(const uint4*)(ostage + px * OROW + within);
#ifdef DYF_NT_STORES  // experiment: non-temporal stores
            typedef __attribute__((ext_vector_type(4))) unsigned nt_u32x4;
            __builtin_nontemporal_store(__builtin_bit_cast(nt_u32x4, ov), (nt_u32x4*)((unsigned char*)tbase + off));
#else
            *(uint4*)((unsigned char*)tbase + off) = ov;
#endif
        }
    };

    auto run = [&](auto act_c, auto mode_c) {
        // Every wave owns an even number of tiles (launcher: tiles per workgroup and the tile count are multiples of 16), and both
        // prefetches are unconditional (the one past the wave's last tile re-reads a valid tile): no branch joins between a
        // load and its use, so the compiler's s_waitcnt vmcnt counts stay exact and the loads really fly a tile ahead.  (With
        // `if (more) load(...)` the join made every k-step wait for the NEWEST 16 loads.)
        uint4 xa[E0_KSTEPS], xb[E0_KSTEPS];
        int t = t0 + wave;
        if (t >= t1) return;
        load(t, xa);
        for (; t < t1; t += 2 * E0_WAVES) {  // two tiles per turn: the fragment sets swap roles without register moves
            load(t + E0_WAVES, xb);
            tile(t, xa, act_c, mode_c);
            load(t + 2 * E0_WAVES < t1 ? t + 2 * E0_WAVES : t, xa);
            tile(t + E0_WAVES, xb, act_c, mode_c);
        }
    };
    auto by_mode = [&](auto act_c) {
        if (a.drop.mode == 0) run(act_c, std::integral_constant<int, 0>{});
        else if (a.drop.mode == 1) run(act_c, std::integral_constant<int, 1>{});
        else run(act_c, std::integral_constant<int, 2>{});
    };
    if (a.act == ACT_RELU) by_mode(std::integral_constant<int, ACT_RELU>{});
    else if (a.act == ACT_LEAKY) by_mode(std::integral_constant<int, ACT_LEAKY>{});
    else if (a.act == ACT_SILU) by_mode(std::integral_constant<int, ACT_SILU>{});
    else by_mode(std::integral_constant<int, ACT_NONE>{});
#endif
}

// wpk [cout][4 (kh)][64 = kw * 16 + c] (the composed weights of engine.hip, K index k = kh * 64 + kw * 16 + c) -> A fragments
// [k-step s][nt][lane][8]: lane (l31, hi) holds channel 32 nt + l31, k = 16 s + 8 hi + {0..7}
void pack_enc0_stem_frag(const el16_t* wpk, int cout, el16_t* out) {
    const int nt_n = cout / 32;
    for (int s = 0; s < E0_KSTEPS; ++s)
        for (int nt = 0; nt < nt_n; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int j = 0; j < 8; ++j)
                    out[(((size_t)s * nt_n + nt) * 64 + lane) * 8 + j] = wpk[(size_t)(32 * nt + (lane & 31)) * 256 + 16 * s + 8 * (lane >> 5) + j];
}

// the fused-stem view of enc0 (engine.hip fused_enc0_args): 16-channel pixels declared as 64-channel ones, kh = 4, kw = 1
bool conv_enc0_stem_supported(const ConvArgs& a) {
    const bool on = !(dyf_form("DYF_ENC0_STEM") && atoi(dyf_form("DYF_ENC0_STEM")) == 0);
    if (!on || a.pix_pitch0 != 16 || a.c0 != 64 || a.c1 != 0 || a.kh != 4 || a.kw != 1 || a.stride != 2 || a.pad != 0) return false;
    if (a.up2x || a.residual || a.out_f32 || !a.out_el16 || (a.cout != 64 && a.cout != 128)) return false;
    if (a.wo % 32 != 0 || (a.ho * (a.wo / 32)) % 16 != 0 || a.h != 2 * a.ho + 2 || a.w != 2 * a.wo + 2) return false;
    const long long nsel = a.n_sel > 0 ? a.n_sel : a.n;
    if (nsel * a.ho * (a.wo / 32) < 4096) return false;  // persistent form: needs >= 16 tiles for each of 256 workgroups
    return (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t conv_enc0_stem_init() {
    hipError_t e = hipFuncSetAttribute((const void*)conv_enc0_stem_kernel<4>, hipFuncAttributeMaxDynamicSharedMemorySize, e0_lds_bytes(4));
    if (e == hipSuccess) e = hipFuncSetAttribute((const void*)conv_enc0_stem_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, e0_lds_bytes(2));
    return e;
}

hipError_t launch_conv_enc0_stem(const ConvArgs& a, const el16_t* wfrag, hipStream_t stream) {
    const int tiles = a.n * a.ho * (a.wo / 32);
    int nwg = std::min(256, (tiles + 15) / 16);
    int per = (tiles + nwg - 1) / nwg;
    per = (per + 15) / 16 * 16;  // a wave keeps its column segment (wo = 128: 4 segments) and its fragment-set parity
    nwg = (tiles + per - 1) / per;
    dyf_form_note("conv_enc0_stem_kernel", a.n);
    if (a.cout == 128)
        hipLaunchKernelGGL(conv_enc0_stem_kernel<4>, dim3(nwg), dim3(E0_THREADS), e0_lds_bytes(4), stream, a, wfrag, tiles, per);
    else
        hipLaunchKernelGGL(conv_enc0_stem_kernel<2>, dim3(nwg), dim3(E0_THREADS), e0_lds_bytes(2), stream, a, wfrag, tiles, per);
    return hipGetLastError();
}
