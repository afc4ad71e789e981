// Small-M convolutions (gfx950): the UNet bottleneck -- enc4, enc5 (2x2 / s2), dec0, dec1 (1x1) of src/models/unet_simple.py:119-139,
// 512 -> 512 channels on 4^2 .. 16^2 planes -- and, at one to a few rows, the 4x4 / s2 and 3x3 layers around it (enc1-enc3, dec2):
// the batch sizes one GPU sees when an ensemble is sharded over a node.  There the 128 x 128 tiles of conv_igemm_kernel give 8 .. 40 workgroups on a 256-CU chip,
// so the launcher split K over blockIdx.y and a second kernel (conv_splitk_finish_kernel) added the partials and ran the epilogue:
// 13-20 us + 5-8 us per layer, five layers per forward, a quarter of a 10-row forward (profiles/r04a_bench_nb10_kernel_stats.txt).
//
// Here a workgroup owns a 32-pixel x 32-channel output tile and splits K over its SIXTEEN WAVES (each walks a sixteenth of the
// (tap, 64-channel chunk) steps), which are added through LDS in wave order -- one launch, no partials in HBM, the summation order
// of an output element fixed whatever the batch:
//   * 16 column blocks x ceil(M / 32) pixel tiles: 32 workgroups at one row of enc4, 640 at twenty;
//   * both MFMA operands come straight from global memory, no LDS staging and no barrier in the K loop: the weights in the fragment
//     order of conv_igemm2_kernel (pack_conv_frag: one wave-wide 16-byte load = one 32-channel x 16-k fragment, 1 KB contiguous), the
//     pixel operand as 16 bytes per lane from the lane's own input pixel (a (tap, chunk) step is one contiguous 128-byte piece of one
//     input pixel; taps in the zero padding get an out-of-range buffer offset = zeros); a wave's loads are all in flight together;
//   * operands swapped (D^T = W X^T) as in every conv kernel here: an accumulator lane holds 4 x 4 consecutive channels of one pixel;
//     after the LDS sum wave w finishes register group w (4 channels per lane): affine (conv bias + BatchNorm + FiLM) -> activation ->
//     dropout (+ residual) -> 16-bit (or raw fp32 for the GroupNorm block), 8-byte stores.
#include "conv.h"

#include <cstdlib>
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) unsigned u32x4;

namespace {
constexpr int SK_DEPTH = 8;   // k16 sub-steps of operand loads in flight per wave (K <= 2048: a wave's whole share)
}

// The small layers are LATENCY-bound, not bandwidth-bound: a layer's weights (0.5 - 2 MB) are re-fetched from the Infinity Cache
// every forward (~2 us per dependent round trip) and the first form of this kernel -- four waves per tile walking K / 4 each with
// 8 loads in flight -- needed four round trips (10-18 us per launch, no better than split-K + finish).  Sixteen waves per tile
// each own K / 16 = at most 8 sub-steps at K = 2048 and issue ALL their loads at once: one round trip.
// NW = 16 waves per tile (above), or 8 once the sixteen-wave workgroups would not fit the chip side by side (two per CU by wave slots:
// 512): enc4 / the low-res dec1 conv at 20 rows are 640 tiles -- 1.25 rounds of 7-8 us each, 23 us per launch in
// profiles/r05c_bench_nb10 -- while eight-wave workgroups (32 KB of LDS, four per CU) all run at once, each wave with twice the
// sub-steps (two batches of loads at K = 2048).
template <int NW>
__global__ __launch_bounds__(NW * 64) void conv_skinny_kernel_t(ConvArgs a, int M, int tiles_m) {
#if defined(__HIP_DEVICE_COMPILE__)
    constexpr int SK_WAVES = NW;
    __shared__ __attribute__((aligned(16))) float red[SK_WAVES][4][64][4];  // [wave][register group][lane][4 registers]: 64 KB at 16 waves
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hi = lane >> 5;
    const int tm = blockIdx.x % tiles_m, cb32 = blockIdx.x / tiles_m;  // pixel tile, 32-channel block
    const int taps = a.kh * a.kw, cpt0 = a.c0 >> 6, cpt = (a.c0 + a.c1) >> 6, nk = taps * cpt;  // chunks of src0, then of src1 (channel concat)
    const int plane = a.ho * a.wo;
    // this lane's output pixel (its operand row) and input window origin
    const int p = tm * 32 + l31;
    const bool valid = p < M;
    const int pc = valid ? p : M - 1;
    const int n_img = pc / plane, rem = pc - n_img * plane, oy = rem / a.wo, ox = rem - oy * a.wo;
    const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;  // input pixel of tap (0, 0): may lie in the zero padding
    const size_t npix = (size_t)a.n * a.h * a.w;
    const auto rsrc_x = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * a.c0 * 2), 0x00020000);
    const auto rsrc_x1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0, (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_w = __builtin_amdgcn_make_buffer_rsrc((void*)a.wpk_frag, 0, (int)(unsigned)((size_t)a.cout * nk * 128), 0x00020000);
    // weight fragments (pack_conv_frag, 128-channel column blocks): [tn][K step = chunk * taps + tap][wn][ks][half][lane] x 16 B
    const int tn = cb32 >> 2, wn = (cb32 >> 1) & 1, half = cb32 & 1;
    const unsigned w_lane = (unsigned)lane * 16u + (unsigned)half * 1024u + (unsigned)wn * 8192u;
    const unsigned w_step = 16384u;  // bytes per K step of a 128-channel column block
    const unsigned w_base = (unsigned)(tn * nk) * w_step;
    const unsigned x_lane = (unsigned)hi * 16u;

    // this wave's share of the k16 sub-steps g = step * 4 + ks
    const int G = nk * 4;
    const int g0 = G * wave / SK_WAVES, g1 = G * (wave + 1) / SK_WAVES;
    auto load = [&](int g, u32x4& wf, u32x4& xf) {
        const int step = g >> 2, ks = g & 3;
        const int chunk = step / taps, tap = step - chunk * taps;
        const int dy = tap / a.kw, dx = tap - dy * a.kw;
        wf = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, w_lane + (unsigned)ks * 2048u, w_base + (unsigned)step * w_step, 0);
        // a tap in the zero padding: an out-of-range offset, which the buffer bounds check answers with zeros
        const int iy = iy0 + dy, ix = ix0 + dx;
        const bool in = (unsigned)iy < (unsigned)a.h && (unsigned)ix < (unsigned)a.w;
        const bool first = chunk < cpt0;  // wave-uniform (a wave's sub-steps are contiguous, but may straddle the two sources)
        const int cs = first ? a.c0 : a.c1, cl = first ? chunk : chunk - cpt0;
        const unsigned xo = in ? (unsigned)((n_img * a.h + iy) * a.w + ix) * (unsigned)(cs * 2) + x_lane + (unsigned)(cl * 128 + ks * 32)
                               : 0xFFFFFFFFu;
        xf = first ? __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, xo, 0, 0) : __builtin_amdgcn_raw_buffer_load_b128(rsrc_x1, xo, 0, 0);
    };
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    u32x4 wq[SK_DEPTH], xq[SK_DEPTH];
#pragma unroll
    for (int d = 0; d < SK_DEPTH; ++d)
        if (g0 + d < g1) load(g0 + d, wq[d], xq[d]);
    for (int g = g0; g < g1; g += SK_DEPTH) {
#pragma unroll
        for (int d = 0; d < SK_DEPTH; ++d) {
            if (g + d < g1) {
                const u32x4 wf = wq[d], xf = xq[d];
                if (g + d + SK_DEPTH < g1) load(g + d + SK_DEPTH, wq[d], xq[d]);
                acc = DYF_MFMA_32x32x16(__builtin_bit_cast(el16x8_t, wf), __builtin_bit_cast(el16x8_t, xf), acc, 0, 0, 0);
            }
        }
    }
    // ---- sum over the sixteen waves, in wave order: every wave parks its accumulators (four 16-byte pieces per lane), wave w < 4
    // finishes register group w (channels 8 w + 4 hi + {0..3} of the block)
#pragma unroll
    for (int g = 0; g < 4; ++g) *(float4*)&red[wave][g][lane][0] = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]);
    __syncthreads();
    if (wave >= 4 || !valid) return;
    const int g = wave;
    float4 s4 = *(const float4*)&red[0][g][lane][0];
#pragma unroll
    for (int i = 1; i < SK_WAVES; ++i) {
        const float4 t4 = *(const float4*)&red[i][g][lane][0];
        s4.x += t4.x; s4.y += t4.y; s4.z += t4.z; s4.w += t4.w;
    }
    float v[4] = {s4.x, s4.y, s4.z, s4.w};
    // ---- epilogue: 4 consecutive channels of this lane's pixel
    const int ch = cb32 * 32 + 8 * g + 4 * hi;
    const uint32_t e0 = (uint32_t)p * (uint32_t)a.cout + (uint32_t)ch;
    const uint32_t row0 = (uint32_t)n_img * (uint32_t)(plane * a.cout);
    const size_t ci = (size_t)(a.coef_div > 1 ? n_img / a.coef_div : n_img) * a.coef_stride + ch;
    const float4 ca = *(const float4*)(a.coef_a + ci), cc = *(const float4*)(a.coef_c + ci);
    v[0] = fmaf(v[0], ca.x, cc.x); v[1] = fmaf(v[1], ca.y, cc.y); v[2] = fmaf(v[2], ca.z, cc.z); v[3] = fmaf(v[3], ca.w, cc.w);
    act_drop<4>(v, e0, row0, a.act, a.drop, drop_row_key(a.drop, n_img));
    if (a.residual) {
        const uint2 rr = *(const uint2*)(a.residual + (size_t)e0);
        v[0] += el16_lo(rr.x); v[1] += el16_hi(rr.x); v[2] += el16_lo(rr.y); v[3] += el16_hi(rr.y);
    }
    if (a.out_f32) *(float4*)(a.out_f32 + (size_t)e0) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.out_el16) *(uint2*)(a.out_el16 + (size_t)e0) = make_uint2(pack_el16x2(v[0], v[1]), pack_el16x2(v[2], v[3]));
#endif
}

// shapes the kernel takes: stride == kernel (1 or 2), pad 0, one source, 128-channel column blocks (the fragment layout of
// pack_conv_frag), fragment copy registered
bool conv_skinny_supported(const ConvArgs& a) {
    if (a.up2x || a.wpk_frag == nullptr || a.pix_pitch0 != 0 || (a.c1 != 0 && a.src1 == nullptr)) return false;
    if (a.kh < 1 || a.kw < 1 || a.kh * a.kw > 16 || a.stride < 1 || a.pad < 0) return false;
    // 16 .. 128 k16 sub-steps: at least one per wave, at most SK_DEPTH (one memory round trip).  Deeper K (enc3: 256, dec2: 576
    // sub-steps) was measured SLOWER here than split-K over workgroups (NS at 4 / 7 rows: 3 250 / 4 560 against 3 910 / 5 130 fields/s)
    const int max_steps = dyf_form("DYF_SKINNY_MAX_KSTEPS") ? atoi(dyf_form("DYF_SKINNY_MAX_KSTEPS")) : 32;
    const int nk = a.kh * a.kw * ((a.c0 + a.c1) >> 6);
    if (a.c0 % 64 != 0 || a.c1 % 64 != 0 || a.cout % 128 != 0 || nk < 4 || nk > max_steps) return false;
    if (a.ho != (a.h + 2 * a.pad - a.kh) / a.stride + 1 || a.wo != (a.w + 2 * a.pad - a.kw) / a.stride + 1) return false;
    const size_t npix = (size_t)a.n * a.h * a.w;
    return npix * a.c0 * 2 < 0x7F000000ull && npix * (size_t)a.c1 * 2 < 0x7F000000ull && (size_t)a.cout * a.kh * a.kw * (a.c0 + a.c1) * 2 < 0x7F000000ull &&
           (size_t)a.n * a.ho * a.wo * a.cout < 0xFFFFFFF0ull;
}

hipError_t launch_conv_skinny(const ConvArgs& a, hipStream_t stream) {
    const int M = a.n * a.ho * a.wo, tiles_m = (M + 31) / 32;
    dyf_form_note("conv_skinny_kernel", a.n);
    const long long wgs = (long long)tiles_m * (a.cout / 32);
    // the form follows the tile count of ConvArgs::n_sel rows when the engine pins the forms (the K order of an output differs)
    const long long sel = a.n_sel > 0 ? ((long long)a.n_sel * a.ho * a.wo + 31) / 32 * (a.cout / 32) : wgs;
    const long long w8_from = dyf_form("DYF_SKINNY_W8_FROM") ? atoll(dyf_form("DYF_SKINNY_W8_FROM")) : 513;
    if (sel >= w8_from) {
        dyf_form_note("conv_skinny_kernel<8>", a.n);
        hipLaunchKernelGGL(conv_skinny_kernel_t<8>, dim3((unsigned)wgs), dim3(512), 0, stream, a, M, tiles_m);
    } else
        hipLaunchKernelGGL(conv_skinny_kernel_t<16>, dim3((unsigned)wgs), dim3(1024), 0, stream, a, M, tiles_m);
    return hipGetLastError();
}
