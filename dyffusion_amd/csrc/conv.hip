// Conv2d on NHWC bf16 for gfx950 (MI355X): implicit-GEMM on MFMA + a shape-agnostic direct kernel.
//
// Replaces the reference's nn.Conv2d -> BatchNorm2d -> FiLM -> (Leaky)ReLU -> Dropout ATen sequence
// (src/models/unet_simple.py:29-82) with one kernel per UNetBlock.
//
// Implicit GEMM:  Y[M = N*Ho*Wo pixels][Cout] = A[M][K = taps*Cin] * Wpk[Cout][K]^T
//   * K is walked one (tap, 64-channel chunk) per step: with NHWC activations every A row of a step is one contiguous
//     128-byte segment of one input pixel, so the A tile is a pure gather of 128-B rows.  Rows are fetched with
//     LDS-DMA (global_load_lds_dwordx4, 16 B/lane, 1 KiB per wave instruction); taps that fall into the zero padding
//     read a 128-B zero page instead.  Weights are pre-packed [Cout][tap][Cin] so a B row is the same kind of segment.
//   * LDS image per stage: A[BM][64] + B[BN][64] bf16, row = 128 B, 16-B chunk index XOR (row & 7): the DMA writes LDS
//     linearly, so the swizzle is applied to the per-lane SOURCE chunk and again on the ds_read_b128 fragment reads.
//   * 4 waves, each owns a 64x64 accumulator (2x2 MFMA 32x32x16 bf16 tiles, 64 fp32 regs); two stages, one barrier per
//     K step, next step's DMA in flight under the current step's 16 MFMAs.
//   * operands swapped (D^T = W X^T): an accumulator lane holds 4 consecutive channels of one pixel, so the epilogue --
//     per-(sample, channel) affine (conv bias + BatchNorm + FiLM folded) -> activation -> dropout (+ residual) -> bf16 --
//     runs straight out of the accumulators (v_permlane32_swap pairs lanes l / l+32 for 16-B stores): no LDS round trip.
//   * dispatch (launch_conv): fused-upsample convs and plain 3x3-s1 / 4x4-s2 convs with 256-channel blocks -> the halo
//     kernel (conv_up_halo.hip); other convs with cout % 128 == 0 and >= 384 tiles -> conv_igemm2_kernel (weights
//     streamed in fragment order); the rest -> this file's conv_igemm_kernel; channels % 64 != 0 -> conv_direct_kernel.
//   * blockIdx -> tile map is XCD-aware (block b runs on XCD b % 8): every XCD walks a contiguous range of tiles so
//     the 3x3 / 4x4 halo re-reads of neighbouring tiles hit that XCD's own L2.
#include "conv.h"

#include <type_traits>

#include <cstdlib>
#include <deque>
#include <map>
#include <mutex>
#include <string>

// ---- kernel-form log (common.h): form name + "@rows" -> launches noted since it was enabled
bool g_dyf_form_log_on = false;
static std::mutex g_form_mu;
static std::map<std::string, long long> g_form_log;
void dyf_form_note_slow(const char* form, long long rows) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    ++g_form_log[std::string(form) + "@" + std::to_string(rows)];
}
void dyf_form_log_enable(bool on) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    g_form_log.clear();
    g_dyf_form_log_on = on;
}
std::string dyf_form_log_text() {
    std::lock_guard<std::mutex> lk(g_form_mu);
    std::string t;
    for (auto& kv : g_form_log) t += kv.first + "=" + std::to_string(kv.second) + ";";
    return t;
}
// ---- kernel-form switches (common.h dyf_form): key -> value, set only through dyf_debug_set_form.  Values are interned and never
// freed (a pointer handed out by dyf_form stays valid for the life of the process).
int g_dyf_form_count = 0;
static std::map<std::string, const char*> g_form_values;
static std::deque<std::string> g_form_arena;
const char* dyf_form_slow(const char* key) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    auto it = g_form_values.find(key);
    return it == g_form_values.end() ? nullptr : it->second;
}
void dyf_form_set(const char* key, const char* value) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    if (!key) g_form_values.clear();
    else if (!value) g_form_values.erase(key);
    else {
        g_form_arena.emplace_back(value);
        g_form_values[key] = g_form_arena.back().c_str();
    }
    __atomic_store_n(&g_dyf_form_count, (int)g_form_values.size(), __ATOMIC_RELEASE);
}
std::string dyf_form_text() {
    std::lock_guard<std::mutex> lk(g_form_mu);
    std::string t;
    for (auto& kv : g_form_values) t += kv.first + "=" + kv.second + ";";
    return t;
}
// ---- named-kernel timing (common.h KernelProf): (start, stop, algorithmic bytes) of every launch of the armed name
const char* g_dyf_prof_name = nullptr;
static std::string g_prof_name_store;
struct ProfRec { hipEvent_t e0, e1; double bytes; };
static std::deque<ProfRec> g_prof_recs;
void dyf_prof_begin(hipStream_t st, double bytes) {
    std::lock_guard<std::mutex> lk(g_form_mu);  // (records of concurrent host threads interleave but stay whole; ONE armed name per process)
    ProfRec r{nullptr, nullptr, bytes};
    if (g_prof_recs.size() >= 16384 || hipEventCreate(&r.e0) != hipSuccess || hipEventCreate(&r.e1) != hipSuccess) return;
    (void)hipEventRecord(r.e0, st);
    g_prof_recs.push_back(r);
}
void dyf_prof_end(hipStream_t st) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    if (!g_prof_recs.empty()) (void)hipEventRecord(g_prof_recs.back().e1, st);
}
static void prof_arm_locked(const char* name) {
    for (auto& r : g_prof_recs) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    g_prof_recs.clear();
    g_prof_name_store = name ? name : "";
    g_dyf_prof_name = name ? g_prof_name_store.c_str() : nullptr;
}
void dyf_prof_arm(const char* name) {  // nullptr disarms; pending records are dropped
    std::lock_guard<std::mutex> lk(g_form_mu);
    prof_arm_locked(name);
}
// after the stream has been synchronised: total ms, total bytes, launches of the armed name; disarms
void dyf_prof_collect(double* total_ms, double* total_bytes, int* launches) {
    std::lock_guard<std::mutex> lk(g_form_mu);
    double ms = 0.0, by = 0.0;
    int n = 0;
    for (auto& r : g_prof_recs) {
        float t = 0.0f;
        if (hipEventElapsedTime(&t, r.e0, r.e1) == hipSuccess) { ms += t; by += r.bytes; ++n; }
    }
    *total_ms = ms; *total_bytes = by; *launches = n;
    prof_arm_locked(nullptr);
}
typedef __attribute__((ext_vector_type(16))) float f32x16;

#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))
#define GLB_PTR(p) ((const __attribute__((address_space(1))) void*)(p))

// ------------------------------------------------------------------------------------------------ epilogue math
struct EpiCtx {
    const float* coef_a;
    const float* coef_c;
    int coef_stride;
    int act;
    int cout;
    int howo;
};

// ------------------------------------------------------------------------------------------------ direct kernel
// One thread per (output pixel, output channel).  Correct for every shape; used for layers the MFMA path does not
// cover (channel counts that are not multiples of 64) and as the on-device cross-check of the MFMA kernel.
__global__ void conv_direct_kernel(ConvArgs a, long long total) {
    long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= total) return;
    const int co = (int)(idx % a.cout);
    const long long m = idx / a.cout;
    const int howo = a.ho * a.wo;
    const int n = (int)(m / howo);
    const int rem = (int)(m % howo);
    const int oy = rem / a.wo, ox = rem % a.wo;
    const int cin = a.c0 + a.c1;
    const el16_t* wrow = a.wpk + (size_t)co * a.kh * a.kw * cin;
    float acc = 0.0f;
    for (int ky = 0; ky < a.kh; ++ky) {
        const int iy = oy * a.stride - a.pad + ky;
        if ((unsigned)iy >= (unsigned)a.h) continue;
        for (int kx = 0; kx < a.kw; ++kx) {
            const int ix = ox * a.stride - a.pad + kx;
            if ((unsigned)ix >= (unsigned)a.w) continue;
            const size_t pix = ((size_t)n * a.h + iy) * a.w + ix;
            const el16_t* wt = wrow + (size_t)(ky * a.kw + kx) * cin;
            const el16_t* p0 = a.src0 + pix * (a.pix_pitch0 ? a.pix_pitch0 : a.c0);
            for (int c = 0; c < a.c0; ++c) acc = fmaf(el16_to_f32(p0[c]), el16_to_f32(wt[c]), acc);
            if (a.c1 > 0) {
                const el16_t* p1 = a.src1 + pix * a.c1;
                for (int c = 0; c < a.c1; ++c) acc = fmaf(el16_to_f32(p1[c]), el16_to_f32(wt[a.c0 + c]), acc);
            }
        }
    }
    const size_t ci = (size_t)(a.coef_div > 1 ? n / a.coef_div : n) * a.coef_stride + co;
    float v = fmaf(acc, a.coef_a[ci], a.coef_c[ci]);
    v = apply_act(v, a.act);
    const uint32_t e = (uint32_t)(m * a.cout + co);
    v = drop_apply(v, e, (uint32_t)n * (uint32_t)(a.ho * a.wo * a.cout), a.drop, drop_row_key(a.drop, n));
    if (a.residual) v += el16_to_f32(a.residual[(size_t)m * a.cout + co]);
    if (a.out_el16) a.out_el16[(size_t)m * a.cout + co] = f32_to_el16(v);
    if (a.out_f32) a.out_f32[(size_t)m * a.cout + co] = v;
}

// ------------------------------------------------------------------------------------------------ MFMA kernel
// Gather addressing: every A row a lane fetches is described by a 32-bit byte offset of its window origin
// (n, oy*stride-pad, ox*stride-pad) in each source plus a bit mask of the taps that fall inside the image.  Per K
// step the lane adds one wave-uniform (tap, chunk) byte offset and selects "out of range" for padded taps: the
// LDS-DMA is a raw buffer load (buffer_load_dwordx4 ... offen lds), whose bounds check returns zeros for offsets
// beyond num_records, so the zero padding costs no memory traffic and ~4 VALU instructions per row.
//
// UP = 1: nn.Upsample(scale_factor=2, bilinear) + Conv2d(3x3, pad 1) of a decoder block (unet_simple.py:40-52) fused
// by PHASE DECOMPOSITION.  Output pixel (2i+py, 2j+px) is a 3x3 stencil over the LOW-res pixels (i+a, j+b) with
// weights that are fixed linear combinations of the 3x3 kernel (host: pack_up2x_weights); replicate-clamping the tap
// coordinates reproduces the edge behaviour of align_corners=False.  What clamping cannot express is the conv's ZERO
// padding of the upsampled image (up[-1] = 0 while the clamped stencil yields in[0]); it is repaired by extra
// "correction taps" that only the first/last output row/column see: -w[ky=0,:] on the border row, -w[:,kx=0] on the
// border column and +w[0,0] on the corner (inclusion/exclusion).  Per phase the packed K axis therefore has 16 taps:
// 0-8 stencil, 9-11 row correction, 12-14 column correction, 15 corner; tiles that touch no border skip 9-15.
// The upsampled tensor (4x the bytes of its input) is never materialised and the FLOP count equals the direct conv's.
struct StepInfo {       // wave-uniform description of the K step being issued
    int dy, dx;         // tap displacement in input pixels
    unsigned need;      // UP: border flags a row must have for this tap to contribute
    unsigned tap_bit;   // !UP: bit of this tap in the row's validity mask
    int wslot;          // index of the tap on the packed-weight K axis
};

template <int BM, int BN, int WM, int WN, int UP>
__global__ __launch_bounds__(256, 2) void conv_igemm_kernel(ConvArgs a, int M, int tiles_m, int tiles_n) {
    static_assert(WM * WN == 4 && BM / WM == 64 && BN / WN == 64, "4 waves, 64x64 accumulator each");
#if defined(__HIP_DEVICE_COMPILE__)  // buffer-resource builtins only exist in the device pass
    constexpr int A_BYTES = BM * 128;
    constexpr int B_BYTES = BN * 128;
    constexpr int STAGE = A_BYTES + B_BYTES;
    constexpr int RA = BM / 32;  // A rows gathered per lane per K step
    constexpr int RB = BN / 32;
    constexpr int TH = BM / 16;  // 2-D tile: TH rows of 16 pixels
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    // XCD-aware tile id (bijective for any tile count): XCD x = bid % 8 takes a contiguous run of tiles.
    const int total = tiles_m * tiles_n;
    const int bid = blockIdx.x;
    const int xq = total >> 3, xr = total & 7, xcd = bid & 7;
    const int tile = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (bid >> 3);
    const int tn = tile % tiles_n;
    int tm = tile / tiles_n;
    int phase = 0;
    if (UP) {  // the four phases of one low-res tile are consecutive tiles: they share their input window in L2
        phase = tm & 3;
        tm >>= 2;
    }
    const int py = phase >> 1, px = phase & 1;

    const int cin = a.c0 + a.c1;
    const int cpt = cin >> 6;                 // 64-channel chunks per tap
    const int wtaps = UP ? 16 : a.kh * a.kw;  // taps on the packed-weight K axis
    // the plane the tile walks: output pixels, or (UP) low-res input pixels each owning one output pixel per phase
    const int ph = UP ? a.h : a.ho, pw = UP ? a.w : a.wo;
    const int plane = ph * pw;
    // 2-D tiling when the plane divides into TH x 16 tiles; otherwise BM consecutive raster pixels
    const bool tile2d = UP || ((pw % 16 == 0) && (ph % TH == 0));
    const int tiles_x = pw >> 4, tiles_per_img = tile2d ? tiles_x * (ph / TH) : 1;
    const int sub = lane >> 3;                // row inside an 8-row DMA group
    // LDS swizzle key of a tile row r is (r >> 1) & 7: a ds_read_b128 lane group covers 16 rows whose 128-B rows
    // alternate between the two halves of the 256-B bank row, so the key must change every SECOND row to spread the
    // group over all 16 16-B slots (conflict-free; key r & 7 is 2-way).  DMA rows are (j*4+wave)*8 + sub.
    const int gchunk = (lane & 7) ^ ((((wave & 1) << 2) | (sub >> 1)) & 7);  // swizzled SOURCE chunk of this lane's slot

    const size_t npix = (size_t)a.n * a.h * a.w;
    const int pitch0 = a.pix_pitch0 ? a.pix_pitch0 : a.c0;
    const auto rsrc_a0 = __builtin_amdgcn_make_buffer_rsrc((void*)a.src0, 0, (int)(unsigned)(npix * pitch0 * 2), 0x00020000);
    const auto rsrc_a1 = __builtin_amdgcn_make_buffer_rsrc((void*)(a.c1 ? a.src1 : a.src0), 0,
                                                           (int)(unsigned)(npix * (a.c1 ? a.c1 : a.c0) * 2), 0x00020000);
    const auto rsrc_b = __builtin_amdgcn_make_buffer_rsrc(
        (void*)a.wpk, 0, (int)(unsigned)((size_t)(UP ? 4 : 1) * a.cout * wtaps * cpt * 128), 0x00020000);

    // tile position on the plane (2-D tiles)
    int t_img = 0, t_y0 = 0, t_x0 = 0;
    if (tile2d) {
        t_img = tm / tiles_per_img;
        const int t = tm - t_img * tiles_per_img;
        t_y0 = (t / tiles_x) * TH;
        t_x0 = (t % tiles_x) * 16;
    }
    // UP: which correction-tap groups this tile needs (wave-uniform)
    const bool has_row = UP && (py == 0 ? t_y0 == 0 : t_y0 + TH == ph);
    const bool has_col = UP && (px == 0 ? t_x0 == 0 : t_x0 + 16 == pw);
    const int ntaps = UP ? 9 + (has_row ? 3 : 0) + (has_col ? 3 : 0) + (has_row && has_col ? 1 : 0) : a.kh * a.kw;
    const int nk = ntaps * cpt;               // K steps of this tile
    const unsigned need_row = py == 0 ? 1u : 2u, need_col = px == 0 ? 4u : 8u;

    // ---- per-lane gather descriptors for the A rows this lane fetches
    // a_mask: !UP tap validity bits; UP border flags (1 top, 2 bottom, 4 left, 8 right, 16 row exists)
    unsigned a_off0[RA], a_off1[RA], a_mask[RA];
#pragma unroll
    for (int j = 0; j < RA; ++j) {
        const int row = (j * 4 + wave) * 8 + sub;
        unsigned mask = 0;
        int pix = 0;
        if (tm * BM + row < M) {
            int n_img, oy, ox;
            if (tile2d) {  // 16-pixel-wide 2-D tile: the taps of one tile re-read a small window from L2
                n_img = t_img;
                oy = t_y0 + (row >> 4);
                ox = t_x0 + (row & 15);
            } else {
                const int m = tm * BM + row;
                n_img = m / plane;
                const int rem = m - n_img * plane;
                oy = rem / pw;
                ox = rem - oy * pw;
            }
            if (UP) {
                pix = (n_img * a.h + oy) * a.w + ox;
                mask = 16u | (oy == 0 ? 1u : 0u) | (oy == a.h - 1 ? 2u : 0u) | (ox == 0 ? 4u : 0u) | (ox == a.w - 1 ? 8u : 0u);
            } else {
                const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
                pix = (n_img * a.h + iy0) * a.w + ix0;  // may be "negative": only ever used together with a valid tap
                // taps inside the image: ky in [ylo, yhi), kx in [xlo, xhi) -> bit (ky*kw + kx)
                const int ylo = max(0, -iy0), yhi = min(a.kh, a.h - iy0);
                const int xlo = max(0, -ix0), xhi = min(a.kw, a.w - ix0);
                const unsigned ym = yhi > ylo ? (1u << yhi) - (1u << ylo) : 0u;
                const unsigned xm = xhi > xlo ? (1u << xhi) - (1u << xlo) : 0u;
                for (int ky = 0; ky < a.kh; ++ky) mask |= (((ym >> ky) & 1u) ? xm : 0u) << (ky * a.kw);
            }
        }
        a_mask[j] = mask;
        a_off0[j] = (unsigned)pix * (unsigned)(pitch0 * 2) + gchunk * 16;
        a_off1[j] = (unsigned)pix * (unsigned)(a.c1 * 2) + gchunk * 16;
    }
    // B rows: row r of the tile is output channel tn*BN + r of this phase's weight set; a K step is one 128-B segment
    unsigned b_off[RB];
#pragma unroll
    for (int j = 0; j < RB; ++j) {
        const int row = (j * 4 + wave) * 8 + sub;
        b_off[j] = (unsigned)(phase * a.cout + tn * BN + row) * (unsigned)(wtaps * cpt * 128) + gchunk * 16;
    }

    // issue-side K-step state (wave-uniform): tap counter (position in this tile's tap list) and chunk
    int is_tap = 0, is_chunk = 0;
    int tap_lo = 0;  // first tap position of the list being walked (9 while a UP tile runs its correction taps)

    auto step_info = [&](int tpos) {
        StepInfo si;
        if (UP) {
            int t = tpos;  // position -> tap id: 0-8 stencil, then the correction groups this tile has
            if (t >= 9) {
                t -= 9;
                if (has_row && t < 3) t += 9;
                else {
                    if (has_row) t -= 3;
                    if (has_col && t < 3) t += 12;
                    else t = 15;
                }
            }
            si.wslot = t;
            si.tap_bit = 0;
            if (t < 9) { si.dy = t / 3 - 1; si.dx = t % 3 - 1; si.need = 16u; }
            else if (t < 12) { si.dy = 0; si.dx = t - 10; si.need = 16u | need_row; }
            else if (t < 15) { si.dy = t - 13; si.dx = 0; si.need = 16u | need_col; }
            else { si.dy = 0; si.dx = 0; si.need = 16u | need_row | need_col; }
        } else {
            si.dy = tpos / a.kw;
            si.dx = tpos - si.dy * a.kw;
            si.wslot = tpos;
            si.tap_bit = 1u << tpos;
            si.need = 0;
        }
        return si;
    };

    // issue the LDS-DMA of K step (is_tap, is_chunk) into `stage`
    auto issue = [&](int stage) {
        char* As = smem + stage * STAGE;
        char* Bs = As + A_BYTES;
        const StepInfo si = step_info(is_tap);
        const int cb = is_chunk << 6;
        const bool second = cb >= a.c0;
        const int csrc = second ? a.c1 : pitch0;
        const int coff = second ? cb - a.c0 : cb;
        const unsigned dyoff = (unsigned)(si.dy * a.w * csrc * 2), dxoff = (unsigned)(si.dx * csrc * 2);
        const unsigned chunk_off = (unsigned)(coff * 2);
        // UP: a tap that would step outside the image is clamped back onto the border pixel (replicate)
        const unsigned kill_y = si.dy < 0 ? 1u : (si.dy > 0 ? 2u : 0u), kill_x = si.dx < 0 ? 4u : (si.dx > 0 ? 8u : 0u);
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const unsigned base = second ? a_off1[j] : a_off0[j];
            unsigned vo;
            if (UP) {
                const unsigned f = a_mask[j];
                const unsigned o = base + chunk_off + ((f & kill_y) ? 0u : dyoff) + ((f & kill_x) ? 0u : dxoff);
                vo = ((f & si.need) == si.need) ? o : 0xFFFFFFFFu;
            } else {
                vo = (a_mask[j] & si.tap_bit) ? base + chunk_off + dyoff + dxoff : 0xFFFFFFFFu;
            }
            if (second)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
            else
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(As + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0);
        }
#pragma unroll
        for (int j = 0; j < RB; ++j) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, LDS_PTR(Bs + (j * 4 + wave) * 1024), 16, b_off[j],
                                                     (si.wslot * cpt + is_chunk) * 128, 0, 0);
        }
        // K order: taps fastest, then the 64-channel chunk -- all taps of one chunk are consecutive, so a tile's input
        // window for that chunk (a few tens of KB) is fetched from HBM once and re-read from L2 by the other taps
        if (++is_tap == ntaps) {
            is_tap = tap_lo;
            ++is_chunk;
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    // fragment read offsets: row*128 + ((k16*2 + (lane>>5)) ^ ((row>>1) & 7)) * 16 ; fragment rows are 32-aligned + (lane&31)
    const int l31 = lane & 31, hi = lane >> 5, l7 = (lane >> 1) & 7;
    const int a_row_off = (wm * 64 + l31) * 128;
    const int b_row_off = (wn * 64 + l31) * 128;

    // One K step of MFMAs.  Fragment reads are software-pipelined one k16 sub-step ahead (two register sets), so only
    // the first ds_read latency of a K step is exposed instead of four.
    auto compute = [&](int stage) {
        const char* As = smem + stage * STAGE + a_row_off;
        const char* Bs = smem + stage * STAGE + A_BYTES + b_row_off;
        el16x8_t af[2][2], bfr[2][2];
        {
            const int coff = ((hi ^ l7) << 4);
#pragma unroll
            for (int i = 0; i < 2; ++i) af[0][i] = *(const el16x8_t*)(As + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < 2; ++j) bfr[0][j] = *(const el16x8_t*)(Bs + j * 32 * 128 + coff);
        }
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cs = ks & 1, ns = cs ^ 1;
            if (ks < 3) {
                const int coff = ((((ks + 1) * 2 + hi) ^ l7) << 4);
#pragma unroll
                for (int i = 0; i < 2; ++i) af[ns][i] = *(const el16x8_t*)(As + i * 32 * 128 + coff);
#pragma unroll
                for (int j = 0; j < 2; ++j) bfr[ns][j] = *(const el16x8_t*)(Bs + j * 32 * 128 + coff);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = DYF_MFMA_32x32x16(bfr[cs][j], af[cs][i], acc[i][j], 0, 0, 0);  // D^T = W X^T
        }
    };

    // UP fast path (both sources have the same channel count): the 9 stencil taps are unrolled and each row's gather
    // offset is base + yoff[a] + xoff[b] with per-row, per-direction byte offsets precomputed once per tile (clamped to
    // the border = replicate); one v_add3 per row and K step instead of ~11 VALU.  Rows past M carry an out-of-range
    // base.  The correction taps of border tiles run afterwards through the general `issue` path.
    const bool up_fast = UP && (a.c1 == 0 || a.c1 == a.c0) && npix * (size_t)a.c0 * 2 < 0x7F000000ull;
    if (UP && up_fast) {
        const unsigned rowb = (unsigned)(a.w * a.c0 * 2), pixb = (unsigned)(a.c0 * 2);
        unsigned ybase[RA], y_m[RA], y_p[RA], x_m[RA], x_p[RA];
#pragma unroll
        for (int j = 0; j < RA; ++j) {
            const unsigned f = a_mask[j];
            ybase[j] = (f & 16u) ? a_off0[j] : 0x80000000u;
            y_m[j] = (f & 1u) ? 0u : 0u - rowb;
            y_p[j] = (f & 2u) ? 0u : rowb;
            x_m[j] = (f & 4u) ? 0u : 0u - pixb;
            x_p[j] = (f & 8u) ? 0u : pixb;
        }
        const int ncorr = ntaps - 9;
        int chunk = 0;
        unsigned cb_off = 0;  // byte offset of the current chunk inside its source
        bool second = false;

#define ISSUE_STENCIL(T, STAGE_)                                                                                  \
    {                                                                                                             \
        char* As_ = smem + (STAGE_) * STAGE;                                                                      \
        char* Bs_ = As_ + A_BYTES;                                                                                \
        _Pragma("unroll") for (int j = 0; j < RA; ++j) {                                                          \
            const unsigned vo = ybase[j] + cb_off + ((T) / 3 == 0 ? y_m[j] : ((T) / 3 == 2 ? y_p[j] : 0u)) +      \
                                ((T) % 3 == 0 ? x_m[j] : ((T) % 3 == 2 ? x_p[j] : 0u));                           \
            if (second)                                                                                           \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a1, LDS_PTR(As_ + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0); \
            else                                                                                                  \
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_a0, LDS_PTR(As_ + (j * 4 + wave) * 1024), 16, vo, 0, 0, 0); \
        }                                                                                                         \
        _Pragma("unroll") for (int j = 0; j < RB; ++j)                                                            \
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc_b, LDS_PTR(Bs_ + (j * 4 + wave) * 1024), 16, b_off[j],  \
                                                     ((T) * cpt + chunk) * 128, 0, 0);                            \
    }
#define STENCIL_STEP(T)                                                                    \
    {                                                                                      \
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                   \
        __syncthreads();                                                                   \
        ISSUE_STENCIL((T) + 1, st ^ 1)                                                     \
        compute(st);                                                                       \
        st ^= 1;                                                                           \
    }
        int st = 0;
        ISSUE_STENCIL(0, 0)
        for (;;) {
            STENCIL_STEP(0) STENCIL_STEP(1) STENCIL_STEP(2) STENCIL_STEP(3)
            STENCIL_STEP(4) STENCIL_STEP(5) STENCIL_STEP(6) STENCIL_STEP(7)
            // tap 8: its successor is tap 0 of the next chunk, the first correction step, or nothing
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            const bool last_chunk = chunk + 1 == cpt;
            if (!last_chunk) {
                ++chunk;
                const int cb = chunk << 6;
                second = cb >= a.c0;
                cb_off = (unsigned)((second ? cb - a.c0 : cb) * 2);
                ISSUE_STENCIL(0, st ^ 1)
            } else if (ncorr > 0) {
                tap_lo = 9;
                is_tap = 9;
                is_chunk = 0;
                issue(st ^ 1);
            }
            compute(st);
            st ^= 1;
            if (last_chunk) break;
        }
#undef STENCIL_STEP
#undef ISSUE_STENCIL
        for (int k = 0; k < ncorr * cpt; ++k) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (k + 1 < ncorr * cpt) issue(st ^ 1);
            compute(st);
            st ^= 1;
        }
    } else {
        // split-K (small batches): blockIdx.y walks its share [k0, k1) of the K steps; K order is taps fastest, then chunks
        const int nsplit = a.splitk > 1 ? a.splitk : 1;
        const int k0 = (int)((long long)nk * blockIdx.y / nsplit), k1 = (int)((long long)nk * (blockIdx.y + 1) / nsplit);
        is_chunk = k0 / ntaps;
        is_tap = k0 - is_chunk * ntaps;
        issue(0);
        for (int k = k0; k < k1; ++k) {
            const int cur = (k - k0) & 1;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (k + 1 < k1) issue(cur ^ 1);
            compute(cur);
        }
    }

    if (!UP && a.splitk > 1) {
        // split-K partial: raw fp32 accumulators to splitk_ws[split][m][cout]; conv_splitk_finish_kernel adds the splits in
        // index order (deterministic) and runs the epilogue.  Lane (l31, hi) of accumulator (i, j): pixel l31 of sub-tile i,
        // channels j*32 + 8*g + 4*hi + {0..3}.
        const int l31s = lane & 31, his = lane >> 5;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31s;
            if (tm * BM + row >= M) continue;
            int m;
            if (tile2d) m = (t_img * a.ho + t_y0 + (row >> 4)) * a.wo + t_x0 + (row & 15);
            else m = tm * BM + row;
            float* dst = a.splitk_ws + ((size_t)blockIdx.y * M + m) * a.cout + tn * BN + wn * 64 + 4 * his;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g)
                    *(float4*)(dst + j * 32 + 8 * g) = make_float4(acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]);
        }
        return;
    }

    // ---- epilogue straight from the accumulators.  The operands are swapped (D^T = W X^T), so lane (l31, hi) of
    // accumulator (i, j) holds PIXEL l31 of pixel sub-tile i and CHANNELS j*32 + 8*g + 4*hi + {0..3} (g = register group):
    // affine / activation / dropout / residual run in registers, register groups 2*g2 and 2*g2+1 are packed to bf16 and
    // exchanged between lanes l and l+32 (v_permlane32_swap) so that every lane stores 8 consecutive channels (16 B).
    // No LDS round trip, no barrier; (activation, dropout mode) are wave-uniform and dispatched once.
    auto epilogue = [&](auto act_c, auto mode_c) {
        constexpr int ACT = decltype(act_c)::value, MODE = decltype(mode_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int row = wm * 64 + i * 32 + l31;
            const bool valid = tm * BM + row < M;
            int m, n_img;  // m: output pixel index in the NHWC output tensor
            if (tile2d) {
                n_img = t_img;
                const int y = t_y0 + (row >> 4), x = t_x0 + (row & 15);
                m = UP ? (n_img * a.ho + 2 * y + py) * a.wo + 2 * x + px : (n_img * a.ho + y) * a.wo + x;
            } else {
                m = valid ? tm * BM + row : 0;
                n_img = m / plane;
            }
            const uint32_t ob = (uint32_t)m * (uint32_t)a.cout + (uint32_t)(tn * BN + wn * 64);
            const RngKey key = drop_row_key(a.drop, n_img);  // dropout streams are per batch row
            const uint32_t row0 = (uint32_t)n_img * (uint32_t)(a.ho * a.wo * a.cout);
            const uint32_t cb = (uint32_t)((a.coef_div > 1 ? n_img / a.coef_div : n_img) * a.coef_stride + tn * BN + wn * 64 + 4 * hi);
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int g2 = 0; g2 < 2; ++g2) {
                    const int cg0 = j * 32 + 16 * g2;  // + 4*hi: own channels of group 2*g2; + 8: group 2*g2+1
                    const float4 ca0 = *(const float4*)(a.coef_a + cb + cg0), ca1 = *(const float4*)(a.coef_a + cb + cg0 + 8);
                    const float4 cc0 = *(const float4*)(a.coef_c + cb + cg0), cc1 = *(const float4*)(a.coef_c + cb + cg0 + 8);
                    const float ps = drop_prescale<ACT, MODE>(a.drop);  // dropout scale folded into the affine
                    const float ca[8] = {ca0.x * ps, ca0.y * ps, ca0.z * ps, ca0.w * ps, ca1.x * ps, ca1.y * ps, ca1.z * ps, ca1.w * ps};
                    const float cc[8] = {cc0.x * ps, cc0.y * ps, cc0.z * ps, cc0.w * ps, cc1.x * ps, cc1.y * ps, cc1.z * ps, cc1.w * ps};
                    const uint32_t e0 = ob + cg0 + 4 * hi;
                    float v[8];
#pragma unroll
                    for (int t = 0; t < 8; ++t) v[t] = fmaf(acc[i][j][8 * g2 + t], ca[t], cc[t]);
                    act_drop_fixed<4, ACT, MODE, true>(v, e0, row0, a.drop, key);
                    act_drop_fixed<4, ACT, MODE, true>(v + 4, e0 + 8, row0, a.drop, key);
                    if (a.residual) {
                        const uint2 r0 = valid ? *(const uint2*)(a.residual + (size_t)e0) : make_uint2(0, 0);
                        const uint2 r1 = valid ? *(const uint2*)(a.residual + (size_t)e0 + 8) : make_uint2(0, 0);
                        const uint32_t rw[4] = {r0.x, r0.y, r1.x, r1.y};
#pragma unroll
                        for (int t = 0; t < 8; ++t)
                            v[t] += (t & 1) ? el16_hi(rw[t >> 1]) : el16_lo(rw[t >> 1]);
                    }
                    if (a.out_f32 && valid) {
                        *(float4*)(a.out_f32 + (size_t)e0) = make_float4(v[0], v[1], v[2], v[3]);
                        *(float4*)(a.out_f32 + (size_t)e0 + 8) = make_float4(v[4], v[5], v[6], v[7]);
                    }
                    if (a.out_el16) {
                        uint32_t p0 = pack_el16x2(v[0], v[1]), p1 = pack_el16x2(v[2], v[3]);
                        uint32_t q0 = pack_el16x2(v[4], v[5]), q1 = pack_el16x2(v[6], v[7]);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(p0, q0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(p1, q1, false, false);
                        // lanes 0-31: channels cg0 + 0..7 (own group 2*g2 + partner's); lanes 32-63: cg0 + 8..15
                        uint4 o;
                        o.x = s0[0]; o.y = s1[0]; o.z = s0[1]; o.w = s1[1];
                        if (valid) *(uint4*)(a.out_el16 + (size_t)(ob + cg0 + 8 * hi)) = o;
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
#endif
}

// ------------------------------------------------------------------------------------------------ split-K finish
// Small batches leave the small-plane / deep-K layers (enc1-enc5, dec1, dec2 of unet_simple) with a handful of workgroups that
// each walk up to 144 K steps: a forward at NB = 1 spent 300 of its 316 us there.  conv_igemm_kernel<128,128> then splits the
// K steps over blockIdx.y; this kernel adds the partials in split order and applies the epilogue of conv_direct_kernel.
__global__ void conv_splitk_finish_kernel(ConvArgs a, long long M) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= M * a.cout) return;
    const int co = (int)(i % a.cout);
    const long long m = i / a.cout;
    const int n = (int)(m / ((long long)a.ho * a.wo));
    float acc = 0.0f;
    for (int s = 0; s < a.splitk; ++s) acc += a.splitk_ws[((size_t)s * M + m) * a.cout + co];
    const size_t ci = (size_t)(a.coef_div > 1 ? n / a.coef_div : n) * a.coef_stride + co;
    float v = fmaf(acc, a.coef_a[ci], a.coef_c[ci]);
    v = apply_act(v, a.act);
    v = drop_apply(v, (uint32_t)i, (uint32_t)n * (uint32_t)(a.ho * a.wo * a.cout), a.drop, drop_row_key(a.drop, n));
    if (a.residual) v += el16_to_f32(a.residual[i]);
    if (a.out_el16) a.out_el16[i] = f32_to_el16(v);
    if (a.out_f32) a.out_f32[i] = v;
}

// The same finish with 4 consecutive channels per thread (cout % 4 == 0): 16-byte loads of the partials, one keep word per pair,
// 8-byte stores -- the finish of the rows kernels' split-K launches (conv_halo_rows.hip), whose outputs are whole decoder planes.
__global__ __launch_bounds__(256) void conv_splitk_finish4_kernel(ConvArgs a, long long M) {
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = M * a.cout;
    const long long i = q * 4;
    if (i >= total) return;
    const int co = (int)(i % a.cout);
    const long long m = i / a.cout;
    const int plane = a.ho * a.wo;
    const int n = (int)(m / plane);
    float4 s4 = *(const float4*)(a.splitk_ws + (size_t)i);
    for (int s = 1; s < a.splitk; ++s) {
        const float4 t4 = *(const float4*)(a.splitk_ws + (size_t)s * (size_t)total + (size_t)i);
        s4.x += t4.x; s4.y += t4.y; s4.z += t4.z; s4.w += t4.w;
    }
    const size_t ci = (size_t)(a.coef_div > 1 ? n / a.coef_div : n) * a.coef_stride + co;
    const float4 ca = *(const float4*)(a.coef_a + ci), cc = *(const float4*)(a.coef_c + ci);
    float v[4] = {fmaf(s4.x, ca.x, cc.x), fmaf(s4.y, ca.y, cc.y), fmaf(s4.z, ca.z, cc.z), fmaf(s4.w, ca.w, cc.w)};
    act_drop<4>(v, (uint32_t)i, (uint32_t)n * (uint32_t)(plane * a.cout), a.act, a.drop, drop_row_key(a.drop, n));
    if (a.residual) {
        const uint2 rr = *(const uint2*)(a.residual + (size_t)i);
        v[0] += el16_lo(rr.x); v[1] += el16_hi(rr.x); v[2] += el16_lo(rr.y); v[3] += el16_hi(rr.y);
    }
    if (a.out_f32) *(float4*)(a.out_f32 + (size_t)i) = make_float4(v[0], v[1], v[2], v[3]);
    if (a.out_el16) *(uint2*)(a.out_el16 + (size_t)i) = make_uint2(pack_el16x2(v[0], v[1]), pack_el16x2(v[2], v[3]));
}

hipError_t launch_conv_splitk_finish4(const ConvArgs& a, long long M, hipStream_t stream) {
    if ((a.cout & 3) != 0 || a.splitk_ws == nullptr || a.splitk < 1) return hipErrorInvalidValue;
    const long long quads = M * a.cout / 4;
    hipLaunchKernelGGL(conv_splitk_finish4_kernel, dim3((unsigned)((quads + 255) / 256)), dim3(256), 0, stream, a, M);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------ launchers
bool conv_mfma_supported(const ConvArgs& a) {
    if (a.act == ACT_GELU) return false;  // the MFMA epilogues are instantiated for none / ReLU / LeakyReLU / SiLU
    if (!(a.c0 > 0 && (a.c0 % 64) == 0 && (a.c1 % 64) == 0 && (a.cout % 64) == 0)) return false;
    if (a.kh * a.kw > 32) return false;                       // tap-validity mask is 32 bits
    const size_t npix = (size_t)a.n * a.h * a.w;              // 32-bit buffer offsets
    const size_t lim = 0xFFFFFFF0ull;
    const size_t wtaps = a.up2x ? 64 : (size_t)a.kh * a.kw;
    if (!(npix * a.c0 * 2 < lim && npix * (size_t)a.c1 * 2 < lim && (size_t)a.cout * wtaps * (a.c0 + a.c1) * 2 < lim))
        return false;
    if (a.up2x) {  // fused x2 upsample: 3x3/s1/p1 on a low-res plane that tiles into TH x 16
        const int th = (a.cout % 128 == 0) ? 8 : 16;
        return a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.w % 16 == 0 && a.h % th == 0 &&
               a.ho == 2 * a.h && a.wo == 2 * a.w && a.wpk_up != nullptr;
    }
    return true;
}

template <int BM, int BN, int WM, int WN, int UP>
static hipError_t launch_igemm(ConvArgs a, hipStream_t stream) {
    constexpr int lds = 2 * (BM + BN) * 128;
    long long M = (long long)a.n * a.ho * a.wo;
    int tiles_m = (int)((M + BM - 1) / BM);
    if (UP) {  // tiles walk the LOW-res plane once per phase
        M = (long long)a.n * a.h * a.w;
        tiles_m = (int)(M / BM) * 4;
        a.wpk = a.wpk_up;
    }
    const int tiles_n = a.cout / BN;
    a.splitk = 1;
    if (!UP && BM == 128 && a.splitk_ws) {
        // Split factor: the layer's K depth allows nk / 8 (<= 16); the launch takes as much of it as is needed to fill the 512
        // resident workgroups (power of two), and only when its own tiles fill less than a quarter of them -- a layer with many
        // output pixels pays more for writing / re-reading the partials than it gains (enc1 at NB = 7: 224 tiles, 58 MB of
        // partials).  Tiny problems (<= 32 tiles) always get the layer's full factor, so small batches, their row splits and
        // the paired launches sum in the same order and stay bit-identical.
        const bool enabled = !(dyf_form("DYF_SPLITK") && atoi(dyf_form("DYF_SPLITK")) == 0);
        const int nk = a.kh * a.kw * ((a.c0 + a.c1) >> 6);
        const long long tiles = (((long long)(a.n_sel > 0 ? a.n_sel : a.n) * a.ho * a.wo + BM - 1) / BM) * tiles_n;
        const long long max_tiles = dyf_form("DYF_SPLITK_MAX_TILES") ? atoll(dyf_form("DYF_SPLITK_MAX_TILES")) : 128;
        const long long fill = dyf_form("DYF_SPLITK_FILL") ? atoll(dyf_form("DYF_SPLITK_FILL")) : 512;
        int s = std::min(16, nk / 8);
        while (s > 1 && s * tiles > fill) s >>= 1;
        const long long need = (long long)s * M * a.cout;
        if (enabled && s > 1 && tiles <= max_tiles && need <= a.splitk_cap) a.splitk = s;
    }
    dyf_form_note(a.splitk > 1 ? (BM == 128 ? "conv_igemm_kernel<128,128>+splitk" : "conv_igemm_kernel<256,64>+splitk")
                               : (BM == 128 ? (UP ? "conv_igemm_kernel<128,128,UP>" : "conv_igemm_kernel<128,128>")
                                            : (UP ? "conv_igemm_kernel<256,64,UP>" : "conv_igemm_kernel<256,64>")), a.n);
    hipLaunchKernelGGL((conv_igemm_kernel<BM, BN, WM, WN, UP>), dim3(tiles_m * tiles_n, a.splitk), dim3(256), lds, stream, a, (int)M,
                       tiles_m, tiles_n);
    if (a.splitk > 1)
        hipLaunchKernelGGL(conv_splitk_finish_kernel, dim3((unsigned)((M * a.cout + 255) / 256)), dim3(256), 0, stream, a, M);
    return hipGetLastError();
}

// Raise the dynamic-LDS cap of the MFMA kernels once per process (not legal inside a stream capture).
hipError_t conv_init() {
    hipError_t e = hipSuccess;
#define SET_LDS(BM, BN, WM, WN, UP)                                                                     \
    if (e == hipSuccess)                                                                                \
        e = hipFuncSetAttribute((const void*)conv_igemm_kernel<BM, BN, WM, WN, UP>,                     \
                                hipFuncAttributeMaxDynamicSharedMemorySize, 2 * (BM + BN) * 128);
    SET_LDS(128, 128, 2, 2, 0) SET_LDS(128, 128, 2, 2, 1) SET_LDS(256, 64, 4, 1, 0) SET_LDS(256, 64, 4, 1, 1)
#undef SET_LDS
    if (e == hipSuccess) e = conv_up_halo_init();
    if (e == hipSuccess) e = conv_halo_rows_init();
    if (e == hipSuccess) e = conv_igemm2_init();
    if (e == hipSuccess) e = conv_enc0_stem_init();
    return e;
}

// plain 3x3 / s1 convs with 64 or 128 (a multiple of 64 that is not one of 256) output channels on planes of any size: SP = 5 of the
// halo kernel, when the 16 x 32 tiles cover the plane reasonably (>= 60 %) and the launch has enough of them
static bool halo5_policy(const ConvArgs& a) {
    const bool h5_all = dyf_form("DYF_HALO5_ALL") && atoi(dyf_form("DYF_HALO5_ALL")) != 0;
    if (!(!a.up2x && a.kh == 3 && a.kw == 3 && a.stride == 1 && a.cout % 64 == 0 && (a.cout % 256 != 0 || h5_all) && a.out_f32 == nullptr &&
          a.residual == nullptr))
        return false;
    if (dyf_form("DYF_HALO5") && atoi(dyf_form("DYF_HALO5")) == 0) return false;
    const long long h5_min = dyf_form("DYF_HALO5_MIN_TILES") ? atoll(dyf_form("DYF_HALO5_MIN_TILES")) : 64;
    const long long nsel = a.n_sel > 0 ? a.n_sel : a.n;
    ConvArgs b = a;
    b.wpk_up_frag = conv_lookup_halo3_frag(b.wpk);
    const long long ty = (a.h + 15) / 16, tx = (a.w + 31) / 32;
    const long long tiles5 = nsel * ty * tx * (a.cout / 64);
    const bool covers = 10ll * a.h * a.w >= 6ll * ty * 16 * tx * 32;
    return b.wpk_up_frag && covers && tiles5 >= h5_min && conv_halo5_supported(b);
}

bool conv_plain3x3_takes_halo5(const ConvArgs& a) {
    // (the forms launch_conv_stats tries BEFORE SP = 5 need up2x or cout % 256 == 0: a conv that passes halo5_policy reaches it)
    return conv_mfma_supported(a) && a.gn_part == nullptr && a.gnf.gran == nullptr && halo5_policy(a);
}

hipError_t launch_conv(const ConvArgs& a, int path, hipStream_t stream) {
    if (a.gn_part == nullptr) return launch_conv_stats(a, path, stream, nullptr);
    ConvArgs b = a;  // statistics are only produced through launch_conv_stats (the caller must learn whether they were)
    b.gn_part = nullptr;
    return launch_conv_stats(b, path, stream, nullptr);
}

hipError_t launch_conv_stats(const ConvArgs& a_in, int path, hipStream_t stream, int* gn_slots) {
    if (gn_slots) *gn_slots = 0;
    ConvArgs a = a_in;
    float* const gn_part = gn_slots ? a_in.gn_part : nullptr;
    a.gn_part = nullptr;  // only the form below that produces statistics sees the buffer
    a.gn_slots = 0;
    const long long nsel = a.n_sel > 0 ? a.n_sel : a.n;  // rows the kernel form is chosen for (ConvArgs::n_sel)
    // a fused nearest upsample exists in ONE form: refuse rather than read a low-resolution tensor as the full-size one
    if (a.up_nearest && !(path == 1 && conv_mfma_supported(a) && halo5_policy(a) && a.h % 2 == 0 && a.w % 2 == 0 && a.c1 == 0)) return hipErrorInvalidValue;
    if (path == 1 && conv_mfma_supported(a)) {
        const bool use_halo = !(dyf_form("DYF_UP_HALO") && atoi(dyf_form("DYF_UP_HALO")) == 0);
        // halo form from 32 x 32 low-res planes on; below that (dec2: 16 x 16, 2 tiles per image) the materialised upsample +
        // plain 3x3 halo conv is still slightly ahead (7 715 vs 7 690 fields/s with DYF_HALO_MIN_PLANE=16: 640 workgroups of
        // the fused form fill 1.25 rounds of the 512 resident ones)
        const int halo_min = dyf_form("DYF_HALO_MIN_PLANE") ? atoi(dyf_form("DYF_HALO_MIN_PLANE")) : 32;
        if (a.up2x && a.up_cols)  // sparse-column form: only the halo kernel writes the compact output tensor
            return conv_up_halo_supported(a) ? launch_conv_up_halo(a, stream) : hipErrorInvalidValue;
        if (a.up2x && use_halo && a.h >= halo_min && a.w >= halo_min && conv_up_halo_supported(a)) return launch_conv_up_halo(a, stream);
        // plain 3x3 / s1 convs with cout % 256 == 0 on 8x16-tileable planes: the halo kernel (one window DMA per chunk
        // instead of one gather per tap); DYF_HALO3=0 disables, DYF_HALO3_MIN_TILES sets the smallest launch (measured at NB = 80,
        // enc3 with 320 tiles 115 -> 94 us; round 4, with the rows forms: from 80 tiles on -- NS at 7 / 10 / 25 rows +3.4 / +5.7 /
        // +2.5 % against the 256 of rounds 1-3, nothing lost at 4 or 80 rows; 64 costs 2.4 % at 4 rows)
        const bool h5_all = dyf_form("DYF_HALO5_ALL") && atoi(dyf_form("DYF_HALO5_ALL")) != 0;
        if (!a.up2x && a.kh == 3 && a.kw == 3 && a.cout % 256 == 0 && !h5_all && a.out_f32 == nullptr && a.residual == nullptr) {
            const char* h3 = dyf_form("DYF_HALO3");
            if (!(h3 && atoi(h3) == 0)) {
                ConvArgs b = a;
                b.wpk_up_frag = conv_lookup_halo3_frag(b.wpk);
                const char* mt3 = dyf_form("DYF_HALO3_MIN_TILES");
                const long long tiles3 = (nsel * a.h * a.w / 128) * (a.cout / 256);
                const bool rows = !(dyf_form("DYF_HALO_ROWS") && atoi(dyf_form("DYF_HALO_ROWS")) == 0);
                if (b.wpk_up_frag && tiles3 >= (mt3 ? atoll(mt3) : 80) && conv_halo3_supported(b))
                    return rows && conv_halo_rows3_supported(b) ? launch_conv_halo_rows3(b, stream) : launch_conv_halo3(b, stream);
            }
        }
        // 3x3 / s1 convs with 64 or 128 (any multiple of 64 that is not one of 256) output channels -- the ResNet-UNet levels --
        // on planes of any size: SP = 5 of the halo kernel, when the 16 x 32 tiles cover the plane reasonably (>= 60 %: not
        // 15 x 15).  DYF_HALO5=0 disables, DYF_HALO5_MIN_TILES sets the smallest launch (64 tiles since round 4: with the GroupNorm
        // fused into this form a small launch also saves the three GroupNorm kernels behind the implicit-GEMM fallback -- OISST
        // shapes at 38 / 75 rows +5.8 / +3 % against the 256 of round 3).
        if (halo5_policy(a)) {
            {
                ConvArgs b = a;
                b.wpk_up_frag = conv_lookup_halo3_frag(b.wpk);
                {
                    if (gn_part && a.act == ACT_NONE && a.drop.mode == 0) {  // statistics of the raw conv output
                        b.gn_part = gn_part;
                        b.gn_slots = conv_halo5_gn_slots(a.h, a.w);
                        *gn_slots = b.gn_slots;
                    }
                    return launch_conv_halo5(b, stream);
                }
            }
        }
        if (a.pix_pitch0 == 16 && conv_enc0_stem_supported(a)) {  // enc0 on the fused stem: HBM-bound, its own persistent kernel
            const el16_t* f = conv_lookup_halo3_frag(a.wpk);
            if (f) return launch_conv_enc0_stem(a, f, stream);
        }
        if (!a.up2x && a.kh == 4 && a.kw == 4 && a.stride == 2 && a.cout % 128 == 0 && a.c1 == 0 && a.out_f32 == nullptr &&
            a.residual == nullptr && a.pix_pitch0 == 0) {  // 4x4 / s2 convs: the same kernel on the space-to-depth view
            const char* h3 = dyf_form("DYF_HALO3");
            if (!(h3 && atoi(h3) == 0)) {
                ConvArgs b = a;
                b.wpk_up_frag = conv_lookup_halo3_frag(b.wpk);
                const char* mt3 = dyf_form("DYF_HALO_S2_MIN_TILES");  // (its own switch since round 5; DYF_HALO3_MIN_TILES still applies when unset)
                if (!mt3) mt3 = dyf_form("DYF_HALO3_MIN_TILES");
                // cout % 256 == 0: 8 x 16 tiles x 256 channels; else 16 x 16 tiles x 128 channels
                const long long tiles3 = a.cout % 256 == 0 ? (nsel * a.ho * a.wo / 128) * (a.cout / 256)
                                                           : (nsel * a.ho * a.wo / 256) * (a.cout / 128);
                if (b.wpk_up_frag && tiles3 >= (mt3 ? atoll(mt3) : 80) && conv_halo_s2_supported(b)) return launch_conv_halo_s2(b, stream);
            }
        }
        const bool use_igemm2 = !(dyf_form("DYF_IGEMM2") && atoi(dyf_form("DYF_IGEMM2")) == 0);
        if (!a.up2x && use_igemm2 && a.cout % 64 == 0) {  // cout % 128 == 0: 256 x 128 tiles, else 256 x 64
            ConvArgs b = a;
            if (!b.wpk_frag) b.wpk_frag = conv_lookup_frag(b.wpk);
            // 256 x 128 tiles pay off once they fill the chip (2 workgroups x 256 CUs); below that the 128 x 128 form's
            // finer tiles win (measured at NB = 50: dec2/enc2 with 400 tiles +9 %/+4 %, enc3 with 200 tiles -20 %)
            const long long tiles2 = ((nsel * a.ho * a.wo + 255) / 256) * (a.cout % 128 == 0 ? a.cout / 128 : a.cout / 64);
            const char* mt = dyf_form("DYF_IGEMM2_MIN_TILES");  // tests force the form on small problems
            const long long min_tiles = mt ? atoll(mt) : 384;
            if (tiles2 >= min_tiles && conv_igemm2_supported(b)) return launch_conv_igemm2(b, stream);
        }
        // few rows: 1x1 / 2x2-s2 convs whose 128 x 128 tiles would not even fill a quarter of the chip (the split-K regime of
        // launch_igemm) run on conv_skinny_kernel -- K split over the four waves of a 32 x 32 tile, one launch (DYF_SKINNY=0 disables)
        if (!a.up2x && a.cout % 128 == 0) {
            const bool skinny = !(dyf_form("DYF_SKINNY") && atoi(dyf_form("DYF_SKINNY")) == 0);
            const long long tiles128 = ((nsel * a.ho * a.wo + 127) / 128) * (a.cout / 128);
            ConvArgs b = a;
            if (!b.wpk_frag) b.wpk_frag = conv_lookup_frag(b.wpk);
            // (64 tiles of 128 x 128: NS at 1 / 4 / 7 / 10 rows +10.6 / +4 / +2 / +1 %, nothing lost at 25 / 38; at 128 the 25- and
            // 38-row rollouts lose 2.5 %)
            const long long sk_max = dyf_form("DYF_SKINNY_MAX_TILES") ? atoll(dyf_form("DYF_SKINNY_MAX_TILES")) : 64;
            if (skinny && tiles128 <= sk_max && conv_skinny_supported(b)) return launch_conv_skinny(b, stream);
        }
        if (a.cout % 128 == 0)
            return a.up2x ? launch_igemm<128, 128, 2, 2, 1>(a, stream) : launch_igemm<128, 128, 2, 2, 0>(a, stream);
        return a.up2x ? launch_igemm<256, 64, 4, 1, 1>(a, stream) : launch_igemm<256, 64, 4, 1, 0>(a, stream);
    }
    if (a.up2x) return hipErrorInvalidValue;  // the direct kernel has no fused-upsample form: caller materialises
    const long long total = (long long)a.n * a.ho * a.wo * a.cout;
    const int threads = 256;
    dyf_form_note("conv_direct_kernel", a.n);
    hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)((total + threads - 1) / threads)), dim3(threads), 0, stream,
                       a, total);
    return hipGetLastError();
}

// ---- GroupNorm fused into the conv (gn_fused.h).  The form is used exactly where the un-fused launch would have taken
// conv_up_halo_kernel<5> / conv_igemm2_kernel<2> (same tile thresholds), when a sample's slots are few enough to sweep.
static const int GN_FUSE_MAX_SLOTS = 64;  // gn_fuse_sweep<16>: 4 slot classes x 16

int conv_gn_fused_max_slots(int h, int w) {
    int best = 0;
    const int s5 = conv_halo5_gn_slots(h, w);
    if (s5 <= GN_FUSE_MAX_SLOTS) best = s5;
    const int s16 = conv_gn16_slots(h, w);
    if (s16 <= GN_FUSE_MAX_SLOTS) best = std::max(best, s16);
    const int si = conv_igemm2_gn_slots(h, w);
    if (si > 0 && si <= GN_FUSE_MAX_SLOTS) best = std::max(best, si);
    const int s128 = conv_igemm2_gn_slots_bm128(h, w);
    if (s128 > 0 && s128 <= GN_FUSE_MAX_SLOTS) best = std::max(best, s128);
    return best;
}

hipError_t launch_conv_gn_fused(const ConvArgs& a_in, int path, hipStream_t stream, bool* fused) {
    *fused = false;
    ConvArgs a = a_in;  // (DYF_GN_FUSED=0 is read per engine, dyf_engine_create: the caller then never asks)
    a.gn_part = nullptr;
    a.gn_slots = 0;
    const GnFuse& G = a.gnf;
    if (path != 1 || G.gran == nullptr || G.epoch == nullptr || a.act != ACT_SILU || a.drop.mode == 2 || a.out_el16 == nullptr ||
        a.out_f32 != nullptr || a.up2x || !conv_mfma_supported(a))
        return hipSuccess;
    const int cpg = G.groups > 0 ? a.cout / G.groups : 0;
    if (cpg < 8 || cpg % 8 != 0 || 64 % cpg != 0 || cpg * G.groups != a.cout) return hipSuccess;  // a group lies inside one 64-channel block
    const long long nsel = a.n_sel > 0 ? a.n_sel : a.n;
    if (a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.cout % 256 == 0) {
        // 256-channel level on SMALL planes (15 x 15 at OISST: one 16 x 16 tile per sample): conv_gn16_kernel with four 64-channel column
        // blocks per sample instead of conv_igemm2_kernel<2, true>'s 256-pixel x 128-channel tiles -- half the K chain per workgroup,
        // more than twice the workgroups (400 against 176 at 100 rows).  DYF_GN16_C256=0: off
        const bool on = !(dyf_form("DYF_GN16") && atoi(dyf_form("DYF_GN16")) == 0) && !(dyf_form("DYF_GN16_C256") && atoi(dyf_form("DYF_GN16_C256")) == 0);
        ConvArgs b = a;
        b.wpk_up_frag = conv_lookup_frag64(b.wpk);
        const int slots16 = conv_gn16_slots(a.h, a.w);
        const long long tiles16 = nsel * slots16 * (a.cout / 64);
        const bool covers16 = 10ll * a.h * a.w >= 6ll * slots16 * 256 || (dyf_form("DYF_GN16_ANY_PLANE") && atoi(dyf_form("DYF_GN16_ANY_PLANE")) != 0);
        const long long max_plane = dyf_form("DYF_GN16_C256_MAX_PLANE") ? atoll(dyf_form("DYF_GN16_C256_MAX_PLANE")) : 1024;
        const long long c256_min = dyf_form("DYF_GN16_MIN_TILES") ? atoll(dyf_form("DYF_GN16_MIN_TILES")) : 64;
        if (on && b.wpk_up_frag && covers16 && (long long)a.h * a.w <= max_plane && tiles16 >= c256_min && slots16 <= GN_FUSE_MAX_SLOTS &&
            slots16 <= G.max_slots && conv_gn16_supported(b)) {
            b.gnf.slots = slots16;
            *fused = true;
            return launch_conv_gn16(b, stream);
        }
    }
    if (a.kh == 3 && a.kw == 3 && a.stride == 1 && a.pad == 1 && a.cout % 64 == 0 && a.cout % 256 != 0) {
        const bool h5 = !(dyf_form("DYF_HALO5") && atoi(dyf_form("DYF_HALO5")) == 0);
        const long long h5_min = dyf_form("DYF_HALO5_MIN_TILES") ? atoll(dyf_form("DYF_HALO5_MIN_TILES")) : 64;
        ConvArgs b = a;
        b.wpk_up_frag = conv_lookup_halo3_frag(b.wpk);
        {   // 16 x 16 tiles, three workgroups per CU (conv_gn16.hip; DYF_GN16=0: the 16 x 32 form below)
            const bool g16 = !(dyf_form("DYF_GN16") && atoi(dyf_form("DYF_GN16")) == 0);
            const long long g16_min = dyf_form("DYF_GN16_MIN_TILES") ? atoll(dyf_form("DYF_GN16_MIN_TILES")) : 64;
            const int slots16 = conv_gn16_slots(a.h, a.w);
            const long long tiles16 = nsel * slots16 * (a.cout / 64);
            // (planes that fill less than 60 % of their tiles are left to the other forms; DYF_GN16_ANY_PLANE=1: the tests' tiny planes)
            const bool covers16 = 10ll * a.h * a.w >= 6ll * slots16 * 256 || (dyf_form("DYF_GN16_ANY_PLANE") && atoi(dyf_form("DYF_GN16_ANY_PLANE")) != 0);
            if (g16 && b.wpk_up_frag && covers16 && tiles16 >= g16_min && slots16 <= GN_FUSE_MAX_SLOTS && slots16 <= G.max_slots &&
                conv_gn16_supported(b)) {
                b.gnf.slots = slots16;
                *fused = true;
                return launch_conv_gn16(b, stream);
            }
        }
        const long long ty = (a.h + 15) / 16, tx = (a.w + 31) / 32;
        const long long tiles5 = nsel * ty * tx * (a.cout / 64);
        const bool covers = 10ll * a.h * a.w >= 6ll * ty * 16 * tx * 32;
        const int slots = conv_halo5_gn_slots(a.h, a.w);
        if (h5 && b.wpk_up_frag && covers && tiles5 >= h5_min && slots <= GN_FUSE_MAX_SLOTS && slots <= G.max_slots && conv_halo5_supported(b)) {
            b.gnf.slots = slots;
            *fused = true;
            return launch_conv_halo5(b, stream);
        }
    }
    const bool use_igemm2 = !(dyf_form("DYF_IGEMM2") && atoi(dyf_form("DYF_IGEMM2")) == 0);
    if (use_igemm2 && a.cout % 128 == 0) {
        ConvArgs b = a;
        if (!b.wpk_frag) b.wpk_frag = conv_lookup_frag(b.wpk);
        const long long tiles2 = ((nsel * a.ho * a.wo + 255) / 256) * (a.cout / 128);
        // un-fused, the 256 x 128 tiles pay off from 384 tiles on (below that conv_igemm_kernel<128, 128> is ahead); fused, the form
        // also saves the three GroupNorm launches behind it (statistics, finalise, apply: 15 us of launches at small batches): taken
        // from 32 tiles on.  Measured at the end of round 4, OISST shapes, fields/s with the threshold at 256 (the first choice) /
        // 64 / 16: 300 rows 4 154 / 4 165 / 4 181, 150 rows 3 568 / 3 626 / 3 631, 75 rows 2 360 / 2 494 / 2 479, 38 rows 1 548 /
        // 1 619 / 1 654, 16 rows 811 / 811 / 791 (32: 818) -- DYF_GN_FUSE_MIN_TILES overrides, DYF_IGEMM2_MIN_TILES (tests) wins
        const char* mt = dyf_form("DYF_IGEMM2_MIN_TILES");
        const char* mf = dyf_form("DYF_GN_FUSE_MIN_TILES");
        const long long min_tiles = mt ? atoll(mt) : mf ? atoll(mf) : 32;
        const int slots = conv_igemm2_gn_slots(a.ho, a.wo);
        // flattened-M tiles cut a sample into 128-row slabs at (n * plane) % 128: unless plane % 128 == 0 (or the tiles are 2-D) the
        // fp32 partial sums of a sample are grouped by its POSITION in the launch, and (mean, 1/std) differ in the last bits between
        // batch offsets / ranks -- not acceptable to a batch_invariant engine, which then takes the three-kernel path
        const bool position_free = conv_igemm2_tile2d(a.ho, a.wo) || (a.ho * a.wo) % 128 == 0;
        if (G.invariant && !position_free) return hipSuccess;
        // few tiles: the 128-pixel tile form (half the K chain per wave, twice the workgroups) while the 256-pixel tiles would leave
        // CUs idle -- DYF_IGEMM2_BM128_BELOW tiles (0 = never); not for batch_invariant engines whose planes are not slab-aligned
        // (the same position argument as above, with 64-row slabs)
        const char* b128 = dyf_form("DYF_IGEMM2_BM128_BELOW");  // read per launch (parity test)
        const long long bm128_below = b128 ? atoll(b128) : 224;
        const int slots128 = conv_igemm2_gn_slots_bm128(a.ho, a.wo);
        const bool free128 = (a.wo % 16 == 0 && a.ho % 8 == 0) || (a.ho * a.wo) % 64 == 0;
        if (tiles2 >= min_tiles && tiles2 < bm128_below && slots128 > 0 && slots128 <= GN_FUSE_MAX_SLOTS && slots128 <= G.max_slots &&
            (!G.invariant || free128) && conv_igemm2_supported(b)) {
            b.gnf.slots = slots128;
            b.gnf.bm = 128;
            *fused = true;
            return launch_conv_igemm2(b, stream);
        }
        if (tiles2 >= min_tiles && slots > 0 && slots <= GN_FUSE_MAX_SLOTS && slots <= G.max_slots && conv_igemm2_supported(b)) {
            b.gnf.slots = slots;
            b.gnf.bm = 256;
            *fused = true;
            return launch_conv_igemm2(b, stream);
        }
    }
    return hipSuccess;
}

// Host-side weight transform for the fused x2-upsample conv (see the kernel header): w [cout][cin][3][3] fp32 ->
// [4 phases][cout][16 taps][cin] bf16.
void pack_up2x_weights(const float* w, int cout, int cin, el16_t* out) {
    // coefficient of w[k] (k = 0,1,2) in the stencil tap a = -1,0,+1 and in the border correction, per phase
    static const double E[2][3][3] = {{{0.75, 0.25, 0.0}, {0.25, 0.75, 0.75}, {0.0, 0.0, 0.25}},
                                      {{0.25, 0.0, 0.0}, {0.75, 0.75, 0.25}, {0.0, 0.25, 0.75}}};
    static const double Cc[2][3] = {{-1.0, 0.0, 0.0}, {0.0, 0.0, -1.0}};
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            const int phase = py * 2 + px;
            for (int co = 0; co < cout; ++co)
                for (int ci = 0; ci < cin; ++ci) {
                    const float* k = w + ((size_t)co * cin + ci) * 9;
                    for (int t = 0; t < 16; ++t) {
                        const double* fy;
                        const double* fx;
                        if (t < 9) { fy = E[py][t / 3]; fx = E[px][t % 3]; }
                        else if (t < 12) { fy = Cc[py]; fx = E[px][t - 9]; }
                        else if (t < 15) { fy = E[py][t - 12]; fx = Cc[px]; }
                        else { fy = Cc[py]; fx = Cc[px]; }
                        double v = 0.0;
                        for (int ky = 0; ky < 3; ++ky)
                            for (int kx = 0; kx < 3; ++kx) v += fy[ky] * fx[kx] * (double)k[ky * 3 + kx];
                        out[(((size_t)phase * cout + co) * 16 + t) * cin + ci] = f32_to_el16((float)v);
                    }
                }
        }
}
