// Shared device/host helpers for the gfx950 DYffusion engine.
#pragma once
#include <hip/hip_runtime.h>
#include <cstring>
#include <stdint.h>

// 16-bit storage element of activations and weights in HBM (NHWC), raw bits.  One source, two builds of the library:
//   default        bfloat16  (libdyffusion_hip.so,     v_mfma_f32_32x32x16_bf16) -- BASELINE configs[1]: "bf16"
//   -DDYF_F16=1    IEEE fp16 (libdyffusion_hip_f16.so, v_mfma_f32_32x32x16_f16)  -- BASELINE configs[4]: "fp16 MFMA conv/attn"
// Same MFMA rate, same bytes; fp16 carries 11 mantissa bits instead of 8 (rollout error 4-8x lower) and overflows at 65504.
// Accumulation, normalisation statistics, FiLM coefficients and the sampler state are fp32 in both builds.
#ifndef DYF_F16
#define DYF_F16 0
#endif
typedef uint16_t el16_t;

// Timing-experiment switches that produce WRONG RESULTS (one part of a kernel's work removed to see what it costs; their findings
// are recorded in DESIGN.md) exist only in experiment builds: tools/build_variant.sh compiles ONE translation unit with
// -DDYF_EXPERIMENT_BUILD plus the switch into tools/variants/libvar_<name>.so.  A product build (__graft_entry__.build) that sees
// any of them stops here, and the run-time ones (DYF_GN_FUSE_NOWAIT, DYF_EXP_DEC5_1316) are not compiled in.
#if !defined(DYF_EXPERIMENT_BUILD) &&                                                                                               \
    (defined(HALO_EXP_NO_DMA_WAIT) || defined(HALO_EXP_NO_STORE) || defined(HALO_EXP_W_ALIAS) || defined(HALO_EXP_W_SHARE) ||        \
     defined(HALO_EXP_NO_HALO) || defined(HALO_EXP_NO_EPI) || defined(HALO_EXP_TIMELINE) || defined(HALO_EXP_LDS_SKIP) || defined(FA_EXP_NO_EXP) ||                \
     defined(FA_EXP_NOSYNC) || defined(FA_EXP_NO_VT) || defined(FA4_X_NOEXP) || defined(FA4_X_MFMAONLY) || defined(FA4_X_NOQK) ||    \
     defined(FA4_X_NOPV) || defined(FA4_X_NOSTAGE))
#error "a wrong-results timing switch (HALO_EXP_* / FA_EXP_* / FA4_X_*) is defined in a product build: use tools/build_variant.sh"
#endif

// Kernel-form log (test seam, include/dyffusion_hip_testing.h dyf_debug_form_log*): every launcher that chooses between kernel
// forms notes the form it took and the batch rows of the launch; off unless a test enables it (one predictable branch).
extern bool g_dyf_form_log_on;
void dyf_form_note_slow(const char* form, long long rows);
static inline void dyf_form_note(const char* form, long long rows) {
    if (g_dyf_form_log_on) dyf_form_note_slow(form, rows);
}

// Kernel-form switches (the A/B + parity harness): which of several equivalent kernel forms a launcher takes, thresholds of the
// form policy, a few wrong-results timing probes.  Their values come ONLY from dyf_debug_set_form (include/dyffusion_hip_testing.h,
// called by tests/ and tools/): the library reads NONE of them from the environment (round 5 did: a stray DYF_* variable changed
// kernel forms, and for the training operands the numerics, under a caller who never asked).  dyf_form(key) = the value set for
// `key`, or nullptr -- one relaxed load when nothing is set, which is every production process.  Keys keep their historic names.
extern int g_dyf_form_count;
const char* dyf_form_slow(const char* key);
static inline const char* dyf_form(const char* key) { return g_dyf_form_count ? dyf_form_slow(key) : nullptr; }

// In-rollout timing of ONE named kernel (bench.py `hbm_kernels`; include/dyffusion_hip_testing.h dyf_time_named_kernel_in_rollout):
// the launchers of the HBM-bound kernels (norm / activation / resample / readout) open a KernelProf scope around their launch with
// the launch's ALGORITHMIC bytes (every operand once); while a name is armed, scopes of that name bracket the launch with HIP events on
// its stream.  Off (one pointer test) everywhere else.
extern const char* g_dyf_prof_name;
void dyf_prof_begin(hipStream_t st, double bytes);
void dyf_prof_end(hipStream_t st);
struct KernelProf {
    hipStream_t st;
    bool on;
    KernelProf(const char* name, hipStream_t s, double algorithmic_bytes) : st(s), on(g_dyf_prof_name && !strcmp(name, g_dyf_prof_name)) {
        if (on) dyf_prof_begin(st, algorithmic_bytes);
    }
    ~KernelProf() { if (on) dyf_prof_end(st); }
};

#if DYF_F16
typedef _Float16 el16_native_t;
#define DYF_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, x, y, z)
#define DYF_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, x, y, z)
#define DYF_DTYPE_NAME "fp16"
#else
typedef __bf16 el16_native_t;
#define DYF_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
#define DYF_MFMA_16x16x32(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, x, y, z)
#define DYF_DTYPE_NAME "bf16"
#endif
typedef el16_native_t el16x8_t __attribute__((ext_vector_type(8)));  // one MFMA operand fragment (8 k-values per lane)
typedef el16_native_t el16x2_native_t __attribute__((ext_vector_type(2)));
typedef float f32x2_native_t __attribute__((ext_vector_type(2)));

// ------------------------------------------------------------------------------------------------ element conversions
#if DYF_F16
__device__ __host__ __forceinline__ float el16_to_f32(el16_t v) { return (float)__builtin_bit_cast(_Float16, v); }
// round-to-nearest-even (v_cvt_f16_f32 / compiler-rt on the host); overflow -> inf as IEEE prescribes
__device__ __host__ __forceinline__ el16_t f32_to_el16(float f) { return __builtin_bit_cast(el16_t, (_Float16)f); }
// low / high element of a packed pair
__device__ __forceinline__ float el16_lo(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w & 0xffffu)); }
__device__ __forceinline__ float el16_hi(uint32_t w) { return (float)__builtin_bit_cast(_Float16, (uint16_t)(w >> 16)); }
#else
__device__ __host__ __forceinline__ float el16_to_f32(el16_t v) {
    const uint32_t u = ((uint32_t)v) << 16;
    return __builtin_bit_cast(float, u);
}
// round-to-nearest-even, NaN preserved
__device__ __host__ __forceinline__ el16_t f32_to_el16(float f) {
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (el16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (el16_t)(u >> 16);
}
__device__ __forceinline__ float el16_lo(uint32_t w) { return __uint_as_float(w << 16); }
__device__ __forceinline__ float el16_hi(uint32_t w) { return __uint_as_float(w & 0xffff0000u); }
#endif

// two fp32 -> packed pair, round-to-nearest-even: one v_cvt_pk_bf16_f32 / v_cvt_pk_f16_f32 on gfx950 (the software form
// costs ~8 VALU per pair and dominated the conv epilogues).  Written as a vector conversion, NOT inline asm: the compiler's
// hazard recogniser does not look inside asm, and a v_exp_f32 / v_rcp_f32 result consumed by the very next VALU
// instruction needs a wait state (an asm v_cvt right behind a v_exp read the stale register).
__device__ __forceinline__ uint32_t pack_el16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2_native_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, el16x2_native_t));
#else
    return (uint32_t)f32_to_el16(lo) | ((uint32_t)f32_to_el16(hi) << 16);
#endif
}

// ------------------------------------------------------------------------------------------------ activations
enum { ACT_NONE = 0, ACT_RELU = 1, ACT_LEAKY = 2, ACT_SILU = 3, ACT_GELU = 4 };

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.0f);
    if (act == ACT_LEAKY) return v > 0.0f ? v : 0.2f * v;
    if (act == ACT_SILU) return v / (1.0f + __expf(-v));
    if (act == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));  // exact (erf) GELU, F.gelu default
    return v;
}

// ------------------------------------------------------------------------------------------------ counter RNG
// Stateless keep-mask generator for MC dropout (SURVEY hard-part (c)).  The keep bit of an element is a pure function of
//   (seed, forward index, GLOBAL batch row, dropout layer, element index inside the row's NHWC tensor),
// so a rollout does not depend on how its rows are batched: sampling rows [a, b) on one GPU, the paired 2 nb-row
// interpolator launches and the one-process-per-GPU sharding of an ensemble all draw the same masks (SURVEY 8e
// "independent sub-streams per row", results invariant to the number of GPUs).  One 32-bit word per PAIR of elements,
// 16 bits each, compared against a 16-bit threshold.  tests/rng_host.py re-implements this in numpy so that RNG-mode
// rollouts are checked against the oracle fed with exactly these masks.
__device__ __host__ __forceinline__ uint32_t fmix32(uint32_t h) {
    h ^= h >> 16;
    h *= 0x85EBCA6Bu;
    h ^= h >> 13;
    h *= 0xC2B2AE35u;
    h ^= h >> 16;
    return h;
}

// 24 x 24 -> low 32 bits multiply: v_mul_u32_u24 is a full-rate VALU op on gfx950, v_mul_lo_u32 is quarter rate
__device__ __host__ __forceinline__ uint32_t mul_u24(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umul24(a, b);
#else
    return (uint32_t)(((uint64_t)(a & 0xFFFFFFu) * (uint64_t)(b & 0xFFFFFFu)) & 0xFFFFFFFFull);
#endif
}

// 64-bit stream key of one (forward, row, layer): k0 offsets the Weyl sequence, k1 enters between the two multiply rounds
// (folded into a v_mad_u32_u24, i.e. free), so two streams are not windows of one 2^32-periodic sequence.
struct RngKey {
    uint32_t k0, k1;
};

// keep word of element pair `pair_index` of a row: Weyl sequence (one add per consecutive pair after strength reduction)
// through two xorshift-multiply rounds built on 24-bit multiplies (the conv epilogues spend most of their VALU time here;
// the murmur3 finaliser costs two quarter-rate multiplies per pair).
// (split for kernels that walk consecutive pairs: the Weyl value of pair i + d is that of pair i plus the constant
// d * RNG_WEYL, so a loop carries it with one add instead of a quarter-rate 32-bit multiply per pair)
constexpr uint32_t RNG_WEYL = 0x9E3779B1u;
__device__ __host__ __forceinline__ uint32_t rng_weyl(uint32_t pair_index, RngKey key) { return pair_index * RNG_WEYL + key.k0; }
__device__ __host__ __forceinline__ uint32_t rng_pair_mix(uint32_t x, RngKey key) {
    x ^= x >> 15;
    x = mul_u24(x, 0x735A2Du) + key.k1;
    x ^= x >> 13;
    x = mul_u24(x, 0x97E5B5u);
    x ^= x >> 16;
    return x;
}
__device__ __host__ __forceinline__ uint32_t rng_pair_word(uint32_t pair_index, RngKey key) {
    uint32_t x = pair_index * 0x9E3779B1u + key.k0;
    x ^= x >> 15;
    x = mul_u24(x, 0x735A2Du) + key.k1;
    x ^= x >> 13;
    x = mul_u24(x, 0x97E5B5u);
    x ^= x >> 16;
    return x;
}

// per-(forward, global row) key pair; written to the engine's row-key table by rng_begin_forward_kernel (kernels.hip)
// before every forward that draws masks
__device__ __host__ __forceinline__ RngKey rng_row_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t fwd, uint32_t grow) {
    const uint32_t a = fmix32(seed_lo ^ fmix32(seed_hi + 0x9E3779B9u * (fwd + 1u)));
    const uint32_t b = fmix32(a + 0x85EBCA77u * (grow + 1u));
    return RngKey{b, fmix32(b ^ seed_hi ^ 0x27D4EB2Fu)};
}

// per-layer salt pair (host: make_drop), xor-ed / added into the row key
__device__ __host__ __forceinline__ RngKey rng_layer_salt(uint32_t layer) {
    const uint32_t s = fmix32(0x9E3779B9u * (layer + 1u));
    return RngKey{s, fmix32(s + 0x165667B1u)};
}

__device__ __host__ __forceinline__ RngKey rng_stream_key(RngKey row, RngKey salt) {
    return RngKey{row.k0 ^ salt.k0, row.k1 + salt.k1};
}

__device__ __host__ __forceinline__ uint32_t keep_threshold16(float p) {
    float keep = (1.0f - p) * 65536.0f;
    return keep >= 65536.0f ? 65536u : (uint32_t)keep;
}

// QUAD form (round 5: the softmax probabilities of Attention, attention.py:70 -- N^2 elements per head, where the keep-word hash was
// 2.7x the rest of the softmax's vector work): one 32-bit word per FOUR consecutive elements, 8 bits each against an 8-bit
// threshold.  The keep probability is then k / 256 with k = floor((1 - p) * 256) (>= 1) -- within 2^-8 of 1 - p -- and the survivors are
// scaled by 256 / k instead of 1 / (1 - p): E[mask * scale] = 1 EXACTLY, as for nn.Dropout; what differs from the reference is the
// drop rate itself by < 0.4 % absolute (p = 0.1: 0.1016), invisible next to the Monte-Carlo spread of the ensemble it generates.
// Every other dropout site keeps the 16-bit pair form below (threshold resolution 2^-16).
__device__ __host__ __forceinline__ uint32_t keep_threshold8(float p) {
    const float keep = (1.0f - p) * 256.0f;
    const uint32_t k = keep >= 256.0f ? 256u : (uint32_t)keep;
    return k < 1u ? 1u : k;
}
__device__ __host__ __forceinline__ uint32_t rng_quad_word(uint32_t quad_index, RngKey key) { return rng_pair_word(quad_index, key); }

// element e of the row's NHWC tensor: keep iff its 16-bit slice of pair word e>>1 is below the threshold
__device__ __host__ __forceinline__ bool rng_keep(uint32_t e, RngKey key, uint32_t thresh16) {
    uint32_t w = rng_pair_word(e >> 1, key);
    uint32_t v = (e & 1u) ? (w >> 16) : (w & 0xffffu);
    return v < thresh16;
}

__device__ __host__ __forceinline__ bool rng_keep8(uint32_t e, RngKey key, uint32_t thresh8) {
    const uint32_t w = rng_quad_word(e >> 2, key);
    return ((w >> (8u * (e & 3u))) & 0xffu) < thresh8;
}

// Dropout site id (rng_layer_salt argument) of unet_simple's dropout_input; the 12 UNetBlocks are sites 0..11
#define DYF_INPUT_DROP_SITE 12u

// Dropout descriptor handed to every kernel that ends in a Dropout layer.
struct DropSpec {
    int mode;                 // 0 off, 1 engine RNG, 2 injected mask
    float scale;              // 1/(1-p)
    uint32_t thresh16;        // keep threshold (mode 1)
    RngKey salt;              // rng_layer_salt(layer slot) (mode 1)
    const uint32_t* row_keys; // device [launch rows][2]: rng_row_key of every row of this launch (mode 1)
    const uint8_t* mask;      // device NHWC uint8 keep mask of the whole launch (mode 2)
    uint32_t thresh8;         // quad form (attention probabilities, mode 1): keep threshold k of 256 and the matching scale 256 / k
    float scale8;
};

// stream key of launch row `n` (batch row inside this launch)
__device__ __forceinline__ RngKey drop_row_key(const DropSpec& d, int n) {
    if (d.mode != 1) return RngKey{0u, 0u};
    const uint2 rk = *(const uint2*)(d.row_keys + 2 * (size_t)n);
    return rng_stream_key(RngKey{rk.x, rk.y}, d.salt);
}

// e: element index in the whole launch tensor (mask injection); row0: index of the row's first element (engine RNG keys on
// e - row0)
__device__ __forceinline__ float drop_apply(float v, uint32_t e, uint32_t row0, const DropSpec& d, RngKey key) {
    if (d.mode == 0) return v;
    bool keep = d.mode == 1 ? rng_keep(e - row0, key, d.thresh16) : (d.mask[e] != 0);
    return keep ? v * d.scale : 0.0f;
}

// Epilogue form: activation + dropout over N (even) consecutive elements starting at the EVEN element index e0.  The
// (activation, dropout mode) pair is wave-uniform, so it is dispatched ONCE per group with scalar branches and each
// variant is a straight-line loop (the per-element form above is if-converted by the compiler into ~25 VALU/element
// with the SiLU exp/rcp always evaluated); the keep word is hashed once per PAIR.
template <int ACT>
__device__ __forceinline__ float act_fixed(float v) {
    if constexpr (ACT == ACT_RELU) return fmaxf(v, 0.0f);
    else if constexpr (ACT == ACT_LEAKY) return fmaxf(v, 0.2f * v);
    else if constexpr (ACT == ACT_SILU) return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));  // 1-ulp rcp, no IEEE divide
    else if constexpr (ACT == ACT_GELU) return 0.5f * v * (1.0f + erff(v * 0.70710678118654752f));
    else return v;
}

// The dropout scale 1/(1-p) commutes with the positively homogeneous activations (none, ReLU, LeakyReLU): epilogues fold
// it into their affine coefficients (drop_prescale) and the dropout itself is a select.  SiLU keeps the multiply.
template <int ACT, int MODE>
__device__ __forceinline__ float drop_prescale(const DropSpec& d) {
    return (MODE != 0 && (ACT == ACT_NONE || ACT == ACT_RELU || ACT == ACT_LEAKY)) ? d.scale : 1.0f;
}

template <int N, int ACT, int MODE, bool PRESCALED = false>
__device__ __forceinline__ void act_drop_fixed(float* v, uint32_t e0, uint32_t row0, const DropSpec& d, RngKey key) {
#pragma unroll
    for (int t = 0; t < N; ++t) v[t] = act_fixed<ACT>(v[t]);
    constexpr bool folded = PRESCALED && (ACT == ACT_NONE || ACT == ACT_RELU || ACT == ACT_LEAKY);
    if constexpr (MODE == 1) {
        const uint32_t th = d.thresh16;
        const float sc = folded ? 1.0f : d.scale;
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            const uint32_t w = rng_pair_word(((e0 - row0) >> 1) + p, key);
            if constexpr (folded) {
                v[2 * p] = (w & 0xffffu) < th ? v[2 * p] : 0.0f;
                v[2 * p + 1] = (w >> 16) < th ? v[2 * p + 1] : 0.0f;
            } else {
                v[2 * p] = (w & 0xffffu) < th ? v[2 * p] * sc : 0.0f;
                v[2 * p + 1] = (w >> 16) < th ? v[2 * p + 1] * sc : 0.0f;
            }
        }
    } else if constexpr (MODE == 2) {
#pragma unroll
        for (int t = 0; t < N; ++t) v[t] = d.mask[e0 + t] != 0 ? (folded ? v[t] : v[t] * d.scale) : 0.0f;
    }
}

template <int N, int ACT>
__device__ __forceinline__ void act_drop_mode(float* v, uint32_t e0, uint32_t row0, const DropSpec& d, RngKey key) {
    if (d.mode == 0) act_drop_fixed<N, ACT, 0>(v, e0, row0, d, key);
    else if (d.mode == 1) act_drop_fixed<N, ACT, 1>(v, e0, row0, d, key);
    else act_drop_fixed<N, ACT, 2>(v, e0, row0, d, key);
}

template <int N>
__device__ __forceinline__ void act_drop(float* v, uint32_t e0, uint32_t row0, int act, const DropSpec& d, RngKey key) {
    if (act == ACT_RELU) act_drop_mode<N, ACT_RELU>(v, e0, row0, d, key);
    else if (act == ACT_LEAKY) act_drop_mode<N, ACT_LEAKY>(v, e0, row0, d, key);
    else if (act == ACT_SILU) act_drop_mode<N, ACT_SILU>(v, e0, row0, d, key);
    else act_drop_mode<N, ACT_NONE>(v, e0, row0, d, key);
}

// ------------------------------------------------------------------------------------------------ bilinear
// align_corners=False source coordinate (ATen area_pixel_compute_source_index); oracle: nets.bilinear_resize_explicit
// nearest = true: F.interpolate(mode="nearest") (ATen nearest_neighbor_compute_source_index: floor(dst * scale), clamped) as
// the degenerate stencil i0 = i1, lam = 0 -- every consumer of the two-tap form then also evaluates outer_sample_mode="nearest"
__device__ __forceinline__ void bilinear_coord(int dst, float scale, int in_size, int& i0, int& i1, float& lam, bool nearest = false) {
    if (nearest) {
        i0 = min((int)floorf((float)dst * scale), in_size - 1);
        i1 = i0;
        lam = 0.0f;
        return;
    }
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
    lam = src - (float)i0;
}
