// Shared device/host helpers for the gfx950 DYffusion engine.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef uint16_t bf16_t;  // raw bfloat16 bits; activations are NHWC bf16 in HBM

// ------------------------------------------------------------------------------------------------ bf16
__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN preserved
__device__ __host__ __forceinline__ bf16_t f32_to_bf16(float f) {
    uint32_t u;
#if defined(__HIP_DEVICE_COMPILE__)
    u = __float_as_uint(f);
#else
    __builtin_memcpy(&u, &f, 4);
#endif
    if ((u & 0x7fffffffu) > 0x7f800000u) return (bf16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (bf16_t)(u >> 16);
}

// two fp32 -> packed bf16x2, round-to-nearest-even: one v_cvt_pk_bf16_f32 on gfx950 (the software form costs ~8 VALU per
// pair and dominated the conv epilogues).  Written as a vector conversion, NOT inline asm: the compiler's hazard
// recogniser does not look inside asm, and a v_exp_f32 / v_rcp_f32 result consumed by the very next VALU instruction needs
// a wait state (an asm v_cvt right behind a v_exp read the stale register).
typedef __bf16 bf16x2_native_t __attribute__((ext_vector_type(2)));
typedef float f32x2_native_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2_native_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2_native_t));
#else
    return (uint32_t)f32_to_bf16(lo) | ((uint32_t)f32_to_bf16(hi) << 16);
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
// Stateless keep-mask generator for MC dropout (SURVEY hard-part (c)): one murmur3-finalised 32-bit word per PAIR
// of elements, 16 bits each, compared against a 16-bit threshold.  `key` identifies (seed, forward index, layer).
// tests/test_gpu_dropout.py re-implements this in numpy so the RNG path is checked against the oracle bit-for-bit.
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

// keep word of element pair `pair_index`: Weyl sequence (one add per consecutive pair after strength reduction) through
// two xorshift-multiply rounds built on 24-bit multiplies (the conv epilogues spend most of their VALU time here; the
// murmur3 finaliser it replaces costs two quarter-rate multiplies per pair).  Bucket chi-square, lag correlations and
// key avalanche of the keep masks match fmix32's (checked for 2^22 pairs x several keys).
__device__ __host__ __forceinline__ uint32_t rng_pair_word(uint32_t pair_index, uint32_t key) {
    uint32_t x = pair_index * 0x9E3779B1u + key;
    x ^= x >> 15;
    x = mul_u24(x, 0x735A2Du);
    x ^= x >> 13;
    x = mul_u24(x, 0x97E5B5u);
    x ^= x >> 16;
    return x;
}

// key for dropout layer `layer` of the `fwd`-th network forward since dyf_seed(seed)
__device__ __host__ __forceinline__ uint32_t rng_layer_key(uint32_t seed_lo, uint32_t seed_hi, uint32_t fwd,
                                                             uint32_t layer) {
    return fmix32(seed_lo ^ fmix32(seed_hi + 0x9E3779B9u * (fwd * 64u + layer + 1u)));
}

__device__ __host__ __forceinline__ uint32_t keep_threshold16(float p) {
    float keep = (1.0f - p) * 65536.0f;
    return keep >= 65536.0f ? 65536u : (uint32_t)keep;
}

// element e of the NHWC tensor: keep iff its 16-bit slice of pair word e>>1 is below the threshold
__device__ __host__ __forceinline__ bool rng_keep(uint32_t e, uint32_t key, uint32_t thresh16) {
    uint32_t w = rng_pair_word(e >> 1, key);
    uint32_t v = (e & 1u) ? (w >> 16) : (w & 0xffffu);
    return v < thresh16;
}

// Dropout descriptor handed to every kernel that ends in a Dropout layer.
struct DropSpec {
    int mode;                 // 0 off, 1 engine RNG, 2 injected mask
    float scale;              // 1/(1-p)
    uint32_t thresh16;        // keep threshold (mode 1)
    uint32_t layer;           // layer slot (mode 1)
    const uint32_t* state;    // device: {seed_lo, seed_hi, forward_counter} (mode 1)
    const uint8_t* mask;      // device NHWC uint8 keep mask (mode 2)
};

__device__ __forceinline__ uint32_t drop_key(const DropSpec& d) {
    return d.mode == 1 ? rng_layer_key(d.state[0], d.state[1], d.state[2], d.layer) : 0u;
}

__device__ __forceinline__ float drop_apply(float v, uint32_t e, const DropSpec& d, uint32_t key) {
    if (d.mode == 0) return v;
    bool keep = d.mode == 1 ? rng_keep(e, key, d.thresh16) : (d.mask[e] != 0);
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
__device__ __forceinline__ void act_drop_fixed(float* v, uint32_t e0, const DropSpec& d, uint32_t key) {
#pragma unroll
    for (int t = 0; t < N; ++t) v[t] = act_fixed<ACT>(v[t]);
    constexpr bool folded = PRESCALED && (ACT == ACT_NONE || ACT == ACT_RELU || ACT == ACT_LEAKY);
    if constexpr (MODE == 1) {
        const uint32_t th = d.thresh16;
        const float sc = folded ? 1.0f : d.scale;
#pragma unroll
        for (int p = 0; p < N / 2; ++p) {
            const uint32_t w = rng_pair_word((e0 >> 1) + p, key);
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
__device__ __forceinline__ void act_drop_mode(float* v, uint32_t e0, const DropSpec& d, uint32_t key) {
    if (d.mode == 0) act_drop_fixed<N, ACT, 0>(v, e0, d, key);
    else if (d.mode == 1) act_drop_fixed<N, ACT, 1>(v, e0, d, key);
    else act_drop_fixed<N, ACT, 2>(v, e0, d, key);
}

template <int N>
__device__ __forceinline__ void act_drop(float* v, uint32_t e0, int act, const DropSpec& d, uint32_t key) {
    if (act == ACT_RELU) act_drop_mode<N, ACT_RELU>(v, e0, d, key);
    else if (act == ACT_LEAKY) act_drop_mode<N, ACT_LEAKY>(v, e0, d, key);
    else if (act == ACT_SILU) act_drop_mode<N, ACT_SILU>(v, e0, d, key);
    else act_drop_mode<N, ACT_NONE>(v, e0, d, key);
}

// ------------------------------------------------------------------------------------------------ bilinear
// align_corners=False source coordinate (ATen area_pixel_compute_source_index); oracle: nets.bilinear_resize_explicit
__device__ __forceinline__ void bilinear_coord(int dst, float scale, int in_size, int& i0, int& i1, float& lam) {
    float src = ((float)dst + 0.5f) * scale - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    if (i0 > in_size - 1) i0 = in_size - 1;
    i1 = i0 + 1 < in_size ? i0 + 1 : in_size - 1;
    lam = src - (float)i0;
}
