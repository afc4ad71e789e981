// Shared between train.hip (the recorded forward / backward of arch unet_simple) and train_gemm.hip (its convolutions on the
// fp32 matrix cores).
#pragma once
#include "common.h"

#include <algorithm>

// ---- the 16-bit operand format of the TRAINING convs: always bf16, in both builds of the library (the fp16 build too).
// Training tensors, master weights, accumulators and statistics are fp32; only the MFMA operands are rounded while they are staged,
// so the format is independent of the engine's inference storage type.  fp16 operands would need the reference's GradScaler
// (Lightning precision=16): with mean-reduced losses dL/dout is ~1e-6..1e-7 at real batch sizes, below fp16's normal range, and the
// rounded gradient operand keeps a few bits or flushes to zero.  bf16 has fp32's exponent: no loss scale, no skipped steps.
typedef __bf16 t16_native_t;
typedef t16_native_t t16x8_t __attribute__((ext_vector_type(8)));
typedef t16_native_t t16x2_native_t __attribute__((ext_vector_type(2)));
typedef uint16_t t16_t;
#define T16_MFMA_32x32x16(a, b, c, x, y, z) __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, x, y, z)
__device__ __host__ __forceinline__ t16_t f32_to_t16(float f) {  // round-to-nearest-even, NaN preserved
    uint32_t u = __builtin_bit_cast(uint32_t, f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (t16_t)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (t16_t)(u >> 16);
}
__device__ __host__ __forceinline__ float t16_to_f32(t16_t v) { return __builtin_bit_cast(float, ((uint32_t)v) << 16); }
__device__ __forceinline__ uint32_t pack_t16x2(float lo, float hi) {  // one v_cvt_pk_bf16_f32
#if defined(__HIP_DEVICE_COMPILE__)
    const f32x2_native_t v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, t16x2_native_t));
#else
    return (uint32_t)f32_to_t16(lo) | ((uint32_t)f32_to_t16(hi) << 16);
#endif
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: the training launchers that need more than the 64 KB
// default raise it once per (kernel, device) -- a process-wide flag left the second GPU of a process at the default, and its first
// launch failed.  Returns false (and the launcher declines the shape) when the attribute cannot be set.
#include <atomic>
template <typename K>
static inline bool train_raise_dynamic_lds(K kernel, int bytes) {
    static std::atomic<unsigned char> done[64];  // per instantiation (= per kernel); 0 not tried, 1 set, 2 failed
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    unsigned char st = done[dev].load(std::memory_order_acquire);
    if (st == 0) {
        st = hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess ? 1 : 2;
        done[dev].store(st, std::memory_order_release);
    }
    return st == 1;
}

namespace dyf {

struct TConv {  // geometry of one nn.Conv2d on NHWC fp32 tensors: x (n, h, w, cin) -> y (n, ho, wo, cout), k x k / stride s / pad p
    int n, h, w, cin, ho, wo, cout, k, s, p;
};

// fp32 MFMA implicit-GEMM forms (train_gemm.hip); return false when the shape is not covered (caller falls back to the VALU kernel)
// ws / ws_floats: workspace for the split-K partial sums of launches with few output tiles (null: never split)
// Operand format of the training convs for the calls of THIS thread: set by the training entry points from the engine's
// dyf_train_set_precision (0 = the DYF_TRAIN_OPERANDS environment variable decides: unset / fp32 -> fp32 operands).
extern thread_local int g_train_precision;
bool train_operands16();
struct TrainPrecisionScope {
    int prev;
    explicit TrainPrecisionScope(int p) : prev(g_train_precision) { g_train_precision = p; }
    ~TrainPrecisionScope() { g_train_precision = prev; }
};

bool tgemm_conv_fwd(const TConv& g, const float* x, const float* wt, const float* bias, float* y, float* ws, size_t ws_floats,
                    hipStream_t st);
bool tgemm_conv_dgrad(const TConv& g, const float* dz, const float* w, const float* bias, float* dx, float* ws, size_t ws_floats,
                      hipStream_t st);
bool tgemm_conv_wgrad(const TConv& g, const float* dz, const float* x, float* dw, hipStream_t st);

// 3 x 3 / stride 1 / pad 1 layers with 16-bit operands, tile + halo staged once (train_halo16.hip); false: shape not covered.
// mode 0: forward (A = x, W = wt[tap][ci][co]); mode 1: data gradient (A = dz, W = w[co][tap][ci]); ws receives the 16-bit weights
bool thalo_conv3x3(const TConv& g, int mode, const float* A, const float* W, const float* bias, float* C, float* ws, size_t ws_floats,
                   hipStream_t st);
bool thalo_wgrad3x3(const TConv& g, const float* dz, const float* x, float* dw, hipStream_t st);

}  // namespace dyf
