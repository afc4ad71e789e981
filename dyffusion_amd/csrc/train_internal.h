// Shared between train.hip (the recorded forward / backward of arch unet_simple) and train_gemm.hip (its convolutions on the
// fp32 matrix cores).
#pragma once
#include "common.h"

#include <algorithm>

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
