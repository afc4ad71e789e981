/*
 * dyffusion_hip_testing.h -- test / benchmark seams of libdyffusion_hip.so.  NOT part of the drop-in boundary
 * (include/dyffusion_hip.h): op-level entry points for the kernel parity tests, per-layer timers for bench.py's roofline
 * object, and activation read-back for per-layer parity analysis.  Same conventions as the public header.
 */
#ifndef DYFFUSION_HIP_TESTING_H
#define DYFFUSION_HIP_TESTING_H

#include "dyffusion_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Time the dominant conv kernel: average HIP-event duration (ms) of the conv layer `layer` (0..11, encoder then
 * decoder blocks) of `net` at batch nb over `iters` launches on `stream`; also returns its 2*MAC count. */
dyf_status dyf_time_conv_layer(dyf_engine* engine, int32_t net, int32_t layer, int32_t nb, int32_t iters,
                               void* stream, double* avg_ms, double* flops, double* algorithmic_bytes);

/* Benchmark introspection: average duration (HIP events on `stream`) of the conv launch of decoder block `layer` (6..11)
 * over ONE eagerly launched rollout of the current plan (all forecaster + interpolator forwards, MC dropout as configured),
 * re-using the inputs of the last dyf_sample call.  `launches` receives the number of launches averaged. */
dyf_status dyf_time_layer_in_rollout(dyf_engine* engine, int32_t layer, int32_t nb, void* stream, double* avg_ms,
                                     int32_t* launches);

/* The same for a ResNet-UNet pair (arch unet.Unet: BASELINE configs[2] OISST, configs[4] 512^2): kind 0 = the 3x3 weight-
 * standardised convs of the full-resolution level with cin == cout == dim (the largest share of an OISST forward), 1 = the
 * bottleneck Attention core (flash kernel), 2 = the GroupNorm(+FiLM+SiLU+dropout(+residual)) chain of the full-resolution level
 * (dim channels; all kernels of one Block's normalisation count as ONE launch).  avg_ms is per launch over nb rows; flops = 2*MAC
 * of one launch (0 for kind 2), algorithmic_bytes = 16-bit operands once each (kind 2: one read + one write of the tensor). */
dyf_status dyf_time_kernel_in_rollout(dyf_engine* engine, int32_t kind, int32_t nb, void* stream, double* avg_ms,
                                      int32_t* launches, double* flops, double* algorithmic_bytes);

/* The HBM-bound kernels (bench.py `hbm_kernels`; north_star: "HBM GB/s for the norm/activation kernels"): over ONE eagerly launched
 * rollout of the current plan at nb rows, HIP events on `stream` around every launch of the kernel called `kernel` --
 * "layernorm_c_vec_kernel", "up2x_quad_kernel", "up2x_epilogue_kernel", "stem16_rows_kernel", "readout_dma_kernel",
 * "up2x_nearest_vec_kernel", "gn_apply_walk_kernel", "gn_apply_part_kernel" (the launchers that open a KernelProf scope, csrc/common.h).
 * total_ms = sum of the launch durations, total_bytes = sum of their ALGORITHMIC bytes (every operand once, 16-bit activations),
 * launches = how many (0: the rollout does not launch that kernel).  GB/s = total_bytes / total_ms / 1e6. */
dyf_status dyf_time_named_kernel_in_rollout(dyf_engine* engine, const char* kernel, int32_t nb, void* stream, double* total_ms,
                                            int32_t* launches, double* total_bytes);

/* ---- op-level seam (tests only): one Conv2d + fused epilogue on NHWC bf16 tensors ---------------------------- */
/* x_dev (N,H,W,Cin) bf16 bits; w (Cout,Cin,kh,kw) host fp32; scale/shift (N,Cout) device fp32 or NULL;
 * y_dev (N,Ho,Wo,Cout) bf16 bits.  act: 0 none, 1 relu, 2 leaky(0.2).  path: 0 direct, 1 MFMA implicit GEMM. */
dyf_status dyf_op_conv2d(dyf_engine* engine, const uint16_t* x_dev, const float* w_host, int32_t n, int32_t h,
                         int32_t w, int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                         const float* scale_dev, const float* shift_dev, int32_t act, int32_t path, uint16_t* y_dev,
                         void* stream);

/* Upsample(x2, bilinear, align_corners=False) + Conv2d(3x3, pad 1) + epilogue in one kernel (phase-decomposed MFMA
 * implicit GEMM; unet_simple.py:40-52).  x_dev (N,H,W,Cin) bf16 -> y_dev (N,2H,2W,Cout) bf16. */
dyf_status dyf_op_upconv2d(dyf_engine* engine, const uint16_t* x_dev, const float* w_host, int32_t n, int32_t h,
                           int32_t w, int32_t cin, int32_t cout, const float* scale_dev, const float* shift_dev,
                           int32_t act, uint16_t* y_dev, void* stream);

/* LinearAttention core (attention.py:28-49, 4 heads of 32 channels): qkv_dev (N,HW,384) bf16 = to_qkv output ->
 * out_dev (N,HW,128) bf16 = softmax_d(q)*scale . (softmax_n(k) . v^T / HW), the input of to_out.  Runs the pixel-parallel
 * MFMA kernels the ResNet-UNet uses. */
dyf_status dyf_op_linear_attention(dyf_engine* engine, const uint16_t* qkv_dev, int32_t n, int32_t hw, uint16_t* out_dev,
                                   void* stream);

/* The fused LinearAttention block the ResNet-UNet runs for dim 64 / 128 (csrc/unet_kernels.hip linattn_fused_*): xn_dev
 * (N,HW,C) 16-bit = the PreNorm LayerNorm output, xres_dev (N,HW,C) = the block input (residual); wqkv_host (384,C),
 * wout_host (C,128), bout_host (C) fp32 = to_qkv / to_out parameters -> y_dev (N,HW,C) = to_out(core(to_qkv(xn))) + xres.
 * The to_qkv and core outputs are never written to memory. */
dyf_status dyf_op_linear_attention_fused(dyf_engine* engine, const uint16_t* xn_dev, const uint16_t* xres_dev, int32_t n,
                                         int32_t hw, int32_t c, const float* wqkv_host, const float* wout_host,
                                         const float* bout_host, uint16_t* y_dev, void* stream);

/* Attention core (attention.py:62-72, 4 heads of 32 channels, no dropout): qkv_dev (N,HW,384) 16-bit = to_qkv output ->
 * out_dev (N,HW,128) = softmax_j(q_i . k_j / sqrt(32)) . v_j, the input of to_out.  Runs the MFMA flash kernel the bottleneck
 * of the ResNet-UNet uses (HW up to 65 535: 16 384 tokens for the 512^2 synthetic configuration). */
dyf_status dyf_op_attention(dyf_engine* engine, const uint16_t* qkv_dev, int32_t n, int32_t hw, uint16_t* out_dev,
                            void* stream);
/* ... with nn.Dropout(p) on the softmax probabilities (attention.py:70), masks from the engine's generator: the form the
 * interpolator's bottleneck runs under MC dropout (n <= 2 max_batch rows). */
dyf_status dyf_op_attention_dropout(dyf_engine* engine, const uint16_t* qkv_dev, int32_t n, int32_t hw, float p, uint16_t* out_dev,
                                    void* stream);

/* One training convolution (csrc/train_gemm.hip: fp32 matrix-core forward / dgrad / wgrad of nn.Conv2d on NHWC fp32 tensors) on
 * hash-random data against the plain VALU kernel of csrc/train.hip.  kind 0 forward, 1 data gradient, 2 weight gradient (the
 * matrix-core launchers of train_gemm.hip / train_halo16.hip), 3 weight gradient and 4 forward through the step's own dispatchers
 * (conv_wgrad / conv_fwd: reaches the small-channel forms); with 16-bit operands selected the inputs are rounded to 16 bit first; geometry
 * as nn.Conv2d(cin, cout, k, stride s, padding p) on (n, h, w).  out_host[0] = max |mfma - valu| / max |valu| with the split-K
 * workspace, [1] = the same for the unsplit launch, [2] = 1 if the matrix-core form took the shape. */
dyf_status dyf_train_conv_check(dyf_engine* engine, int32_t kind, int32_t n, int32_t h, int32_t w, int32_t cin, int32_t cout,
                                int32_t k, int32_t s, int32_t p, uint32_t seed, float* out_host);

/* Read back the output of UNetBlock `layer` (0..11: encoder then decoder blocks) of the most recent unet_simple forward
 * as fp32 NCHW (NB, cout, h, w) -- per-layer parity analysis against the oracle's taps (oracle/nets.py `taps=`).  The last
 * decoder block is returned dense; positions its sparse-column form did not compute are NaN. */
dyf_status dyf_debug_read_block_output(dyf_engine* engine, int32_t net, int32_t layer, int32_t nb, float* out_dev,
                                       void* stream);

/* Fused GroupNorm (csrc/gn_fused.h) failure drill: timeout_ticks = bound of a granule sweep in 100 MHz ticks (0 = the default, 2 s);
 * force_timeout != 0 makes every sweep wait for a tag nobody publishes, i.e. behave as if a workgroup of its sample never arrived.
 * Drops the captured graphs (both values travel as kernel arguments) after waiting for the device. */
dyf_status dyf_debug_gn_fuse(dyf_engine* engine, uint32_t timeout_ticks, int32_t force_timeout);

/* Kernel-form log: which kernel FORM every launcher took (conv_halo_rows_kernel<0/1/2>, conv_up_halo_kernel<3/4/5>,
 * conv_igemm2_kernel<1/2>, conv_igemm_kernel<..>(+splitk), conv_enc0_stem_kernel, stem16_rows_kernel, up2x_quad_kernel,
 * readout_dma_kernel, ...) and for how many batch rows.  Forms are chosen per launch from tile counts (csrc/conv.hip
 * launch_conv), so a parity test at NB = 1 does not exercise the kernels a NB = 80 benchmark runs: tests enable the log, run the
 * engine (a hipGraph capture notes its launches once; replays launch nothing on the host) and assert the forms.  Process-wide.
 * dyf_debug_form_log(1) clears and enables, (0) disables; dyf_debug_form_log_read writes "form@rows=count;..." (sorted by name),
 * NUL-terminated and truncated to cap bytes, and returns the untruncated length. */
void dyf_debug_form_log(int32_t enable);
int32_t dyf_debug_form_log_read(char* buf, int32_t cap);

/* Kernel-form switches (tests/ and tools/ only).  The launchers choose between equivalent kernel forms from tile counts; the
 * parity tests force each form on small problems, the A/B tools flip one form at a time, a few keys are wrong-results timing
 * probes.  This call is the ONLY way to set them: libdyffusion_hip.so reads none of them from the environment (the only
 * environment variables it reads are DYF_VERBOSE -- print the engine's form policy at creation -- and DYF_RCCL_LIB -- the librccl
 * to dlopen).  Process-wide (per library: the bf16 and the fp16 build each hold their own table); read per launch / per engine
 * creation / per weight upload as DESIGN.md 7.1 lists.  key = the historic switch name ("DYF_IGEMM2_MIN_TILES", ...), value = its
 * text; value NULL removes the key, key NULL removes every key.  dyf_debug_forms writes "key=value;..." like
 * dyf_debug_form_log_read. */
void dyf_debug_set_form(const char* key, const char* value);
int32_t dyf_debug_forms(char* buf, int32_t cap);

#ifdef __cplusplus
}
#endif
#endif /* DYFFUSION_HIP_TESTING_H */
