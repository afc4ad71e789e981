// Conv2d (+ fused epilogue) on NHWC bf16 activations: argument block shared by the direct and the MFMA kernels.
#pragma once
#include "common.h"
#include "gn_fused.h"

#include <string>
#include <vector>

// kernel-form log (common.h dyf_form_note; test seam dyf_debug_form_log*)
void dyf_form_log_enable(bool on);
std::string dyf_form_log_text();

struct ConvArgs {
    // input: channel-concatenation of up to two NHWC bf16 tensors (skip connections are never materialised)
    const el16_t* src0;
    const el16_t* src1;
    int c0, c1;          // channels taken from src0 / src1 (c1 == 0: single source)
    int pix_pitch0;      // elements between consecutive pixels of src0 (0: = c0).  The fused stem feeds enc0 a 16-channel
                         // tensor and declares c0 = 64: one K chunk then spans 4 horizontally adjacent pixels
    int n, h, w;         // input batch / height / width
    int ho, wo;          // output height / width
    int kh, kw, stride, pad;
    int cout;
    const el16_t* wpk;   // packed weights [cout][kh*kw][c0+c1] bf16
    const el16_t* wpk_frag;  // the same weights in MFMA fragment order (pack_conv_frag) for conv_igemm2, or null: looked up
                             // in the registry (conv_register_frag) by launch_conv
    // fused x2 bilinear upsample in front of a 3x3/s1/p1 conv: sources are the LOW-res tensors (h, w), output is
    // (2h, 2w); wpk_up holds the phase-decomposed weights [4][cout][16][c0+c1] (pack_up2x_weights)
    int up2x;
    const el16_t* wpk_up;
    const el16_t* wpk_up_frag;  // the same weights in MFMA fragment order (pack_up2x_frag) for the halo kernel, or null
    // sparse output columns of the halo kernel (plan_up_sparse_columns): device lists [2][up_npad] of low-res columns per
    // horizontal phase, halo origin per list tile; null = every column
    const int16_t* up_cols;
    const int16_t* up_cbase;
    const int16_t* up_cidx;  // [2][up_npad]: column of each list entry's output pixel in the COMPACT output tensor
    int up_ntiles, up_npad, up_nvalid0, up_nvalid1;
    // mixed list tiling of conv_halo_rows_kernel<1, SH> (plan_up_sparse_columns_mixed): up_mix[s] list tiles of shape s per row
    // block of that shape (32 entries x 4 rows | 16 x 8 | 4 x 32), laid out [32-entry tiles | 16-entry tiles | 4-entry tiles] in
    // up_cols / up_cidx (up_npad = 32 a + 16 b + 4 c) and up_cbase (a + b + c entries); all zero = uniform 32- (or 16-) slot tiles
    int up_mix[3];
    int up_sh_off, up_sh_cb;  // set per launch by launch_conv_halo_rows_up: first list entry / first cbase entry of the launch's shape
    int up_wo_store;         // columns of the compact output tensor [n][ho][up_wo_store][cout]
    // epilogue: v = acc * A[row*coef_stride + co] + C[row*coef_stride + co]; row = sample index (coef_stride may be 0
    // to broadcast one row); conv bias, eval-BatchNorm and FiLM (x*(scale+1)+shift) are all folded into A and C.
    const float* coef_a;
    const float* coef_c;
    int coef_stride;
    int coef_div;        // samples per coefficient row (0/1: one row per sample; paired interpolator calls use nb)
    int act;
    DropSpec drop;
    // nearest x2 upsample in front of a plain 3x3 conv (unet.py Upsample = nn.Upsample(scale_factor=2, mode="nearest") + Conv2d), fused:
    // up_nearest != 0 -> src0 is the LOW-resolution tensor (n, h / 2, w / 2, c0) and (h, w) stay the conv's (upsampled) input size; the
    // halo gather of conv_up_halo_kernel<5> reads pixel (y >> 1, x >> 1).  Only that form understands it: set it only when
    // conv_plain3x3_takes_halo5() says the launch goes there (the caller materialises the upsample otherwise)
    int up_nearest;
    const el16_t* residual;  // NHWC bf16 tensor added after activation/dropout (Residual(...) wrappers), or null
    el16_t* out_el16;    // NHWC bf16 output (or null)
    float* out_f32;      // NHWC fp32 output (GroupNorm input) (or null)
    const el16_t* zero_page;  // >= 128 B of zeros in HBM: source of padded taps for the LDS-DMA gather
    // halo form of the fused-upsample conv: fp32 scratch [n][2*wo + 2*(ho-2)][cout] for the border corrections
    // (up_border_kernel fills it, conv_up_halo_kernel starts its border accumulators from it); required by that form
    float* up_border;
    // split-K of conv_igemm_kernel<128,128> at small batches: fp32 scratch [splitk][n*ho*wo][cout] (engine workspace; null =
    // never split), its capacity in floats, and the split factor (set by the launcher)
    float* splitk_ws;
    long long splitk_cap;
    int splitk;
    int n_sel;          // batch the kernel FORM is selected for (tile-count thresholds, split-K); 0 = this launch's n
    // GroupNorm statistics of the OUTPUT from the fp32 accumulators (the conv feeds a GroupNorm: unet.py:58-76): when non-null,
    // a kernel form that supports it (conv_up_halo_kernel<5>; launch_conv_stats reports whether the launch did) writes, per
    // sample, `gn_slots` partial (sum, sum of squares) pairs of every 8-channel octet of y = acc * A + C over the pixels of one
    // (tile, wave): gn_part[((n * gn_slots + slot) * (cout / 8) + octet) * 2 + {0, 1}] fp32.  Fixed slots, no atomics: the
    // consumer (gn_apply_part_kernel) adds them in slot order, so results are reproducible run to run.
    float* gn_part;
    int gn_slots;       // set by the launcher
    // GroupNorm FUSED into this conv's epilogue (gn_fused.h; gnf.gran != null): y = conv + bias is normalised per (sample, group),
    // FiLM / `act` (SiLU) / `drop` / `residual` applied, and the finished activation stored -- only through launch_conv_gn_fused,
    // which reports whether a kernel form that can do it took the launch (coef_a / coef_c are then unused)
    GnFuse gnf;
};
// floats of ConvArgs::up_border for an n x (2h x 2w) x cout output
inline size_t conv_up_border_floats(int n, int h, int w, int cout) { return (size_t)n * (4 * (size_t)w + 4 * (size_t)h - 4) * cout; }

// path: 0 direct (any shape), 1 implicit-GEMM MFMA (needs c0 % 64 == 0, c1 % 64 == 0, cout % 64 == 0)
hipError_t conv_init();
bool conv_mfma_supported(const ConvArgs& a);
hipError_t launch_conv(const ConvArgs& a, int path, hipStream_t stream);
// launch_conv for a conv whose output feeds a GroupNorm (a.gn_part != null): *gn_slots receives the partial-sum slots per sample
// the launch wrote (0: this kernel form does not produce statistics -- the caller runs the statistics pass)
hipError_t launch_conv_stats(const ConvArgs& a, int path, hipStream_t stream, int* gn_slots);
// launch_conv for a conv followed by GroupNorm + FiLM + SiLU + Dropout (+ residual), a.gnf filled in except `slots`: *fused = the
// launch did all of it (conv_up_halo_kernel<5, 2>, conv_igemm2_kernel<2, true>); false = NOTHING was launched (the shape / batch is
// not served by a fused form: the caller runs the conv and the GroupNorm kernels)
hipError_t launch_conv_gn_fused(const ConvArgs& a, int path, hipStream_t stream, bool* fused);
int conv_gn_fused_max_slots(int h, int w);
bool conv_igemm2_tile2d(int ho, int wo);  // conv_igemm2_kernel tiles this output plane 2-D (TH x 16 pixel tiles inside one sample)
int conv_igemm2_gn_slots_bm128(int ho, int wo);  // ... of its 128-pixel tile form (64-row statistics slabs)
int conv_igemm2_gn_slots(int ho, int wo);  // slots per sample of the fused conv_igemm2_kernel<2> on an ho x wo output plane (0: not served)  // upper bound of GnFuse::slots on an h x w plane (sizing of GnFuse::gran), 0 = never fused
int conv_halo5_gn_slots(int h, int w);  // slots per sample of conv_up_halo_kernel<5> on an h x w plane (sizing of gn_part)
void pack_up2x_weights(const float* w, int cout, int cin, el16_t* out);
// halo form of the fused x2-upsample conv (conv_up_halo.hip): 16x16 low-res tile x 4 phases x 64 channels per workgroup
bool conv_up_halo_supported(const ConvArgs& a);
void pack_up2x_frag(const el16_t* wpk_up, int cout, int cin, el16_t* out);
bool plan_up_sparse_columns(const std::vector<uint8_t>& needed, int w, std::vector<int16_t>& cols, std::vector<int16_t>& cbase,
                            std::vector<int16_t>& cidx, std::vector<int16_t>& col_map, int& ntiles, int& nvalid0, int& nvalid1,
                            int slots = 16);
// rows form of the halo kernels (conv_halo_rows.hip): 4 x 32 low-res tiles, one-row MFMA pixel tiles
int conv_halo_rows_slots();
int conv_halo_rows_sparse_halo_w();
int conv_halo_rows_sparse_halo_w_shape(int shape);  // halo width of list-tile shape 0 / 1 / 2 (32 x 4 rows | 16 x 8 | 4 x 32)
// Mixed list tiling for conv_halo_rows_kernel<1, SH>: the per-phase lists are cut into a 32-entry, b 16-entry and c 4-entry tiles
// (mix = {a, b, c}) so that no MFMA lane is padding when the lists allow it (52 entries = 3 x 16 + 4); h = low-res rows (the 16- /
// 4-entry shapes own 8 / 32 rows).  False when a tile does not fit its shape's halo or the rows do not divide.
bool plan_up_sparse_columns_mixed(const std::vector<uint8_t>& needed, int w, int h, std::vector<int16_t>& cols, std::vector<int16_t>& cbase,
                                  std::vector<int16_t>& cidx, std::vector<int16_t>& col_map, int mix[3], int& nvalid0, int& nvalid1);
bool conv_halo_rows_up_supported(const ConvArgs& a);   // in addition to conv_up_halo_supported
bool conv_halo_rows3_supported(const ConvArgs& a);     // in addition to conv_halo3_supported (h % 8 / w % 16 not needed)
hipError_t conv_halo_rows_init();
// sum of ConvArgs::splitk raw fp32 partials [split][m][cout] (in split order) + the conv epilogue, 4 channels per thread (conv.hip);
// needs cout % 4 == 0
hipError_t launch_conv_splitk_finish4(const ConvArgs& a, long long M, hipStream_t stream);
hipError_t launch_conv_halo_rows_up(const ConvArgs& a, hipStream_t stream);  // main kernel only (after up_border_kernel)
hipError_t launch_conv_halo_rows3(const ConvArgs& a, hipStream_t stream);
hipError_t conv_up_halo_init();
hipError_t launch_conv_up_halo(const ConvArgs& a, hipStream_t stream);
// plain 3x3 / stride 1 / pad 1 conv on the halo kernel (conv_up_halo.hip, SP = 2): cout % 256 == 0, h % 8 == 0, w % 16 == 0;
// ConvArgs::wpk_up_frag carries the pack_halo3_frag weights (looked up in the registry by launch_conv)
bool conv_halo3_supported(const ConvArgs& a);
hipError_t launch_conv_halo3(const ConvArgs& a, hipStream_t stream);
void pack_halo3_frag(const el16_t* wpk, int cout, int cin, el16_t* out);
// 4x4 / stride 2 / pad 1 conv on the halo kernel (SP = 3, space-to-depth view): cout % 256 == 0, single source, output plane
// tiles by 8x16; fragments (pack_halo_s2_frag) share the halo3 registry
bool conv_halo_s2_supported(const ConvArgs& a);
hipError_t launch_conv_halo_s2(const ConvArgs& a, hipStream_t stream);
void pack_halo_s2_frag(const el16_t* wpk, int cout, int cin, el16_t* out);
// plain 3x3 / s1 / p1 conv with cout % 64 == 0 on ANY plane size (SP = 5: four pixel sub-tiles per workgroup, ragged edges)
bool conv_halo5_supported(const ConvArgs& a);
// launch_conv(a, 1, ...) of this plain 3x3 conv (no statistics, no fused GroupNorm) will run on conv_up_halo_kernel<5> -- the one form
// that can take ConvArgs::up_nearest
bool conv_plain3x3_takes_halo5(const ConvArgs& a);
hipError_t launch_conv_halo5(const ConvArgs& a, hipStream_t stream);
void pack_halo3_frag64(const el16_t* wpk, int cout, int cin, el16_t* out);
// the same convs WITH the fused GroupNorm epilogue on 16 x 16-pixel tiles, three workgroups per CU (conv_gn16.hip); fragments of
// pack_halo3_frag64; ONE statistics slot per workgroup
bool conv_gn16_supported(const ConvArgs& a);
int conv_gn16_slots(int h, int w);
hipError_t launch_conv_gn16(const ConvArgs& a, hipStream_t stream);
void conv_register_halo3_frag(const el16_t* wpk_dev, const el16_t* frag_dev);
const el16_t* conv_lookup_halo3_frag(const el16_t* wpk_dev);
// 3 x 3 convs with cout % 256 == 0 keep the 256-channel-block fragments (pack_halo3_frag) in the halo3 registry; their 64-channel-block
// copy (pack_halo3_frag64, what conv_gn16_kernel streams) lives here
void conv_register_frag64(const el16_t* wpk_dev, const el16_t* frag_dev);
const el16_t* conv_lookup_frag64(const el16_t* wpk_dev);
// enc0 on the fused stem (conv_enc0_stem.hip): persistent, weights resident in LDS, pixel fragments straight from global memory;
// the fragments (pack_enc0_stem_frag) are registered in the halo3 registry under the composed weights' pointer
bool conv_enc0_stem_supported(const ConvArgs& a);
hipError_t conv_enc0_stem_init();
hipError_t launch_conv_enc0_stem(const ConvArgs& a, const el16_t* wfrag, hipStream_t stream);
void pack_enc0_stem_frag(const el16_t* wpk, int cout, el16_t* out);
// second implicit-GEMM form (conv_igemm2.hip): 256 px x 128 ch per workgroup, weights streamed in fragment order
bool conv_igemm2_supported(const ConvArgs& a);
hipError_t conv_igemm2_init();
hipError_t launch_conv_igemm2(const ConvArgs& a, hipStream_t stream);
void pack_conv_frag(const el16_t* wpk, int cout, int taps, int cin, el16_t* out);
// small-M 1x1 / 2x2-s2 convs (conv_skinny.hip): 32 x 32 tiles, K split over the workgroup's four waves, one launch (replaces
// conv_igemm_kernel<128,128> + split-K + conv_splitk_finish_kernel in the few-rows regime); fragments: pack_conv_frag
bool conv_skinny_supported(const ConvArgs& a);
hipError_t launch_conv_skinny(const ConvArgs& a, hipStream_t stream);
// registry device-pointer(wpk) -> device-pointer(fragment-ordered copy); filled when weights are uploaded
void conv_register_frag(const el16_t* wpk_dev, const el16_t* frag_dev);
void conv_unregister_frag(const void* wpk_dev);
const el16_t* conv_lookup_frag(const el16_t* wpk_dev);
