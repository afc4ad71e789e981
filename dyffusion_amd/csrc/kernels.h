// HBM-bound kernels around the convolutions: launch prototypes.
#pragma once
#include "common.h"

#define DYF_MAX_IN_CH 32   // max channels entering the network stem (inputs + condition)
#define DYF_MAX_OUT_CH 8   // max output channels of the readout

// K9: sinusoidal features -> Linear -> GELU -> Linear -> SiLU  (misc.py:20-32,54-67; SiLU of every block's time_mlp)
struct TimeMlpArgs {
    const float* time;      // [rows]
    int rows, dim;          // dim = sinusoid width, time_dim = 2*dim
    const float* w1;        // [2dim][dim]
    const float* b1;
    const float* w2;        // [2dim][2dim]
    const float* b2;
    float* silu_out;        // [rows][2dim]
    const float* learned_w = nullptr;  // LearnedSinusoidalPosEmb (misc.py:35-51): `learned_half` frequencies; features
    int learned_half = 0;              // [t, sin(2 pi t w), cos(2 pi t w)], w1 is [2dim][2 learned_half + 1]
};
hipError_t launch_time_mlp(const TimeMlpArgs& a, hipStream_t s);

// K9: per-block FiLM heads fused with the folded norm:  A = a_n*(1+scale), C = c_n*(1+scale)+shift  (all blocks of
// one network in one launch).  With no time embedding scale = shift = 0.
struct FilmArgs {
    const float* silu;      // [rows][tdim] or null
    int rows, tdim, total_c;
    const float* wf;        // [2*total_c][tdim]: per block, scale rows then shift rows (reference Linear layout)
    const float* bf;        // [2*total_c]
    const int* blk_of;      // [total_c] block index of each flattened channel
    const int* blk_off;     // [nblocks] flattened channel offset of each block
    const int* blk_cout;    // [nblocks]
    const float* norm_a;    // [total_c] folded norm scale (1 for the GroupNorm block)
    const float* norm_c;    // [total_c] folded norm shift (0 for the GroupNorm block)
    float* coef_a;          // [rows][total_c]
    float* coef_c;
};
hipError_t launch_film(const FilmArgs& a, hipStream_t s);

// K4+K1: outer bilinear resample of the channel-concatenated NCHW fp32 inputs fused with the 1x1 stem conv
// (unet_simple.py:185-195 upsampler + :166 init_conv) -> NHWC bf16.
struct StemArgs {
    const float* src[4];    // up to 4 NCHW fp32 tensors, concatenated on channels
    int ch[4];
    int nsrc, cin;          // cin = sum(ch)
    int n, h, w;            // native grid
    int src_rows;           // rows held by the source tensors (0: = n); output row r reads source row r % src_rows, so
                            // one launch can run the same inputs under several FiLM rows (paired interpolator calls)
    int uh, uw;             // resampled grid (== h, w when there is no outer resampling)
    int resample;           // 0: identity, 1: resample to (uh, uw)
    int nearest;            // outer_sample_mode: 0 bilinear (align_corners=False), 1 nearest
    const float* wgt;       // [dim][cin] fp32
    const float* bias;      // [dim]
    int dim;
    el16_t* out;            // [n][uh][uw][dim]
    DropSpec drop;          // dropout_input on the 1x1 conv's output (stem_kernel only; mode 0 = off)
    // start of a forward that draws masks, folded into the fused-stem kernels (their block 0 does the work of
    // rng_begin_forward_kernel: nothing in the stem reads the row keys, the first consumer is the next launch): null = not folded
    uint32_t* rng_state;
    uint32_t* rng_row_keys;
    int rng_rows, rng_rows_per_fwd;
};
hipError_t launch_stem(const StemArgs& a, hipStream_t s);
// Fused-stem form: only the outer resample, written as a zero-bordered [n][uh+2][uw+2][16] bf16 tensor whose channel
// `cin` is 1 inside the image (carries init_conv's bias through the composed enc0 weights); the 1x1 conv itself is
// folded into the first encoder block's weights (engine.hip: compose_stem_enc0).
hipError_t launch_stem16(const StemArgs& a, hipStream_t s);

// K2 (materialised form): bilinear x2 upsample of cat[src0, src1] (NHWC bf16) -> NHWC bf16
struct Up2xArgs {
    const el16_t* src0;
    const el16_t* src1;
    int c0, c1;
    int n, h, w;            // low-res dims
    el16_t* out;            // [n][2h][2w][c0+c1]
};
hipError_t launch_up2x(const Up2xArgs& a, hipStream_t s);

// Decoder blocks with a 1 x 1 conv (dec0, dec1 of unet_simple: Upsample(x2, bilinear) -> Conv2d(k = 1) -> BatchNorm -> FiLM -> ReLU ->
// Dropout, unet_simple.py:40-52,72-80): a pointwise conv commutes with the per-channel bilinear upsample (its taps sum to 1), so the
// conv runs on the LOW-res cat[x, skip] -- a quarter of the pixels, no materialised upsample -- into an fp32 tensor, and this pass
// upsamples it and applies the block's epilogue: out[n][2h][2w][c] = drop(act(lerp(lo) * A + C)).
struct Up2xEpiArgs {
    const float* lo;        // [n][h][w][c] fp32: the conv WITHOUT bias (the bias lives in C)
    int n, h, w, c;
    const float* coef_a;    // as ConvArgs: row = sample / coef_div, stride coef_stride
    const float* coef_c;
    int coef_stride, coef_div;
    int act;
    DropSpec drop;
    el16_t* out;            // [n][2h][2w][c]
};
hipError_t launch_up2x_epilogue(const Up2xEpiArgs& a, hipStream_t s);

// K5: GroupNorm(G) + FiLM + LeakyReLU + Dropout on an fp32 NHWC tensor -> NHWC bf16 (unet_simple.py:56 + :72-80)
struct GroupNormArgs {
    const float* x;         // [n][hw][c]
    int n, hw, c, groups;
    const float* gamma;
    const float* beta;
    const float* film_a;    // [rows][...] (1+scale) at film_off + ch ; stride film_stride (0 = broadcast)
    const float* film_c;
    int film_stride;
    int film_div;           // samples per coefficient row (0/1: one row per sample)
    int act;
    DropSpec drop;
    el16_t* out;
};
hipError_t launch_groupnorm(const GroupNormArgs& a, hipStream_t s);

// K3+K4: ConvTranspose2d(k4,s2,p1) readout evaluated only where the final bilinear resample needs it
// (unet_simple.py:141-151 + :195) -> NCHW fp32
struct ReadoutArgs {
    const el16_t* x;        // [n][ih][iw][cin] decoder output
    int n, ih, iw, cin;
    int iw_store;           // columns actually stored per row of x (== iw, or the compact width of the sparse-column decoder block)
    const int16_t* col_map; // [iw]: column -> stored column, -1 = not stored (compact x), or null
    const float* wgt;       // [kh][kw][cin][cout] fp32 (repacked ConvTranspose weight)
    const el16_t* wfrag;    // the same weights as MFMA 16x16x32 A fragments [16 taps][2 k halves][64 lanes][8] bf16 (cin == 64), or null
    const float* bias;      // [cout]
    int cout;
    int oh, ow;             // native grid
    int nearest;            // final resample: 0 bilinear, 1 nearest (outer_sample_mode)
    float* out;             // [n][cout][oh][ow]
    // per-geometry tap tables (launch_readout_tables, built once at weight upload), or null: entry = {int off[4]; float w[4]}.
    // row_tab[oy]: byte offset of input row i(oy, kh) inside a sample (i * iw_store * 128, -1 outside) and its bilinear row weight;
    // col_tab[ox]: byte offset of the stored input column j(ox, kw) (col_map applied, -1 outside) and its column weight
    const uint4* row_tab;
    const uint4* col_tab;
};
hipError_t launch_readout(const ReadoutArgs& a, hipStream_t s);
// fills a.row_tab (oh entries of 32 B) / a.col_tab (ow entries) for the geometry and col_map of `a` (device pointers, writable)
hipError_t launch_readout_tables(const ReadoutArgs& a, hipStream_t s);

// K10: sampler elementwise (dyffusion.py:381-391, :219-227)
hipError_t launch_cold_update(float* x_s, const float* x_cur, const float* x_next, long long count, hipStream_t s, float* copy = nullptr);
// out = tau*cond + (1-tau)*noise ; noise from `noise` if non-null else Box-Muller on the counter RNG
hipError_t launch_noisy_condition(float* out, const float* cond, const float* noise, float tau, long long count,
                                  int row_elems, uint32_t* rng_state, hipStream_t s);
// rng_state (device): {seed_lo, seed_hi, forward counter, global index of the engine's batch row 0, noise-call counter, ...}
#define DYF_RNG_STATE_WORDS 8

// boundary conditions of the physical-systems benchmark on a (fields, rows, C, H, W) fp32 stack, in place (kernels.hip)
struct BcArgs {
    float* preds;
    int kind;                    // 0 navier-stokes, 1 spring-mesh
    int n_fields, rows, c, h, w, n_meta;
    const int* row_meta;         // [rows]
    const float* time_factor;    // [n_fields] or [n_fields][n_meta]: 1 - exp(-5 t)
    int times_per_meta;
    const uint8_t* fixed_mask;   // [n_meta][c][h][w]
    const float* in_velocity;    // [n_meta]
    const float* vertex_y;       // [n_meta][w]
    const float* boundary;       // [n_meta][c][h][w]
};
hipError_t launch_boundary_conditions(const BcArgs& a, hipStream_t s);
hipError_t launch_nhwc_to_nchw_f32(const el16_t* src, int n, int h, int w, int w_store, int c, const int16_t* col_map,
                                   float* out, hipStream_t s);
hipError_t launch_rng_clone(uint32_t* dst, const uint32_t* src, uint32_t row_add, bool counters_only, hipStream_t s);
hipError_t launch_rng_begin_forward(uint32_t* rng_state, uint32_t* row_keys, int rows, int rows_per_fwd, hipStream_t s);
hipError_t launch_fill_f32(float* p, float v, long long count, hipStream_t s);
// on-device ensemble metrics (evaluation.py:10-118): sums[3] (fp64, device) = {sum (mean-y)^2, sum var, sum crps}; n_members <= 64 (one KB of LDS per member)
// sum of the criterion terms |p-t| (kind 0), (p-t)^2 (1), smooth-L1 (2) over `count` fp32 elements -> *sum (device double)
hipError_t launch_criterion_sum(const float* p, const float* t, long long count, int kind, double* sum, hipStream_t s);
hipError_t launch_ensemble_metrics(const float* preds, const float* targets, int n_members, long long n_points, double* sums,
                                   hipStream_t s);

// dyf_sample_gather: all-gather receive layout [world][slots][nb][row floats] -> [slots][total_rows][row] in global row order;
// rank r owns the contiguous block of rows shard(r) (the first total_rows % world ranks one extra), its padding rows are dropped
hipError_t launch_gather_unpack(const float* recv, float* out, int world, int slots, int nb, int total_rows, long long row, hipStream_t s);
