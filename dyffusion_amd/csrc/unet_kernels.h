// Kernels of the ResNet-UNet backbone (src/models/unet.py, OISST / synthetic configs): launch prototypes.
#pragma once
#include "common.h"

// init_conv: k x k (7x7, pad 3) conv of the channel-concatenated NCHW fp32 inputs -> NHWC bf16 (unet.py:149-151, :273)
struct StemConvArgs {
    const float* src[4];
    int ch[4];
    int nsrc, cin;
    int n, h, w;
    int k, pad;
    const float* wgt;   // [k*k][cin][dim] fp32
    const float* bias;  // [dim]
    int dim;
    el16_t* out;        // [n][h][w][dim]
    const el16_t* wfrag = nullptr;  // pack_stem_frag(): MFMA form (dim == 64), or null -> VALU form
    int ksteps = 0;                 // stem_frag_steps(k, cin) = cin * ceil(k*k / 16)
};
// k16 steps of the MFMA stem (channel-major K, the taps of a channel padded to whole steps), 0 = shape not taken by it
int stem_frag_steps(int k, int cin);
// [tap][cin][dim] fp32 -> A fragments of the MFMA stem: [channel][step of the channel][32-channel block][hi/lo part][lane][8] 16-bit
void pack_stem_frag(const float* wgt, int k, int cin, int dim, el16_t* out);
hipError_t launch_stem_conv(const StemConvArgs& a, hipStream_t s);

// K5: GroupNorm(G) + FiLM + SiLU + Dropout (+ residual) of unet.Block (unet.py:58-76) on a bf16 NHWC tensor
struct GnActArgs {
    const el16_t* x;        // raw conv output [n][hw][c]
    int n, hw, c, groups;
    const float* gamma;
    const float* beta;
    const float* film_a;    // (1+scale) rows, or null (second Block of a ResnetBlock has no FiLM)
    const float* film_c;
    int film_stride;
    int act;
    DropSpec drop;
    const el16_t* residual; // added last (ResnetBlock: h + residual_conv(x)), or null
    el16_t* out;
    double* stats;          // device scratch of gn_stats_doubles(n, groups) doubles for the vectorised form, or null
    // statistics already taken by the producing conv's epilogue (ConvArgs::gn_part, from the fp32 accumulators): per sample
    // `part_slots` partial (sum, sum of squares) pairs per 8-channel octet, [n][part_slots][c / 8][2] fp32, or null.  The apply
    // kernel then finalises (mean, 1/std) itself, adding the slots in index order: ONE launch and no statistics pass over the tensor.
    const float* part = nullptr;
    int part_slots = 0;
};
#define GN_MAX_BLOCKS 64    // workgroups per sample of the statistics pass
inline size_t gn_stats_doubles(size_t n, size_t groups) { return n * groups * (2 * GN_MAX_BLOCKS + 1); }
hipError_t launch_gn_act(const GnActArgs& a, hipStream_t s);
bool gn_part_supported(int c, int groups);

// forward-epoch word of the fused GroupNorm convs (gn_fused.h): ++*epoch, once per forward, on the forward's stream
hipError_t launch_gn_epoch_bump(uint32_t* epoch, hipStream_t s);

// K6: channel LayerNorm (gain only, biased variance, eps 1e-5) + optional Dropout (unet.py:43-52; attention.py:12)
struct LayerNormArgs {
    const el16_t* x;    // [pixels][c]
    long long pixels;   // n * hw
    int hw;             // pixels per batch row (the dropout streams are per row)
    int c;
    const float* g;     // [c]
    DropSpec drop;
    el16_t* out;
};
hipError_t launch_layernorm_c(const LayerNormArgs& a, hipStream_t s);

// K7: LinearAttention core (attention.py:22-31, rescale "qkv"): qkv [n][hw][3*heads*32] -> out [n][hw][heads*32]
struct LinAttnArgs {
    const el16_t* qkv;
    int n, hw, heads;   // dim_head = 32
    el16_t* out;
    float* scratch;     // device fp32 [n*heads][ceil(hw/1024)*1088 + 1024] for the pixel-parallel form, or null
};
hipError_t launch_linear_attention(const LinAttnArgs& a, hipStream_t s);

// K7 fused: LayerNorm output -> to_qkv -> LinearAttention core -> to_out + bias + residual, qkv never written (dim 64 / 128)
struct LinAttnFusedArgs {
    const el16_t* xn;         // [n][hw][c] normalised (and dropped) input
    const el16_t* xres;       // [n][hw][c] residual (the block input)
    int n, hw, c;
    const el16_t* wqkv_frag;  // linattn_fused_pack() orders
    const el16_t* wout_frag;
    const float* bout;        // [c]
    el16_t* y;                // [n][hw][c]
    float* scratch;           // as LinAttnArgs.scratch, sized for 8-group workgroups (rn_alloc_workspace)
    int groups_per_block;     // 32-pixel groups per workgroup: 32 (the large-batch form), 16 or 8; 0 = chosen by the launcher from the
                              // workgroup count (batch_invariant engines pin 32: the partials are merged per workgroup)
    long long scratch_floats; // capacity of `scratch` (0 = unknown: only 32-group workgroups are used)
};
bool linattn_fused_supported(int c);
hipError_t linattn_fused_init();
void linattn_fused_pack(const float* w_qkv, const float* w_out, int c, el16_t* qkv_frag, el16_t* out_frag);
hipError_t launch_linear_attention_fused(const LinAttnFusedArgs& a, hipStream_t s);

// K8 (VALU form for short sequences): softmax(q*scale . k) -> Dropout -> . v  (attention.py:62-72)
struct AttnArgs {
    const el16_t* qkv;      // [n][hw][3*heads*32]
    int n, hw, heads;
    DropSpec drop;          // on the probabilities; element index ((n*heads + h)*hw + i)*hw + j
    el16_t* out;            // [n][hw][heads*32]
};
hipError_t launch_attention(const AttnArgs& a, hipStream_t s);

// final 1x1 conv to the output channels -> NCHW fp32 (unet.py:244-245, :309)
struct HeadArgs {
    const el16_t* x;    // [n][hw][c]
    int n, hw, c, cout;
    const float* wgt;   // [cout][c]
    const float* bias;
    float* out;         // [n][cout][hw]
};
hipError_t launch_head(const HeadArgs& a, hipStream_t s);

// nn.Upsample(scale_factor=2, mode="nearest") (unet.py:16-19) on NHWC bf16
hipError_t launch_up2x_nearest(const el16_t* src, int n, int h, int w, int c, el16_t* out, hipStream_t s);
// Dropout over a whole NHWC tensor (input_dropout of unet.Unet); per_row = elements of one sample (even)
hipError_t launch_drop16(const el16_t* x, el16_t* y, int n, long long per_row, const DropSpec& d, hipStream_t s);
