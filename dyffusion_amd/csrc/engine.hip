// Host side of libdyffusion_hip.so: engine object, weight preparation (K11), per-network forward, sampling-loop
// executor with hipGraph capture, and the extern "C" entry points declared in include/dyffusion_hip.h.
//
// Path being replaced (reference, read-only): src/diffusion/dyffusion.py:335-431 (sample_loop / sample),
// :140-163 + :480-494 (q_sample / _interpolate), :205-239 (predict_x_last) and src/models/unet_simple.py:164-197.
#include "engine_internal.h"

void dyf_form_set(const char* key, const char* value);  // conv.hip: the kernel-form switch table (common.h dyf_form)
std::string dyf_form_text();
void dyf_prof_arm(const char* name);  // conv.hip: named-kernel timing (common.h KernelProf)
void dyf_prof_collect(double* total_ms, double* total_bytes, int* launches);

#include <dlfcn.h>
#include <mutex>
#include <rccl/rccl.h>  // types only: the functions are resolved with dlsym (rccl_api below)
#include "unet_kernels.h"
#include "../../include/dyffusion_hip_testing.h"

using namespace dyf;

namespace {

thread_local bool g_creating_group_child = false;  // dyf_engine_create called for a row group: no groups of its own

// Row groups by default (dyf_set_row_groups / DYF_ROW_GROUPS override): the ResNet-UNet on small planes launches ~150 kernels per
// forward, most of them under-filling 256 CUs or leaving a ragged last round (528 workgroups on 512 slots at the 15 x 15 level
// of the OISST shapes at 300 rows); three concurrent rollouts of a third of the rows each cover one another's tails and launch
// gaps (+16 % at 300 rows).  unet_simple at 80 rows gains 1.5 % with two groups -- not worth a second copy
// of the workspace -- and large planes (512 x 512) fill the chip with a few rows: no groups.
int default_row_groups(const dyf_engine* e) {
    if (e->cfg.net[0].arch != DYF_ARCH_UNET_RESNET || e->cfg.net[1].arch != DYF_ARCH_UNET_RESNET) return 1;
    if (e->cfg.batch_invariant) return 1;  // forms are pinned to 2 max_batch rows: a group's smaller max_batch would change them
    const long long plane = (long long)e->cfg.height * e->cfg.width;
    if (plane > 128 * 128) return 1;
    const long long pix = plane * e->cfg.max_batch;
    // OISST shapes (60 x 60), fields/s with 1 / 2 / 3 / 4 groups: 300 rows 3 198 / 3 568 / 3 722 / 2 966; 150 rows 2 758 / 3 005 /
    // 3 126 / 2 209; 80 rows 2 182 / 2 231 / 2 055 / 1 347; 512 rows 3 620 / 3 805 / 3 835 / 3 541 (round 3).  Re-measured at the end
    // of round 4 (GroupNorm fused at every level, 1 / 2 / 3 groups): 64 rows 2 328 / 2 158 / 2 197; 75 rows 2 307 / 2 387 / 2 439;
    // 100 rows 2 737 / 2 810 / 2 898; 120 rows 3 101 / 3 085 / 3 181 -> three groups from 72 rows on, none below (two groups are
    // what an engine that owns a communicator is capped to, dyf_comm_init)
    if (pix >= 72ll * 60 * 60) return 3;
    return 1;
}

// ------------------------------------------------------------------------------------------------ geometry
void layout_blocks(Net& n) {
    const int d = n.dim;
    // (cin, cout, kernel, stride, pad, norm, act): unet_simple.py:119-139 through UNetBlock.__init__ (:14-56)
    const int enc[6][5] = {{d, 2 * d, 4, 2, 1}, {2 * d, 2 * d, 4, 2, 1}, {2 * d, 4 * d, 4, 2, 1},
                           {4 * d, 8 * d, 4, 2, 1}, {8 * d, 8 * d, 2, 2, 0}, {8 * d, 8 * d, 2, 2, 0}};
    const int dec[6][5] = {{8 * d, 8 * d, 1, 1, 0},  {16 * d, 8 * d, 1, 1, 0}, {16 * d, 4 * d, 3, 1, 1},
                           {8 * d, 2 * d, 3, 1, 1},  {4 * d, 2 * d, 3, 1, 1},  {4 * d, d, 3, 1, 1}};
    int off = 0;
    for (int i = 0; i < 12; ++i) {
        const int* s = i < 6 ? enc[i] : dec[i - 6];
        UBlock& b = n.blk[i];
        b.cin = s[0]; b.cout = s[1]; b.k = s[2]; b.stride = s[3]; b.pad = s[4];
        b.gn = (i == 5);
        b.transposed = i >= 6;
        b.act = i < 6 ? ACT_LEAKY : ACT_RELU;
        b.film_off = off;
        off += b.cout;
    }
    n.total_c = off;
}

// returns "" or an error message
std::string layout_geometry(Net& n, int H, int W) {
    n.uh = n.cfg.upsample_h > 0 ? n.cfg.upsample_h : H;
    n.uw = n.cfg.upsample_w > 0 ? n.cfg.upsample_w : W;
    int h = n.uh, w = n.uw;
    for (int i = 0; i < 6; ++i) {
        UBlock& b = n.blk[i];
        b.in_h = h; b.in_w = w;
        b.out_h = (h + 2 * b.pad - b.k) / b.stride + 1;
        b.out_w = (w + 2 * b.pad - b.k) / b.stride + 1;
        if (b.out_h < 1 || b.out_w < 1) return "resampled grid too small for six stride-2 encoder blocks";
        h = b.out_h; w = b.out_w;
    }
    for (int i = 6; i < 12; ++i) {
        UBlock& b = n.blk[i];
        b.in_h = 2 * h; b.in_w = 2 * w;
        b.out_h = b.in_h + 2 * b.pad - b.k + 1;
        b.out_w = b.in_w + 2 * b.pad - b.k + 1;
        h = b.out_h; w = b.out_w;
        if (i < 11) {  // torch.cat([x, skip]) must line up (unet_simple.py:176-177)
            const UBlock& skip = n.blk[10 - i];
            if (skip.out_h != h || skip.out_w != w)
                return "decoder/skip spatial sizes do not match (resampled grid must be divisible by 64)";
        }
    }
    // 2*MAC of conv/linear layers, as torch.utils.flop_counter counts them
    double f = 2.0 * n.uh * n.uw * (double)n.cin_total * n.dim;
    for (int i = 0; i < 12; ++i) {
        const UBlock& b = n.blk[i];
        f += 2.0 * b.out_h * b.out_w * (double)b.cout * b.cin * b.k * b.k;
    }
    f += 2.0 * h * w * (double)n.dim * n.cfg.out_channels * 16;  // ConvTranspose2d k4: per INPUT pixel
    if (n.cfg.with_time_emb) {
        f += 2.0 * ((double)n.dim * n.tdim + (double)n.tdim * n.tdim);
        f += 2.0 * (double)n.tdim * 2 * n.total_c;
    }
    n.flops_per_sample = f;
    return "";
}

// ------------------------------------------------------------------------------------------------ forward
DropSpec make_drop(const dyf_engine* e, const Net& n, const FwdOpts& o, int layer) {
    DropSpec d{};
    const float p = n.cfg.dropout;
    d.mode = (p > 0.0f) ? o.dropout_mode : 0;
    d.scale = 1.0f / (1.0f - p);
    d.thresh16 = keep_threshold16(p);
    d.salt = rng_layer_salt((uint32_t)layer);
    d.row_keys = e->row_keys;
    // injected masks arrive in execution order: dropout_input (when its p > 0) first, then the 12 blocks
    d.mask = (d.mode == 2 && o.masks) ? o.masks[layer + (n.cfg.input_dropout > 0.0f ? 1 : 0)] : nullptr;
    if (d.mode == 2 && d.mask == nullptr) d.mode = 0;
    return d;
}

// dropout_input of unet_simple (unet_simple.py:116,168): a Dropout on init_conv's output -- the first site of a forward
DropSpec make_input_drop(const dyf_engine* e, const Net& n, const FwdOpts& o) {
    DropSpec d{};
    const float p = n.cfg.input_dropout;
    d.mode = (p > 0.0f) ? o.dropout_mode : 0;
    d.scale = 1.0f / (1.0f - p);
    d.thresh16 = keep_threshold16(p);
    d.salt = rng_layer_salt(DYF_INPUT_DROP_SITE);
    d.row_keys = e->row_keys;
    d.mask = (d.mode == 2 && o.masks) ? o.masks[0] : nullptr;
    if (d.mode == 2 && d.mask == nullptr) d.mode = 0;
    return d;
}

dyf_status run_conv(dyf_engine* e, const ConvArgs& a, hipStream_t st) {
    const int path = (e->cfg.enable_mfma && conv_mfma_supported(a)) ? 1 : 0;
    HIP_TRY(e, launch_conv(a, path, st));
    return DYF_OK;
}

// enc0 on the fused stem: a 4(kh) x 1 conv over "64-channel" pixels that are really 4 adjacent 16-channel pixels of the
// zero-bordered stem16 tensor (stride 2, physical padding instead of pad=1).
void fused_enc0_args(const dyf_engine* e, const Net& n, ConvArgs& a) {
    a.src0 = e->ws.stem16; a.c0 = 64; a.pix_pitch0 = 16;
    a.h = n.uh + 2; a.w = n.uw + 2;
    a.kh = 4; a.kw = 1; a.stride = 2; a.pad = 0;
    a.wpk = n.enc0_fused_w;
}

// Fused x2-upsample conv pays ~7 extra K taps on every tile that touches an image border; on small planes most tiles
// do, and materialising the (small) upsampled tensor is cheaper (measured: break-even at a 32x32 low-res plane).
bool use_fused_up(const dyf_engine* e, const UBlock& b, const ConvArgs& f) {
    return e->cfg.enable_mfma && e->fuse_up2x && b.wpk_up != nullptr && f.h >= e->fuse_min_plane && f.w >= e->fuse_min_plane &&
           conv_mfma_supported(f);
}

ConvArgs block_conv_args(const dyf_engine* e, const Net& n, const UBlock& b, int nb) {
    ConvArgs a{};
    a.n = nb; a.h = b.in_h; a.w = b.in_w; a.ho = b.out_h; a.wo = b.out_w;
    a.kh = b.k; a.kw = b.k; a.stride = b.stride; a.pad = b.pad; a.cout = b.cout;
    a.wpk = b.wpk;
    a.act = b.act;
    a.zero_page = e->ws.zero_page;
    a.splitk_ws = e->ws.splitk; a.splitk_cap = DYF_SPLITK_FLOATS;
    a.n_sel = e->cfg.batch_invariant ? 2 * e->cfg.max_batch : 0;
    return a;
}

dyf_status net_forward(dyf_engine* e, int which, const Source* srcs, int nsrc, int nb, const FwdOpts& o, float* out_dev,
                       hipStream_t st) {
    Net& n = e->net[which];
    if (n.rn) return rn_forward(e, which, srcs, nsrc, nb, o, out_dev, st);
    if (n.sc) return sc_forward(e, which, srcs, nsrc, nb, o, out_dev, st);
    Workspace& ws = e->ws;
    const int H = e->cfg.height, W = e->cfg.width;
    // a forward that draws masks starts by filling the row-key table and advancing the forward counter: a one-block kernel of its
    // own, or -- fused stem, no dropout inside the stem -- block 0 of the stem launch (kernels.hip stem_rng_begin)
    const bool draws = o.dropout_mode == 1 && (n.cfg.dropout > 0.0f || n.cfg.input_dropout > 0.0f);
    const bool fold_rng = !(dyf_form("DYF_FOLD_RNG_BEGIN") && atoi(dyf_form("DYF_FOLD_RNG_BEGIN")) == 0);
    const bool rng_in_stem = draws && fold_rng && n.stem_fused && e->cfg.enable_mfma && e->fuse_stem && n.cfg.input_dropout == 0.0f;
    if (draws && !rng_in_stem)
        HIP_TRY(e, launch_rng_begin_forward(e->rng_state, e->row_keys, nb, o.src_rows > 0 ? o.src_rows : nb, st));
    // ---- stem: outer resample + 1x1 conv
    StemArgs sa{};
    int ctot = 0;
    for (int i = 0; i < nsrc; ++i) {
        sa.src[i] = srcs[i].p;
        sa.ch[i] = srcs[i].ch;
        ctot += srcs[i].ch;
    }
    if (ctot != n.cin_total)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "channel count of the network inputs does not match its configuration");
    sa.nsrc = nsrc; sa.cin = ctot; sa.n = nb; sa.src_rows = o.src_rows; sa.h = H; sa.w = W; sa.uh = n.uh; sa.uw = n.uw;
    sa.resample = (n.uh != H || n.uw != W) ? 1 : 0;
    sa.nearest = n.cfg.outer_nearest;
    sa.wgt = n.stem_w; sa.bias = n.stem_b; sa.dim = n.dim;
    // a Dropout between init_conv and the first encoder conv (input_dropout > 0) breaks their composition: separate stem kernel
    const bool fused_stem = n.stem_fused && e->cfg.enable_mfma && e->fuse_stem && n.cfg.input_dropout == 0.0f;
    sa.drop = make_input_drop(e, n, o);
    if (rng_in_stem) {
        sa.rng_state = e->rng_state; sa.rng_row_keys = e->row_keys; sa.rng_rows = nb; sa.rng_rows_per_fwd = o.src_rows > 0 ? o.src_rows : nb;
    }
    if (fused_stem) {
        sa.out = ws.stem16;
        HIP_TRY(e, launch_stem16(sa, st));
    } else {
        sa.out = ws.stem;
        HIP_TRY(e, launch_stem(sa, st));
    }
    // ---- encoder
    const el16_t* x = ws.stem;
    for (int i = 0; i < 6; ++i) {
        const UBlock& b = n.blk[i];
        ConvArgs a = block_conv_args(e, n, b, nb);
        a.src0 = x; a.c0 = b.cin; a.src1 = nullptr; a.c1 = 0;
        if (i == 0 && fused_stem) fused_enc0_args(e, n, a);
        if (!b.gn) {
            a.coef_a = o.coef_a + b.film_off; a.coef_c = o.coef_c + b.film_off; a.coef_stride = o.coef_stride;
            a.coef_div = o.coef_div;
            a.drop = make_drop(e, n, o, i);
            a.out_el16 = ws.enc[i];
            dyf_status s = run_conv(e, a, st);
            if (s != DYF_OK) return s;
        } else {
            a.coef_a = b.static_a; a.coef_c = b.static_c; a.coef_stride = 0;
            a.act = ACT_NONE;
            a.drop = DropSpec{};
            a.out_f32 = ws.enc5_raw;
            dyf_status s = run_conv(e, a, st);
            if (s != DYF_OK) return s;
            GroupNormArgs g{};
            g.x = ws.enc5_raw; g.n = nb; g.hw = b.out_h * b.out_w; g.c = b.cout; g.groups = 8;
            g.gamma = b.gamma; g.beta = b.beta;
            g.film_a = o.coef_a + b.film_off; g.film_c = o.coef_c + b.film_off; g.film_stride = o.coef_stride;
            g.film_div = o.coef_div;
            g.act = b.act; g.drop = make_drop(e, n, o, i); g.out = ws.enc[i];
            HIP_TRY(e, launch_groupnorm(g, st));
        }
        x = ws.enc[i];
    }
    // ---- decoder: x2 bilinear upsample of cat[x, skip] (materialised), conv, fused epilogue
    const el16_t* skip = nullptr;
    int skip_c = 0;
    const UBlock* sparse_out = nullptr;
    int lh = n.blk[5].out_h, lw = n.blk[5].out_w;
    for (int i = 6; i < 12; ++i) {
        const UBlock& b = n.blk[i];
        ConvArgs a = block_conv_args(e, n, b, nb);
        a.coef_a = o.coef_a + b.film_off; a.coef_c = o.coef_c + b.film_off; a.coef_stride = o.coef_stride;
        a.coef_div = o.coef_div;
        a.drop = make_drop(e, n, o, i);
        a.out_el16 = ws.dec[i - 6];
        // fused form: the conv gathers straight from the low-res cat[x, skip] (phase decomposition, conv.hip)
        ConvArgs f = a;
        f.src0 = x; f.c0 = b.cin - skip_c; f.src1 = skip; f.c1 = skip_c; f.h = lh; f.w = lw;
        f.up2x = 1; f.wpk_up = b.wpk_up; f.wpk_up_frag = b.wpk_up_frag; f.up_border = ws.up_border;
        f.up_cols = b.up_cols; f.up_cbase = b.up_cbase; f.up_cidx = b.up_cidx; f.up_ntiles = b.up_ntiles; f.up_npad = b.up_npad;
        f.up_nvalid0 = b.up_nvalid0; f.up_nvalid1 = b.up_nvalid1; f.up_wo_store = b.up_wo_store;
        f.up_mix[0] = b.up_mix[0]; f.up_mix[1] = b.up_mix[1]; f.up_mix[2] = b.up_mix[2];
        if (i == 11 && b.up_cols && e->poison_dec5)  // test hook: a needed-but-unwritten pixel of the sparse form shows as NaN
            HIP_TRY(e, hipMemsetAsync(ws.dec[5], 0xFF, (size_t)nb * b.out_h * b.out_w * b.cout * sizeof(el16_t), st));  // whole buffer
        // sparse-column form (last block only): its output tensor is compact, the readout below must know
        if (f.up_cols && !(use_fused_up(e, b, f) && conv_up_halo_supported(f))) f.up_cols = nullptr;
        if (f.up_cols) sparse_out = &b;
        const bool prof = e->prof_layer == i && e->prof_ev.size() < 4096;
        hipEvent_t pe0 = nullptr, pe1 = nullptr;
        if (prof) {  // dyf_time_layer_in_rollout: HIP events around this block's conv, on the launch stream
            HIP_TRY(e, hipEventCreate(&pe0));
            HIP_TRY(e, hipEventCreate(&pe1));
        }
        // 1 x 1 decoder blocks (dec0, dec1): the pointwise conv commutes with the per-channel bilinear upsample, so it runs on the
        // LOW-res cat[x, skip] (a quarter of the pixels, nothing materialised) into fp32 and one pass upsamples + applies the block's
        // epilogue (kernels.h Up2xEpiArgs); DYF_DEC_COMMUTE=0 keeps upsample -> conv
        const bool commute = !(dyf_form("DYF_DEC_COMMUTE") && atoi(dyf_form("DYF_DEC_COMMUTE")) == 0);
        const bool commuted = commute && b.k == 1 && b.stride == 1 && b.pad == 0 && e->cfg.enable_mfma && (b.cout & 3) == 0 &&
                              (size_t)nb * lh * lw * b.cout * sizeof(float) <= (size_t)nb * b.in_h * b.in_w * b.cin * sizeof(el16_t);
        if (use_fused_up(e, b, f)) {
            if (prof) HIP_TRY(e, hipEventRecord(pe0, st));
            HIP_TRY(e, launch_conv(f, 1, st));
            if (prof) HIP_TRY(e, hipEventRecord(pe1, st));
        } else if (commuted) {
            ConvArgs lo = a;
            lo.src0 = x; lo.c0 = b.cin - skip_c; lo.src1 = skip; lo.c1 = skip_c;
            lo.h = lh; lo.w = lw; lo.ho = lh; lo.wo = lw;
            lo.coef_a = ws.ones_f; lo.coef_c = ws.zeros_f; lo.coef_stride = 0; lo.coef_div = 0;
            lo.act = ACT_NONE; lo.drop = DropSpec{};
            lo.out_el16 = nullptr; lo.out_f32 = (float*)ws.up;  // (the materialised-upsample scratch is free in this form)
            if (prof) HIP_TRY(e, hipEventRecord(pe0, st));
            dyf_status s = run_conv(e, lo, st);
            if (s != DYF_OK) return s;
            Up2xEpiArgs u{};
            u.lo = (const float*)ws.up; u.n = nb; u.h = lh; u.w = lw; u.c = b.cout;
            u.coef_a = a.coef_a; u.coef_c = a.coef_c; u.coef_stride = a.coef_stride; u.coef_div = a.coef_div;
            u.act = a.act; u.drop = a.drop; u.out = a.out_el16;
            HIP_TRY(e, launch_up2x_epilogue(u, st));
            if (prof) HIP_TRY(e, hipEventRecord(pe1, st));
        } else {  // materialise the upsampled tensor, then a plain conv
            Up2xArgs u{};
            u.src0 = x; u.c0 = b.cin - skip_c; u.src1 = skip; u.c1 = skip_c; u.n = nb; u.h = lh; u.w = lw; u.out = ws.up;
            HIP_TRY(e, launch_up2x(u, st));
            a.src0 = ws.up; a.c0 = b.cin; a.src1 = nullptr; a.c1 = 0;
            if (prof) HIP_TRY(e, hipEventRecord(pe0, st));
            dyf_status s = run_conv(e, a, st);
            if (s != DYF_OK) return s;
            if (prof) HIP_TRY(e, hipEventRecord(pe1, st));
        }
        if (prof) {
            e->prof_ev.emplace_back(pe0, pe1);
            e->prof_rows.push_back(nb);
        }
        x = ws.dec[i - 6];
        lh = b.out_h; lw = b.out_w;
        if (i < 11) {
            skip = ws.enc[10 - i];
            skip_c = n.blk[10 - i].cout;
        }
    }
    // ---- readout (sparse transposed conv + final resample)
    ReadoutArgs r{};
    r.x = x; r.n = nb; r.ih = lh; r.iw = lw; r.cin = n.dim; r.wgt = n.ro_w; r.wfrag = n.ro_wfrag; r.bias = n.ro_b; r.cout = n.cfg.out_channels;
    r.oh = H; r.ow = W; r.out = out_dev; r.nearest = n.cfg.outer_nearest;
    e->last_dec5_sparse = sparse_out != nullptr;
    r.iw_store = sparse_out ? sparse_out->up_wo_store : lw;
    r.col_map = sparse_out ? sparse_out->up_col_map : nullptr;
    if (n.ro_row_tab && (sparse_out != nullptr) == n.ro_tab_sparse) {  // the tables were built for this storage layout
        r.row_tab = n.ro_row_tab;
        r.col_tab = n.ro_col_tab;
    }
    HIP_TRY(e, launch_readout(r, st));
    return DYF_OK;
}

// coefficient rows for `rows` time values already on the device
dyf_status compute_coefs(dyf_engine* e, Net& n, const float* time_dev, int rows, float* coef_a, float* coef_c,
                         hipStream_t st) {
    FilmArgs f{};
    if (n.cfg.with_time_emb) {
        TimeMlpArgs t{};
        t.time = time_dev; t.rows = rows; t.dim = n.dim; t.w1 = n.t_w1; t.b1 = n.t_b1; t.w2 = n.t_w2; t.b2 = n.t_b2;
        t.silu_out = e->ws.silu;
        if (n.t_learned) { t.learned_w = n.t_learned; t.learned_half = n.cfg.learned_sinusoidal_dim / 2; }
        HIP_TRY(e, launch_time_mlp(t, st));
        f.silu = e->ws.silu;
    }
    f.rows = rows; f.tdim = n.tdim; f.total_c = n.total_c; f.wf = n.film_w; f.bf = n.film_b; f.blk_of = n.blk_of;
    f.blk_off = n.blk_off; f.blk_cout = n.blk_cout; f.norm_a = n.norm_a; f.norm_c = n.norm_c; f.coef_a = coef_a;
    f.coef_c = coef_c;
    HIP_TRY(e, launch_film(f, st));
    return DYF_OK;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int32_t dyf_abi_version(void) { return DYF_ABI_VERSION; }

int32_t dyf_dtype(void) { return DYF_F16 ? DYF_DTYPE_F16 : DYF_DTYPE_BF16; }

const char* dyf_last_error(const dyf_engine* engine) { return engine ? engine->err.c_str() : g_create_error.c_str(); }

static void destroy_groups(dyf_engine* e) {
    for (dyf_engine* c : e->groups) {
        if (c->group_stream) { (void)hipStreamSynchronize(c->group_stream); (void)hipStreamDestroy(c->group_stream); }
        if (c->group_done) (void)hipEventDestroy(c->group_done);
        dyf_engine_destroy(c);
    }
    e->groups.clear();
    e->last_groups = 0;
}

void dyf_engine_destroy(dyf_engine* e) {
    if (!e) return;
    (void)hipSetDevice(e->cfg.device);
    destroy_groups(e);
    if (e->group_fork) (void)hipEventDestroy(e->group_fork);
    for (auto& kv : e->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    rn_destroy(e->net[0]);
    rn_destroy(e->net[1]);
    sc_destroy(e->net[0]);
    sc_destroy(e->net[1]);
    (void)dyf_comm_destroy(e);
    if (e->s_log) (void)hipFree(e->s_log);
    if (e->cap_stream) (void)hipStreamDestroy(e->cap_stream);
    if (e->gn_err_host) (void)hipHostFree(e->gn_err_host);
    train_destroy(e);
    release_allocs(e->allocs);
    release_allocs(e->net_allocs[0]);
    release_allocs(e->net_allocs[1]);
    release_allocs(e->plan_allocs);
    delete e;
}

dyf_status dyf_engine_create(const dyf_engine_config* cfg, dyf_engine** out_engine) {
    if (!cfg || !out_engine) return fail(nullptr, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (cfg->abi_version != DYF_ABI_VERSION) return fail(nullptr, DYF_ERR_INVALID_ARGUMENT, "ABI version mismatch");
    if (cfg->dtype != dyf_dtype())
        return fail(nullptr, DYF_ERR_INVALID_ARGUMENT, std::string("this library is the ") + DYF_DTYPE_NAME +
                    " build: engine dtype must match (libdyffusion_hip.so = bf16, libdyffusion_hip_f16.so = fp16)");
    if (cfg->height < 1 || cfg->width < 1 || cfg->max_batch < 1)
        return fail(nullptr, DYF_ERR_INVALID_ARGUMENT, "height, width and max_batch must be positive");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev < 1)
        return fail(nullptr, DYF_ERR_HIP, "no HIP device available: the DYffusion engine requires an MI355X (gfx950)");
    if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, DYF_ERR_INVALID_ARGUMENT, "device ordinal out of range");
    if (hipSetDevice(cfg->device) != hipSuccess) return fail(nullptr, DYF_ERR_HIP, "hipSetDevice failed");

    dyf_engine* e = new dyf_engine();
    e->cfg = *cfg;
    if (const char* fu = dyf_form("DYF_FUSE_UP2X")) e->fuse_up2x = atoi(fu) != 0;
    if (const char* fm = dyf_form("DYF_FUSE_MIN_PLANE")) e->fuse_min_plane = atoi(fm);
    if (const char* fs = dyf_form("DYF_FUSE_STEM")) e->fuse_stem = atoi(fs) != 0;
    if (const char* pi = dyf_form("DYF_PAIR_INTERP")) e->pair_interp = atoi(pi) != 0;
    if (const char* pz = dyf_form("DYF_POISON_DEC5")) e->poison_dec5 = atoi(pz) != 0;
    if (conv_init() != hipSuccess || linattn_fused_init() != hipSuccess || hipStreamCreateWithFlags(&e->cap_stream, hipStreamNonBlocking) != hipSuccess) {
        delete e;
        return fail(nullptr, DYF_ERR_HIP, "engine initialisation failed (conv_init / stream create)");
    }
    auto bail = [&](dyf_status st, const std::string& m) {
        g_create_error = m;
        dyf_engine_destroy(e);
        return st;
    };
    for (int w = 0; w < 2; ++w) {
        Net& n = e->net[w];
        n.cfg = cfg->net[w];
        if (n.cfg.arch != DYF_ARCH_UNET_SIMPLE && n.cfg.arch != DYF_ARCH_UNET_RESNET && n.cfg.arch != DYF_ARCH_SIMPLE_CONV_NET)
            return bail(DYF_ERR_UNSUPPORTED, "arch must be unet_simple (0), unet / resnet (1) or simple_conv_net (2)");
        if (n.cfg.dim < 4 || (n.cfg.dim & 1)) return bail(DYF_ERR_INVALID_ARGUMENT, "dim must be even and >= 4");
        if (n.cfg.input_dropout < 0.0f || n.cfg.input_dropout >= 1.0f) return bail(DYF_ERR_INVALID_ARGUMENT, "input_dropout must be in [0, 1)");
        if (n.cfg.input_dropout != 0.0f && n.cfg.arch == DYF_ARCH_SIMPLE_CONV_NET)
            return bail(DYF_ERR_UNSUPPORTED, "input_dropout > 0: arch simple_conv_net has no such layer");
        if (n.cfg.dropout < 0.0f || n.cfg.dropout >= 1.0f) return bail(DYF_ERR_INVALID_ARGUMENT, "dropout must be in [0, 1)");
        n.cin_total = n.cfg.in_channels + n.cfg.cond_channels;
        if (n.cin_total < 1 || n.cin_total > DYF_MAX_IN_CH || n.cfg.out_channels < 1 || n.cfg.out_channels > DYF_MAX_OUT_CH)
            return bail(DYF_ERR_UNSUPPORTED, "channel counts outside the supported range (inputs+cond <= 32, outputs <= 8)");
        n.dim = n.cfg.dim;
        n.tdim = 2 * n.cfg.dim;
        if (n.cfg.arch == DYF_ARCH_UNET_RESNET) {
            if (n.cfg.block_dropout1 < 0.0f || n.cfg.block_dropout1 >= 1.0f || n.cfg.attn_dropout < 0.0f || n.cfg.attn_dropout >= 1.0f)
                return bail(DYF_ERR_INVALID_ARGUMENT, "dropout rates must be in [0, 1)");
            std::string m = rn_configure(e, n);
            if (!m.empty()) return bail(DYF_ERR_INVALID_ARGUMENT, m);
            continue;
        }
        if (n.cfg.arch == DYF_ARCH_SIMPLE_CONV_NET) {
            std::string m = sc_configure(e, n);
            if (!m.empty()) return bail(DYF_ERR_INVALID_ARGUMENT, m);
            continue;
        }
        n.n_drop_sites = (n.cfg.dropout > 0.0f ? 12 : 0) + (n.cfg.input_dropout > 0.0f ? 1 : 0);
        layout_blocks(n);
        std::string m = layout_geometry(n, cfg->height, cfg->width);
        if (!m.empty()) return bail(DYF_ERR_INVALID_ARGUMENT, m);
    }
    // ---- workspace (sized for max_batch, shared by the two networks which run back to back)
    // sized for 2 x max_batch rows: the sampler runs the two interpolator calls of a step as ONE forward over 2 nb rows
    const size_t nb = (size_t)cfg->max_batch * 2;
    size_t stem_el = 0, enc_el[6] = {}, dec_el[6] = {}, up_el = 0, raw_el = 0, tc = 0, td = 0, ub_el = 0;
    for (int w = 0; w < 2; ++w) {
        const Net& n = e->net[w];
        stem_el = std::max(stem_el, nb * n.uh * n.uw * n.dim);
        for (int i = 0; i < 6; ++i) {
            const UBlock& b = n.blk[i];
            enc_el[i] = std::max(enc_el[i], nb * b.out_h * b.out_w * b.cout);
            const UBlock& d = n.blk[6 + i];
            dec_el[i] = std::max(dec_el[i], nb * d.out_h * d.out_w * d.cout);
            up_el = std::max(up_el, nb * d.in_h * d.in_w * d.cin);
            ub_el = std::max(ub_el, conv_up_border_floats((int)nb, d.out_h / 2, d.out_w / 2, d.cout));
        }
        raw_el = std::max(raw_el, nb * n.blk[5].out_h * n.blk[5].out_w * n.blk[5].cout);
        tc = std::max<size_t>(tc, n.total_c);
        td = std::max<size_t>(td, n.tdim);
    }
    Workspace& ws = e->ws;
#define ALLOC(ptr, count)                                           \
    do {                                                            \
        dyf_status _s = dev_alloc(e, &(ptr), (count));              \
        if (_s != DYF_OK) return bail(_s, e->err);                  \
    } while (0)
    ALLOC(ws.stem, stem_el);
    {
        size_t s16 = 0;
        for (int w = 0; w < 2; ++w) s16 = std::max(s16, nb * (size_t)(e->net[w].uh + 2) * (e->net[w].uw + 2) * 16);
        ALLOC(ws.stem16, s16);
    }
    for (int i = 0; i < 6; ++i) {
        ALLOC(ws.enc[i], enc_el[i]);
        ALLOC(ws.dec[i], dec_el[i]);
    }
    ALLOC(ws.enc5_raw, raw_el);
    ALLOC(ws.up, up_el);
    ALLOC(ws.silu, nb * td);
    ALLOC(ws.coef_a, nb * tc);
    ALLOC(ws.coef_c, nb * tc);
    ALLOC(ws.zero_page, 256);
    ALLOC(ws.up_border, ub_el);
    ALLOC(ws.coef_pair, 4 * tc);
    ALLOC(ws.splitk, DYF_SPLITK_FLOATS);
    {
        size_t cmax = 64;
        for (int w = 0; w < 2; ++w)
            for (int i = 0; i < 12; ++i) cmax = std::max<size_t>(cmax, (size_t)e->net[w].blk[i].cout);
        ALLOC(ws.zeros_f, cmax);
        ALLOC(ws.ones_f, cmax);
        std::vector<float> ones(cmax, 1.0f);
        if (hipMemcpy(ws.ones_f, ones.data(), cmax * sizeof(float), hipMemcpyHostToDevice) != hipSuccess)
            return bail(DYF_ERR_HIP, "hipMemcpy(ones)");
    }
    ALLOC(e->s_pair, 2 * (size_t)cfg->max_batch * DYF_MAX_OUT_CH * cfg->height * cfg->width);
    ALLOC(e->s_time, 64);
    ALLOC(e->rng_state, DYF_RNG_STATE_WORDS);
    ALLOC(e->row_keys, 4 * nb);  // [2 x max_batch rows][2]
#undef ALLOC
    // host-visible error word of the fused GroupNorm convs (gn_fused.h): pinned + mapped, read by the host without a sync
    if (hipHostMalloc((void**)&e->gn_err_host, 64, hipHostMallocMapped) == hipSuccess) {
        memset(e->gn_err_host, 0, 64);
        if (hipHostGetDevicePointer((void**)&e->gn_err_dev, e->gn_err_host, 0) != hipSuccess) e->gn_err_dev = nullptr;
    }
    if (!e->gn_err_dev) e->gn_fuse_disabled = true;  // no way to report a timed-out sweep: keep to the three-kernel path
    if (const char* gf = dyf_form("DYF_GN_FUSED")) if (atoi(gf) == 0) e->gn_fuse_disabled = true;
    {
        dyf_status rs = rn_alloc_workspace(e);
        if (rs != DYF_OK) return bail(rs, e->err);
        rs = sc_alloc_workspace(e);
        if (rs != DYF_OK) return bail(rs, e->err);
    }
    if (const char* gm = dyf_form("DYF_GROUP_MIN_ROWS")) e->group_min_rows = std::max(1, atoi(gm));
    *out_engine = e;
    if (!g_creating_group_child) {
        // default: DYF_ROW_GROUPS, else by architecture (measured on the ResNet-UNet shapes)
        int g = 1;
        if (const char* rg = dyf_form("DYF_ROW_GROUPS")) g = atoi(rg);
        else g = default_row_groups(e);
        if (g > 1) {
            dyf_status gs = dyf_set_row_groups(e, g);
            if (gs != DYF_OK) {
                g_create_error = e->err;
                dyf_engine_destroy(e);
                *out_engine = nullptr;
                return gs;
            }
        }
    }
    return DYF_OK;
}

dyf_status dyf_set_row_groups(dyf_engine* e, int32_t n_groups) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (e->is_group_child) return fail(e, DYF_ERR_STATE, "row groups do not nest");
    if (n_groups < 1 || n_groups > 8) return fail(e, DYF_ERR_INVALID_ARGUMENT, "n_groups must be in [1, 8]");
    if (e->net[0].loaded || e->net[1].loaded)
        return fail(e, DYF_ERR_STATE, "dyf_set_row_groups must be called before dyf_load_weights (the groups hold their own packed weights)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipDeviceSynchronize());
    destroy_groups(e);
    if (n_groups == 1 || e->cfg.max_batch < 2 * e->group_min_rows) return DYF_OK;
    n_groups = std::min<int>(n_groups, e->cfg.max_batch / e->group_min_rows);
    if (!e->group_fork) HIP_TRY(e, hipEventCreateWithFlags(&e->group_fork, hipEventDisableTiming));
    dyf_engine_config cc = e->cfg;
    // a child holds a HALF of max_batch even when there are three or more groups: a call may then run on two groups only (what
    // sample_into_stack does while the engine owns a communicator, below) without re-creating anything
    cc.max_batch = (e->cfg.max_batch + std::min(n_groups, 2) - 1) / std::min(n_groups, 2);
    for (int g = 0; g < n_groups; ++g) {
        dyf_engine* c = nullptr;
        g_creating_group_child = true;
        dyf_status cs = dyf_engine_create(&cc, &c);
        g_creating_group_child = false;
        if (cs != DYF_OK) {
            destroy_groups(e);
            return fail(e, cs, "row group engine: " + g_create_error);
        }
        c->is_group_child = true;
        // kernel forms of a group's launches are chosen by the tile count of all n_groups concurrent launches (OISST 300 rows:
        // 3 680 -> 3 750 fields/s; the 100-row shares otherwise fall below the tile thresholds of the large-batch forms)
        c->form_rows_scale = n_groups;
        if (const char* fs = dyf_form("DYF_GROUP_FORM_SCALE")) c->form_rows_scale = atoi(fs) != 0 ? n_groups : 1;
        e->groups.push_back(c);
        if (hipStreamCreateWithFlags(&c->group_stream, hipStreamNonBlocking) != hipSuccess ||
            hipEventCreateWithFlags(&c->group_done, hipEventDisableTiming) != hipSuccess) {
            destroy_groups(e);
            return fail(e, DYF_ERR_HIP, "row group stream / event creation failed");
        }
    }
    return DYF_OK;
}

int32_t dyf_row_groups(const dyf_engine* e) { return e ? std::max<int>(1, (int)e->groups.size()) : 0; }

dyf_status dyf_seed(dyf_engine* e, uint64_t seed) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    // seed words and both counters; the row offset (word 3) belongs to dyf_set_row_offset
    const uint32_t st[3] = {(uint32_t)(seed & 0xffffffffu), (uint32_t)(seed >> 32), 0u};
    const uint32_t zero = 0u;
    // rollouts read and bump rng_state on the caller's / the engine's capture stream, which may be non-blocking streams the
    // null-stream copy below is not ordered against: let everything in flight finish first (re-seeding is rare)
    HIP_TRY(e, hipDeviceSynchronize());
    HIP_TRY(e, hipMemcpy(e->rng_state, st, sizeof(st), hipMemcpyHostToDevice));
    HIP_TRY(e, hipMemcpy(e->rng_state + 4, &zero, sizeof(zero), hipMemcpyHostToDevice));
    return DYF_OK;
}

dyf_status dyf_set_row_offset(dyf_engine* e, uint32_t first_row) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    if (e->row_offset_known && e->row_offset == first_row) return DYF_OK;  // a rank sets the same offset before every predict call
    HIP_TRY(e, hipDeviceSynchronize());  // as dyf_seed: in-flight rollouts on other streams read this word
    HIP_TRY(e, hipMemcpy(e->rng_state + 3, &first_row, sizeof(first_row), hipMemcpyHostToDevice));
    e->row_offset = first_row;
    e->row_offset_known = true;
    return DYF_OK;
}

static dyf_status load_weights_one(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                                   const float* const* data, const int64_t* const* shapes, const int32_t* ndims);

dyf_status dyf_load_weights(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                            const float* const* data, const int64_t* const* shapes, const int32_t* ndims) {
    dyf_status ts = load_weights_one(e, which, n_tensors, names, data, shapes, ndims);
    if (ts != DYF_OK || !e) return ts;
    for (dyf_engine* c : e->groups) {  // every row group packs its own copy
        ts = load_weights_one(c, which, n_tensors, names, data, shapes, ndims);
        if (ts != DYF_OK) return fail(e, ts, "row group: " + c->err);
    }
    return DYF_OK;
}

static dyf_status load_weights_one(dyf_engine* e, int32_t which, int32_t n_tensors, const char* const* names,
                                   const float* const* data, const int64_t* const* shapes, const int32_t* ndims) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 1) return fail(e, DYF_ERR_INVALID_ARGUMENT, "net must be 0 (forecaster) or 1 (interpolator)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    Net& n = e->net[which];
    // a reload replaces the previous packed weights (and invalidates the plan's coefficient tables and captured graphs)
    e->plan.set = false;
    n.loaded = false;
    if (!e->net_allocs[which].empty()) {
        HIP_TRY(e, hipDeviceSynchronize());
        release_allocs(e->net_allocs[which]);
    }
    AllocScope scope(e, &e->net_allocs[which]);
    std::map<std::string, TensorView> sd;
    for (int i = 0; i < n_tensors; ++i) {
        TensorView v;
        v.data = data[i];
        v.shape.assign(shapes[i], shapes[i] + ndims[i]);
        sd[names[i]] = v;
    }
    if (n.sc) return sc_load_weights(e, n, sd);
    if (n.rn) {
        dyf_status rs = rn_load_weights(e, n, sd);
        if (rs != DYF_OK) return rs;
        n.loaded = true;
        n.table_of_time.clear();
        n.ntables = 0;
        e->plan.set = false;
        if (e->is_group_child) return DYF_OK;  // row groups only sample: no fp32 training copy, no gradient buffers
        return rn_train_store_weights(e, which, sd);  // fp32 copy in the training layout (train_resnet.inc)
    }
    std::string missing;
    auto get = [&](const std::string& key, std::vector<int64_t> want) -> const TensorView* {
        auto it = sd.find(key);
        if (it == sd.end()) {
            if (missing.empty()) missing = "missing tensor '" + key + "' in state_dict";
            return nullptr;
        }
        if (it->second.shape != want) {
            if (missing.empty()) missing = "tensor '" + key + "' has an unexpected shape";
            return nullptr;
        }
        return &it->second;
    };
    auto vec = [](const TensorView* t) { return std::vector<float>(t->data, t->data + t->numel()); };
    const int64_t d = n.dim, td = n.tdim;
#define NEED(var, key, ...)                                              \
    const TensorView* var = get(key, std::vector<int64_t>{__VA_ARGS__}); \
    if (!var) return fail(e, DYF_ERR_INVALID_ARGUMENT, missing)
#define UP(dst, hostvec)                                  \
    do {                                                  \
        dyf_status _s = dev_upload(e, &(dst), (hostvec)); \
        if (_s != DYF_OK) return _s;                      \
    } while (0)

    if (n.cfg.with_time_emb) {
        NEED(w1, "time_emb_mlp.1.weight", td, d);
        NEED(b1, "time_emb_mlp.1.bias", td);
        NEED(w2, "time_emb_mlp.3.weight", td, td);
        NEED(b2, "time_emb_mlp.3.bias", td);
        UP(n.t_w1, vec(w1)); UP(n.t_b1, vec(b1)); UP(n.t_w2, vec(w2)); UP(n.t_b2, vec(b2));
    }
    {
        NEED(sw, "init_conv.weight", d, (int64_t)n.cin_total, 1, 1);
        NEED(sb, "init_conv.bias", d);
        UP(n.stem_w, vec(sw)); UP(n.stem_b, vec(sb));
    }
    std::vector<float> film_w((size_t)2 * n.total_c * n.tdim, 0.0f), film_b((size_t)2 * n.total_c, 0.0f);
    std::vector<float> norm_a(n.total_c, 1.0f), norm_c(n.total_c, 0.0f);
    std::vector<int> blk_of(n.total_c), blk_off(12), blk_cout(12);
    for (int i = 0; i < 12; ++i) {
        UBlock& b = n.blk[i];
        const std::string pre = (i < 6 ? "input_ops." + std::to_string(i) : "output_ops." + std::to_string(i - 6));
        const std::string conv = pre + ".ops." + (b.transposed ? "1" : "0");
        const std::string norm = pre + ".ops." + (b.transposed ? "2" : "1");
        NEED(cw, conv + ".weight", (int64_t)b.cout, (int64_t)b.cin, (int64_t)b.k, (int64_t)b.k);
        NEED(cb, conv + ".bias", (int64_t)b.cout);
        NEED(nw, norm + ".weight", (int64_t)b.cout);
        NEED(nbias, norm + ".bias", (int64_t)b.cout);
        blk_off[i] = b.film_off;
        blk_cout[i] = b.cout;
        for (int c = 0; c < b.cout; ++c) blk_of[b.film_off + c] = i;
        // pack [cout][cin][kh][kw] fp32 -> [cout][tap][cin] bf16 (K-contiguous rows for the implicit GEMM)
        const int taps = b.k * b.k;
        std::vector<el16_t> pk((size_t)b.cout * taps * b.cin);
        for (int co = 0; co < b.cout; ++co)
            for (int ci = 0; ci < b.cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    pk[((size_t)co * taps + t) * b.cin + ci] = f32_to_el16(cw->data[((size_t)co * b.cin + ci) * taps + t]);
        { dyf_status _s = upload_conv_weights(e, &b.wpk, pk, b.cout, taps, b.cin); if (_s != DYF_OK) return _s; }
        if (i == 0 && b.k == 4 && n.cin_total + 1 <= 16 && b.cout % 64 == 0 && n.uh % 2 == 0 && n.uw % 2 == 0) {
            // compose_stem_enc0: W'[co][kh][kw][c] = sum_d Wenc0[co][d][kh][kw] * Winit[d][c]; channel cin_total carries
            // init_conv's bias (its input is the 1-inside-the-image indicator); channels up to 16 are zero
            const TensorView* sw = &sd["init_conv.weight"];
            const TensorView* sb = &sd["init_conv.bias"];
            std::vector<el16_t> fw((size_t)b.cout * 16 * 16);
            for (int co = 0; co < b.cout; ++co)
                for (int t = 0; t < 16; ++t)
                    for (int c = 0; c < 16; ++c) {
                        double v = 0.0;
                        if (c <= n.cin_total)
                            for (int d2 = 0; d2 < n.dim; ++d2) {
                                const double we = cw->data[((size_t)co * b.cin + d2) * 16 + t];
                                v += we * (c < n.cin_total ? (double)sw->data[(size_t)d2 * n.cin_total + c] : (double)sb->data[d2]);
                            }
                        fw[((size_t)co * 16 + t) * 16 + c] = f32_to_el16((float)v);
                    }
            { dyf_status _s = upload_conv_weights(e, &n.enc0_fused_w, fw, b.cout, 4, 64); if (_s != DYF_OK) return _s; }
            if (b.cout == 64 || b.cout == 128) {  // fragments of the persistent enc0 kernel (conv_enc0_stem.hip)
                std::vector<el16_t> pf(fw.size());
                pack_enc0_stem_frag(fw.data(), b.cout, pf.data());
                el16_t* frag = nullptr;
                UP(frag, pf);
                conv_register_halo3_frag(n.enc0_fused_w, frag);
            }
            n.stem_fused = true;
        }
        if (b.transposed && b.k == 3) {
            std::vector<el16_t> pu((size_t)4 * b.cout * 16 * b.cin);
            pack_up2x_weights(cw->data, b.cout, b.cin, pu.data());
            UP(b.wpk_up, pu);
            if (b.cin % 64 == 0 && b.cout % 64 == 0) {  // MFMA fragment order for the halo kernel
                std::vector<el16_t> pf(pu.size());
                pack_up2x_frag(pu.data(), b.cout, b.cin, pf.data());
                UP(b.wpk_up_frag, pf);
            }
            // The readout (sparse transposed conv + final resample, readout kernels in kernels.hip) reads only the columns
            // of the last decoder block that the final bilinear interpolation touches: for the NS grid (42 native columns
            // from a 512-wide transposed-conv output) 104 of 256.  Plan the column lists of the sparse halo form.
            const bool sparse_ok = !(dyf_form("DYF_SPARSE_DEC5") && atoi(dyf_form("DYF_SPARSE_DEC5")) == 0);  // read per upload
            if (i == 11 && b.wpk_up_frag && sparse_ok) {
                const int iw = b.out_w, tw = 2 * iw, ow = e->cfg.width;
                std::vector<uint8_t> needed(iw, 0);
                for (int ox = 0; ox < ow; ++ox) {
                    // the readout's bilinear_coord (common.h) in double, both neighbours of a near-integer coordinate
                    double src = ((double)ox + 0.5) * ((double)tw / (double)ow) - 0.5;
                    if (src < 0.0) src = 0.0;
                    if (n.cfg.outer_nearest) src = (double)ox * ((double)tw / (double)ow);  // floor(dst * scale): the one column read
                    for (double eps : {-1e-3, 1e-3}) {
                        int v0 = (int)std::floor(std::max(0.0, src + eps));
                        v0 = std::min(v0, tw - 1);
                        for (int v : {v0, std::min(v0 + 1, tw - 1)}) {
                            const int jh = (v + 1) >> 1;  // k4/s2/p1 transposed conv: input columns jh and jh - 1
                            for (int j : {jh, jh - 1})
                                if (j >= 0 && j < iw) needed[j] = 1;
                        }
                    }
                }
                std::vector<int16_t> cols, cbase, cidx, cmap;
                int nt = 0, nv0 = 0, nv1 = 0;
                // the compact tensor is read by the MFMA form of the readout only (dim 64, <= 4 output channels)
                // list tiles of 32 slots for the rows form of the halo kernel (conv_halo_rows.hip), 16 for conv_up_halo_kernel<1>
                const bool rows_env = !(dyf_form("DYF_HALO_ROWS") && atoi(dyf_form("DYF_HALO_ROWS")) == 0);
                int slots = rows_env && (b.out_h / 2) % 4 == 0 ? conv_halo_rows_slots() : 16;
                bool planned = false;
                int mix[3] = {0, 0, 0};
                // mixed list tiling (no padded MFMA lanes: 52 entries = 3 x 16 + 4) first; DYF_SPARSE_MIXED=0: uniform tiles only
                const bool mixed_env = !(dyf_form("DYF_SPARSE_MIXED") && atoi(dyf_form("DYF_SPARSE_MIXED")) == 0);
                if (n.dim == 64 && n.cfg.out_channels <= 4 && mixed_env && slots == conv_halo_rows_slots() &&
                    plan_up_sparse_columns_mixed(needed, iw / 2, b.out_h / 2, cols, cbase, cidx, cmap, mix, nv0, nv1)) {
                    planned = true;
                    nt = mix[0] + mix[1] + mix[2];
                }
                if (!planned && n.dim == 64 && n.cfg.out_channels <= 4) {
                    mix[0] = mix[1] = mix[2] = 0;
                    planned = plan_up_sparse_columns(needed, iw / 2, cols, cbase, cidx, cmap, nt, nv0, nv1, slots);
                    if (!planned && slots != 16) {
                        slots = 16;
                        planned = plan_up_sparse_columns(needed, iw / 2, cols, cbase, cidx, cmap, nt, nv0, nv1, slots);
                    }
                }
                if (planned) {
                    UP(b.up_cols, cols);
                    UP(b.up_cbase, cbase);
                    UP(b.up_cidx, cidx);
                    UP(b.up_col_map, cmap);
                    b.up_ntiles = nt; b.up_npad = nt * slots; b.up_nvalid0 = nv0; b.up_nvalid1 = nv1;
                    b.up_mix[0] = mix[0]; b.up_mix[1] = mix[1]; b.up_mix[2] = mix[2];
                    if (mix[0] | mix[1] | mix[2]) b.up_npad = 32 * mix[0] + 16 * mix[1] + 4 * mix[2];
                    b.up_wo_store = nv0 + nv1;
                }
            }
        }
        if (!b.gn) {  // eval-mode BatchNorm2d folded with the conv bias: y = conv*a + c
            NEED(rm, norm + ".running_mean", (int64_t)b.cout);
            NEED(rv, norm + ".running_var", (int64_t)b.cout);
            for (int c = 0; c < b.cout; ++c) {
                const double a = (double)nw->data[c] / std::sqrt((double)rv->data[c] + 1e-5);
                norm_a[b.film_off + c] = (float)a;
                norm_c[b.film_off + c] = (float)((double)nbias->data[c] + ((double)cb->data[c] - (double)rm->data[c]) * a);
            }
        } else {  // GroupNorm: conv epilogue only adds the bias; the FiLM table carries (1+scale, shift)
            UP(b.gamma, vec(nw)); UP(b.beta, vec(nbias));
            UP(b.static_a, std::vector<float>(b.cout, 1.0f));
            UP(b.static_c, vec(cb));
        }
        if (n.cfg.with_time_emb) {
            NEED(fw, pre + ".time_mlp.1.weight", (int64_t)2 * b.cout, td);
            NEED(fb, pre + ".time_mlp.1.bias", (int64_t)2 * b.cout);
            std::copy(fw->data, fw->data + fw->numel(), film_w.begin() + (size_t)2 * b.film_off * n.tdim);
            std::copy(fb->data, fb->data + fb->numel(), film_b.begin() + (size_t)2 * b.film_off);
        }
    }
    UP(n.film_w, film_w); UP(n.film_b, film_b); UP(n.norm_a, norm_a); UP(n.norm_c, norm_c);
    UP(n.blk_of, blk_of); UP(n.blk_off, blk_off); UP(n.blk_cout, blk_cout);
    {
        const int64_t oc = n.cfg.out_channels;
        NEED(rw, "readout.0.weight", d, oc, 4, 4);
        NEED(rb, "readout.0.bias", oc);
        std::vector<float> pk((size_t)16 * n.dim * oc);
        for (int ci = 0; ci < n.dim; ++ci)
            for (int co = 0; co < oc; ++co)
                for (int t = 0; t < 16; ++t) pk[((size_t)t * n.dim + ci) * oc + co] = rw->data[((size_t)ci * oc + co) * 16 + t];
        UP(n.ro_w, pk); UP(n.ro_b, vec(rb));
        if (n.dim == 64 && oc <= 4) {  // MFMA fragments: lane (m = lane & 15, kg = lane >> 4) of (tap, half): W[tap][half*32 + kg*8 + e][m]
            std::vector<el16_t> wf((size_t)16 * 2 * 64 * 8, 0);
            for (int t = 0; t < 16; ++t)
                for (int h2 = 0; h2 < 2; ++h2)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, kg = lane >> 4;
                        for (int el = 0; el < 8; ++el)
                            if (m < oc)
                                wf[(((size_t)t * 2 + h2) * 64 + lane) * 8 + el] =
                                    f32_to_el16(pk[((size_t)t * n.dim + h2 * 32 + kg * 8 + el) * oc + m]);
                    }
            UP(n.ro_wfrag, wf);
            // tap tables of the readout for this engine's geometry and the storage layout of the last decoder block's output
            // (compact when its sparse column lists were planned above)
            const UBlock& lb = n.blk[11];
            dyf_status _s = dev_alloc(e, &n.ro_row_tab, (size_t)2 * e->cfg.height);
            if (_s == DYF_OK) _s = dev_alloc(e, &n.ro_col_tab, (size_t)2 * e->cfg.width);
            if (_s != DYF_OK) return _s;
            ReadoutArgs ta{};
            ta.ih = lb.out_h; ta.iw = lb.out_w; ta.oh = e->cfg.height; ta.ow = e->cfg.width; ta.nearest = n.cfg.outer_nearest;
            n.ro_tab_sparse = lb.up_cols != nullptr;
            ta.iw_store = n.ro_tab_sparse ? lb.up_wo_store : lb.out_w;
            ta.col_map = n.ro_tab_sparse ? lb.up_col_map : nullptr;
            ta.row_tab = n.ro_row_tab; ta.col_tab = n.ro_col_tab;
            HIP_TRY(e, launch_readout_tables(ta, nullptr));
            HIP_TRY(e, hipDeviceSynchronize());
        }
    }
#undef NEED
#undef UP
    n.loaded = true;
    n.table_of_time.clear();
    n.ntables = 0;
    e->plan.set = false;  // coefficient tables depend on the weights
    if (e->is_group_child) return DYF_OK;
    return train_store_weights(e, which, sd);  // fp32 copy in the training layout (train.hip)
}

dyf_status dyf_net_flops(const dyf_engine* e, int32_t which, double* flops) {
    if (!e || !flops || which < 0 || which > 1) return DYF_ERR_INVALID_ARGUMENT;
    *flops = e->net[which].flops_per_sample;
    return DYF_OK;
}

dyf_status dyf_net_flops_executed(const dyf_engine* e, int32_t which, double* flops) {
    if (!e || !flops || which < 0 || which > 1) return DYF_ERR_INVALID_ARGUMENT;
    const Net& n = e->net[which];
    double f = n.flops_per_sample;
    if (!n.rn && !n.sc && n.loaded) {
        const UBlock& l = n.blk[11];
        if (l.up_cols)  // sparse last decoder block: only the output columns the readout reads
            f -= 2.0 * l.out_h * (double)(l.out_w - (l.up_nvalid0 + l.up_nvalid1)) * l.cout * l.cin * l.k * l.k;
        if (n.stem_fused) {  // init_conv composed into enc0: a 4 (kh) x 64 (4 pixels x 16 channels) contraction per output
            const UBlock& b0 = n.blk[0];
            f -= 2.0 * n.uh * n.uw * (double)n.cin_total * n.dim;
            f -= 2.0 * b0.out_h * b0.out_w * (double)b0.cout * (b0.cin * b0.k * b0.k - 256.0);
        }
        if (n.ro_wfrag)  // readout evaluated at the 4 transposed-conv outputs the final bilinear resample reads
            f -= 2.0 * 16.0 * n.dim * n.cfg.out_channels * ((double)l.out_h * l.out_w - (double)e->cfg.height * e->cfg.width);
    }
    *flops = f;
    return DYF_OK;
}

// A fused GroupNorm conv (gn_fused.h) whose granule sweep timed out raised the engine's host-visible error word and NaN-poisoned
// its output.  The word is looked at (a plain host read, no synchronisation) at the head of every forward / sampling entry point
// and by dyf_poll_errors, which the caller runs once the call's work has completed: the engine then drops its captured graphs and
// keeps to the three-kernel GroupNorm path from now on, and the call that looks fails, naming what happened.
static bool gn_fuse_live(const dyf_engine* e) {
    auto live = [](const dyf_engine* x) {
        if (x->gn_err_host == nullptr || x->gn_fuse_disabled) return false;
        return (x->net[0].rn != nullptr) || (x->net[1].rn != nullptr);
    };
    if (live(e)) return true;
    for (const dyf_engine* c : e->groups)
        if (live(c)) return true;
    return false;
}

static dyf_status gn_fuse_check(dyf_engine* e, bool earlier_call = true) {
    bool hit = false, slow = false;
    auto one = [&](dyf_engine* x) {
        if (x->gn_err_host && ((volatile uint32_t*)x->gn_err_host)[0] != 0u) hit = true;
        if (x->gn_err_host && ((volatile uint32_t*)x->gn_err_host)[1] != 0u) slow = true;
    };
    one(e);
    for (dyf_engine* c : e->groups) one(c);
    if (!hit && !slow) return DYF_OK;
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    one(e);  // (a time-out may have been raised while the device drained)
    for (dyf_engine* c : e->groups) one(c);
    auto disable = [](dyf_engine* x) {
        if (x->gn_err_host) ((volatile uint32_t*)x->gn_err_host)[0] = ((volatile uint32_t*)x->gn_err_host)[1] = 0u;
        x->gn_fuse_disabled = true;
        ++x->gn_fuse_downgrades;
        for (auto& kv : x->graphs) {
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
            if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
        }
        x->graphs.clear();
    };
    disable(e);
    for (dyf_engine* c : e->groups) disable(c);
    if (getenv("DYF_VERBOSE") && atoi(getenv("DYF_VERBOSE")) != 0)
        fprintf(stderr, "[dyffusion_hip] engine %p leaves the fused GroupNorm path (%s): captured graphs dropped, three-kernel GroupNorm from now on\n",
                (void*)e, hit ? "a granule sweep timed out" : "a granule sweep needed > 1 ms: the GPU is shared");
    // slow only: every sweep did match (results are correct) but took > 1 ms -- the sample's workgroups were not co-scheduled (a
    // GPU shared with another process): later calls take the three-kernel path, nothing to report
    if (!hit) return DYF_OK;
    return fail(e, DYF_ERR_STATE,
                earlier_call
                    ? "a fused GroupNorm convolution of an EARLIER call timed out waiting for its sample's statistics (that call's output was "
                      "NaN-poisoned); the engine now runs the un-fused GroupNorm kernels -- repeat the call"
                    : "a fused GroupNorm convolution timed out waiting for its sample's statistics: the output of the call(s) submitted since "
                      "the last successful dyf_poll_errors is NaN-poisoned; the engine now runs the un-fused GroupNorm kernels -- repeat the call");
}

// Asynchronous failures of work ALREADY SUBMITTED (every entry point here only enqueues): a caller that wants the failing call
// itself to fail -- the Python wrappers do -- polls after it.  synchronize != 0 waits for the device first; an engine that has no
// live fused-GroupNorm form returns at once without waiting (nothing here can fail asynchronously).
dyf_status dyf_poll_errors(dyf_engine* e, int32_t synchronize) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (!gn_fuse_live(e)) return DYF_OK;
    if (synchronize) {
        // only the stream the last call was enqueued on (the row groups' streams join it before the call returns): other streams of
        // the process -- RCCL's, the caller's own async work -- are not stalled.  A stream that is being captured cannot be waited
        // for (and nothing has run yet): the check then reads the error words as they are.
        HIP_TRY(e, hipSetDevice(e->cfg.device));
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(e->poll_stream, &cs) != hipSuccess) { (void)hipGetLastError(); cs = hipStreamCaptureStatusNone; }
        if (cs == hipStreamCaptureStatusNone) HIP_TRY(e, hipStreamSynchronize(e->poll_stream));
    }
    return gn_fuse_check(e, false);
}

// state of the fused GroupNorm path: bit 0 = live (a ResNet-UNet engine that still uses it), *downgrades = times it was left
dyf_status dyf_gn_fuse_state(const dyf_engine* e, int32_t* live, int32_t* downgrades) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (live) *live = gn_fuse_live(e) ? 1 : 0;
    int d = e->gn_fuse_downgrades;
    for (const dyf_engine* c : e->groups) d = std::max(d, c->gn_fuse_downgrades);
    if (downgrades) *downgrades = d;
    return DYF_OK;
}

// test hook (include/dyffusion_hip_testing.h): bound of the granule sweeps in 100 MHz ticks (0 = default 2 s) and, force_timeout != 0,
// make every sweep wait for a tag nobody publishes.  Both travel as kernel arguments: captured graphs are dropped.
dyf_status dyf_debug_gn_fuse(dyf_engine* e, uint32_t timeout_ticks, int32_t force_timeout) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    HIP_TRY(e, hipDeviceSynchronize());
    auto set = [&](dyf_engine* x) {
        x->gn_timeout_ticks = timeout_ticks;
        x->gn_test_tag_xor = force_timeout ? 0x5A000000u : 0u;
        for (auto& kv : x->graphs) {
            if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
            if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
        }
        x->graphs.clear();
    };
    set(e);
    for (dyf_engine* c : e->groups) set(c);
    return DYF_OK;
}

dyf_status dyf_net_forward(dyf_engine* e, int32_t which, const float* inputs_dev, const float* time_dev,
                           const float* condition_dev, float* out_dev, int32_t nb, int32_t dropout_mode,
                           const uint8_t* const* masks_dev, void* stream) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (which < 0 || which > 1) return fail(e, DYF_ERR_INVALID_ARGUMENT, "net must be 0 or 1");
    Net& n = e->net[which];
    if (!n.loaded) return fail(e, DYF_ERR_STATE, "dyf_load_weights has not been called for this network");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    if (!inputs_dev || !out_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "inputs/out must not be null");
    e->poll_stream = (hipStream_t)stream;
    if ((n.cfg.cond_channels > 0) != (condition_dev != nullptr))
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "condition must be given iff num_conditional_channels > 0");
    if (n.cfg.with_time_emb && !time_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "time must be given when with_time_emb");
    if (dropout_mode < 0 || dropout_mode > 2) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dropout_mode must be 0, 1 or 2");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    {
        dyf_status gs = gn_fuse_check(e);
        if (gs != DYF_OK) return gs;
    }
    hipStream_t st = (hipStream_t)stream;
    dyf_status s = compute_coefs(e, n, time_dev, nb, e->ws.coef_a, e->ws.coef_c, st);
    if (s != DYF_OK) return s;
    // channel order of the stem: unet_simple cat[inputs, condition] (unet_simple.py:184), unet.Unet cat[condition, x] (unet.py:269)
    Source srcs[2] = {{inputs_dev, n.cfg.in_channels}, {condition_dev, n.cfg.cond_channels}};
    if (n.rn && condition_dev) std::swap(srcs[0], srcs[1]);
    FwdOpts o{e->ws.coef_a, e->ws.coef_c, n.total_c, dropout_mode, masks_dev};
    return net_forward(e, which, srcs, condition_dev ? 2 : 1, nb, o, out_dev, st);
}

// ------------------------------------------------------------------------------------------------ sampler
dyf_status dyf_set_plan(dyf_engine* e, const dyf_plan* p) {
    if (!e || !p) return DYF_ERR_INVALID_ARGUMENT;
    if (!e->net[0].loaded || !e->net[1].loaded) return fail(e, DYF_ERR_STATE, "load both networks' weights before dyf_set_plan");
    if (p->n_steps < 1 || !p->steps) return fail(e, DYF_ERR_INVALID_ARGUMENT, "plan needs at least one step");
    {   // channel bookkeeping of the pair: forecaster (C -> C), interpolator ((window+1)*C -> C), interpolation.py:48-51
        const int C = e->net[DYF_NET_FORECASTER].cfg.in_channels;
        if (e->net[DYF_NET_FORECASTER].cfg.out_channels != C || e->net[DYF_NET_INTERPOLATOR].cfg.out_channels != C)
            return fail(e, DYF_ERR_INVALID_ARGUMENT, "forecaster in/out and interpolator out channels must all equal C");
        const int wC = e->net[DYF_NET_INTERPOLATOR].cfg.in_channels - C;
        if (wC < C || wC % C != 0)
            return fail(e, DYF_ERR_INVALID_ARGUMENT, "interpolator in_channels must be (window+1)*C (interpolation.py:48-51)");
        e->C = C;
        e->wC = wC;
        e->Cs = e->net[DYF_NET_INTERPOLATOR].cfg.cond_channels;  // the interpolator only ever sees the static condition
    }
    if (p->n_out_slots < 1) return fail(e, DYF_ERR_INVALID_ARGUMENT, "n_out_slots must be >= 1");
    const int fc = p->forward_conditioning;
    const int want_cond = e->Cs + (fc == DYF_FCOND_NONE ? 0 : e->wC);
    if (e->net[DYF_NET_FORECASTER].cfg.cond_channels != want_cond)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "forecaster cond_channels inconsistent with forward_conditioning");
    for (int i = 0; i < p->n_steps; ++i)
        if (p->steps[i].out_slot >= p->n_out_slots) return fail(e, DYF_ERR_INVALID_ARGUMENT, "out_slot out of range");
    for (int i = 0; i < p->n_refine; ++i)
        if (p->refine_slots[i] < 0 || p->refine_slots[i] >= p->n_out_slots || !(p->refine_times[i] > 0.0f))
            return fail(e, DYF_ERR_INVALID_ARGUMENT, "refine slot/time out of range");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    PlanHost& ph = e->plan;
    ph.steps.assign(p->steps, p->steps + p->n_steps);
    ph.refine_times.assign(p->refine_times, p->refine_times + p->n_refine);
    ph.refine_slots.assign(p->refine_slots, p->refine_slots + p->n_refine);
    ph.hdr = *p;
    ph.hdr.steps = nullptr; ph.hdr.refine_times = nullptr; ph.hdr.refine_slots = nullptr;
    if (!e->s_init) {  // sampler state, allocated once (fp32 NCHW)
        const size_t nbm = (size_t)e->cfg.max_batch, hw = (size_t)e->cfg.height * e->cfg.width;
        dyf_status s;
        if ((s = dev_alloc(e, &e->s_init, nbm * e->wC * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_static, nbm * std::max(1, e->Cs) * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_xs, nbm * e->C * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_x0hat, nbm * e->C * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_next, nbm * e->C * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_cur, nbm * e->C * hw)) != DYF_OK) return s;
        if ((s = dev_alloc(e, &e->s_noisy, nbm * e->wC * hw)) != DYF_OK) return s;
    }
    // forecast stack
    if (p->n_out_slots > e->stack_slots) {
        if (e->s_stack) {
            HIP_TRY(e, hipDeviceSynchronize());
            e->allocs.erase(std::remove(e->allocs.begin(), e->allocs.end(), (void*)e->s_stack), e->allocs.end());
            (void)hipFree(e->s_stack);
            e->s_stack = nullptr;
        }
        dyf_status s = dev_alloc(e, &e->s_stack, (size_t)p->n_out_slots * e->cfg.max_batch * e->C * e->cfg.height * e->cfg.width);
        if (s != DYF_OK) return s;
        e->stack_slots = p->n_out_slots;
    }
    // coefficient tables: the time value is the same for the whole batch inside the loop (dyffusion.py:360,372), so
    // every (network, time) pair of the plan is evaluated once here instead of once per forward
    ph.set = false;
    HIP_TRY(e, hipDeviceSynchronize());  // graphs of the previous plan may still be running on their tables
    for (auto& kv : e->graphs) {
        if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
        if (kv.second.graph) (void)hipGraphDestroy(kv.second.graph);
    }
    e->graphs.clear();
    release_allocs(e->plan_allocs);
    AllocScope scope(e, &e->plan_allocs);
    for (int w = 0; w < 2; ++w) {
        Net& n = e->net[w];
        n.table_of_time.clear();
        std::vector<float> times;
        auto add = [&](float t) {
            if (!n.table_of_time.count(t)) {
                n.table_of_time[t] = (int)times.size();
                times.push_back(t);
            }
        };
        if (w == DYF_NET_FORECASTER) {
            for (auto& s : ph.steps) add(s.forecaster_time);
        } else {
            for (auto& s : ph.steps) {
                if (s.i_next >= 0.0f) add(s.i_next);
                if (s.i_cur >= 0.0f) add(s.i_cur);
            }
            for (float t : ph.refine_times) add(t);
        }
        if (times.empty()) times.push_back(0.0f);
        n.ntables = (int)times.size();
        dyf_status s = dev_alloc(e, &n.tables, (size_t)n.ntables * 2 * n.total_c);
        if (s != DYF_OK) return s;
        float* tdev = nullptr;
        s = dev_upload(e, &tdev, times);
        if (s != DYF_OK) return s;
        for (int i = 0; i < n.ntables; ++i) {
            float* A = n.tables + (size_t)i * 2 * n.total_c;
            s = compute_coefs(e, n, tdev + i, 1, A, A + n.total_c, 0);
            if (s != DYF_OK) return s;
        }
    }
    {   // FiLM coefficient rows of the refinement pass in REFINE ORDER, contiguous: a batched refinement launch (run_plan) covers
        // k consecutive prediction times with one forward over k * nb rows and reads row (batch row / nb) of this table
        Net& I = e->net[DYF_NET_INTERPOLATOR];
        const size_t row = (size_t)2 * I.total_c;
        e->refine_coef = nullptr;
        if (!ph.refine_times.empty()) {
            dyf_status s = dev_alloc(e, &e->refine_coef, ph.refine_times.size() * row);
            if (s != DYF_OK) return s;
            for (size_t r = 0; r < ph.refine_times.size(); ++r)
                HIP_TRY(e, hipMemcpy(e->refine_coef + r * row, I.tables + (size_t)I.table_of_time.at(ph.refine_times[r]) * row,
                                     row * sizeof(float), hipMemcpyDeviceToDevice));
        }
    }
    {   // ... and the row pairs (i_next, i_cur) of every cold-sampling step's paired interpolator call, staged ONCE here: run_plan used
        // to copy the two rows next to each other with two memcpy nodes per step (~12 us of a 250 us one-row step)
        Net& I = e->net[DYF_NET_INTERPOLATOR];
        const size_t row = (size_t)2 * I.total_c;
        e->pair_coef = nullptr;
        if (!ph.steps.empty() && I.tables && !I.rn && !I.sc) {
            dyf_status s = dev_alloc(e, &e->pair_coef, ph.steps.size() * 2 * row);
            if (s != DYF_OK) return s;
            for (size_t k = 0; k < ph.steps.size(); ++k) {
                const dyf_plan_step& stp = ph.steps[k];
                if (!(stp.i_next >= 0.0f && stp.i_cur >= 0.0f)) continue;
                HIP_TRY(e, hipMemcpy(e->pair_coef + (2 * k) * row, I.tables + (size_t)I.table_of_time.at(stp.i_next) * row, row * sizeof(float),
                                     hipMemcpyDeviceToDevice));
                HIP_TRY(e, hipMemcpy(e->pair_coef + (2 * k + 1) * row, I.tables + (size_t)I.table_of_time.at(stp.i_cur) * row, row * sizeof(float),
                                     hipMemcpyDeviceToDevice));
            }
        }
    }
    HIP_TRY(e, hipDeviceSynchronize());
    ph.set = true;
    e->last_groups = 0;
    for (dyf_engine* c : e->groups) {
        dyf_status cs = dyf_set_plan(c, p);
        if (cs != DYF_OK) {
            ph.set = false;
            return fail(e, cs, "row group: " + c->err);
        }
    }
    return DYF_OK;
}

dyf_status dyf_plan_forward_counts(const dyf_engine* e, int32_t* nf, int32_t* ni) {
    if (!e || !nf || !ni || !e->plan.set) return DYF_ERR_STATE;
    const PlanHost& ph = e->plan;
    int f = 0, i = 0;
    for (auto& s : ph.steps) {
        ++f;
        if (s.i_next >= 0.0f) ++i;
        const bool plain_last = s.is_last && !ph.hdr.cold_for_last_step;
        if (ph.hdr.sampling_cold && !plain_last && s.i_cur >= 0.0f) ++i;
    }
    i += (int)ph.refine_times.size();
    *nf = f; *ni = i;
    return DYF_OK;
}

}  // extern "C"

namespace {

struct MaskCursor {
    const uint8_t* const* masks;
    size_t pos = 0;
    const uint8_t* const* take(bool on, int n_sites) {
        if (!masks || !on) return nullptr;
        const uint8_t* const* p = masks + pos;
        pos += n_sites;
        return p;
    }
};

dyf_status run_plan(dyf_engine* e, int nb, const uint8_t* const* masks, const float* noise_dev, hipStream_t st) {
    const PlanHost& ph = e->plan;
    Net& F = e->net[DYF_NET_FORECASTER];
    Net& I = e->net[DYF_NET_INTERPOLATOR];
    const int H = e->cfg.height, W = e->cfg.width;
    const size_t field = (size_t)nb * e->C * H * W;
    const size_t init_el = (size_t)nb * e->wC * H * W;
    const size_t fbytes = field * sizeof(float);
    MaskCursor cur{masks};
    const bool inject = masks != nullptr;
    const int i_mode = ph.hdr.interpolator_dropout ? (inject ? 2 : 1) : 0;
    const int f_mode = ph.hdr.forecaster_dropout ? (inject ? 2 : 1) : 0;

    auto interp = [&](float t, const float* x_last, float* out) -> dyf_status {
        const int ti = I.table_of_time.at(t);
        const float* A = I.tables + (size_t)ti * 2 * I.total_c;
        Source srcs[3] = {{e->s_init, e->wC}, {x_last, e->C}, {e->s_static, e->Cs}};
        int ns = e->Cs > 0 ? 3 : 2;
        if (I.rn && e->Cs > 0) {  // unet.Unet puts the condition first
            srcs[0] = {e->s_static, e->Cs}; srcs[1] = {e->s_init, e->wC}; srcs[2] = {x_last, e->C};
        }
        FwdOpts o{A, A + I.total_c, 0, i_mode, cur.take(i_mode == 2 && I.n_drop_sites > 0, I.n_drop_sites)};
        return net_forward(e, DYF_NET_INTERPOLATOR, srcs, ns, nb, o, out, st);
    };

    // Two interpolator calls with the same inputs and different times as ONE forward over 2 nb rows (rows [0, nb): t_a,
    // rows [nb, 2 nb): t_b): same arithmetic per row, but the small layers and the tile counts of the mid layers see twice
    // the batch.  The FiLM coefficient rows of the two times are staged next to each other; out: [2][nb][C][H][W].
    // Not with injected masks (their layout is one tensor per forward and site) and only for arch unet_simple.
    const bool can_pair = !inject && !I.rn && !I.sc && e->pair_interp;
    auto interp2 = [&](int step, float ta, float tb, const float* x_last, float* out2) -> dyf_status {
        const size_t row = (size_t)2 * I.total_c;
        const float* pair = e->ws.coef_pair;
        if (e->pair_coef) {  // staged at dyf_set_plan
            pair = e->pair_coef + (size_t)(2 * step) * row;
        } else {
            HIP_TRY(e, hipMemcpyAsync(e->ws.coef_pair, I.tables + (size_t)I.table_of_time.at(ta) * row, row * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
            HIP_TRY(e, hipMemcpyAsync(e->ws.coef_pair + row, I.tables + (size_t)I.table_of_time.at(tb) * row, row * sizeof(float),
                                      hipMemcpyDeviceToDevice, st));
        }
        Source srcs[3] = {{e->s_init, e->wC}, {x_last, e->C}, {e->s_static, e->Cs}};
        const int ns = e->Cs > 0 ? 3 : 2;
        FwdOpts o{pair, pair + I.total_c, (int)row, i_mode, nullptr};
        o.src_rows = nb;
        o.coef_div = nb;
        return net_forward(e, DYF_NET_INTERPOLATOR, srcs, ns, 2 * nb, o, out2, st);
    };

    // x_s = initial_condition[:, -C:]  (dyffusion.py:348); rows are (window*C, H, W) blocks -> strided copy per sample
    if (e->wC == e->C) {
        HIP_TRY(e, hipMemcpyAsync(e->s_xs, e->s_init, fbytes, hipMemcpyDeviceToDevice, st));
    } else {
        HIP_TRY(e, hipMemcpy2DAsync(e->s_xs, (size_t)e->C * H * W * sizeof(float),
                                    e->s_init + (size_t)(e->wC - e->C) * H * W, (size_t)e->wC * H * W * sizeof(float),
                                    (size_t)e->C * H * W * sizeof(float), nb, hipMemcpyDeviceToDevice, st));
    }
    int step_idx = 0;
    // log_every_t: slot (step, what) of e->s_log; what 0 = x0_hat, 1 = x_interpolated_s_next, 2 = x_interpolated_s (dyffusion.py:398-406)
    const bool logging = e->log_on && e->s_log != nullptr;
    auto log_to = [&](int what) { return e->s_log + ((size_t)step_idx * 3 + what) * field; };
#define LOG_COPY(what, src) do { if (logging) HIP_TRY(e, hipMemcpyAsync(log_to(what), (src), fbytes, hipMemcpyDeviceToDevice, st)); } while (0)
    if (logging) { e->log_has_cur.assign(ph.steps.size(), 0); e->log_nb = nb; }
    for (const dyf_plan_step& s : ph.steps) {
        // ---- forecaster: x0_hat = F(x_s, enc(s); cond)
        Source fs[3];
        int nf = 0;
        if (!F.rn) fs[nf++] = {e->s_xs, e->C};  // unet_simple: inputs first; unet.Unet: condition first
        if (ph.hdr.forward_conditioning == DYF_FCOND_DATA) {
            fs[nf++] = {e->s_init, e->wC};
        } else if (ph.hdr.forward_conditioning == DYF_FCOND_DATA_NOISE) {
            const float* nz = noise_dev ? noise_dev + (size_t)step_idx * init_el : nullptr;
            HIP_TRY(e, launch_noisy_condition(e->s_noisy, e->s_init, nz, s.tau, (long long)init_el, e->wC * H * W,
                                              e->rng_state, st));
            fs[nf++] = {e->s_noisy, e->wC};
        }
        if (e->Cs > 0) fs[nf++] = {e->s_static, e->Cs};
        if (F.rn) fs[nf++] = {e->s_xs, e->C};
        {
            const int ti = F.table_of_time.at(s.forecaster_time);
            const float* A = F.tables + (size_t)ti * 2 * F.total_c;
            FwdOpts o{A, A + F.total_c, 0, f_mode, cur.take(f_mode == 2 && F.n_drop_sites > 0, F.n_drop_sites)};
            dyf_status r = net_forward(e, DYF_NET_FORECASTER, fs, nf, nb, o, e->s_x0hat, st);
            if (r != DYF_OK) return r;
            LOG_COPY(0, e->s_x0hat);
        }
        // ---- x_next = I(x0, x0_hat, i(s_next))   (dyffusion.py:374-379)
        const float* x_next = e->s_x0hat;
        const bool cold_pair = can_pair && ph.hdr.sampling_cold && !(s.is_last && !ph.hdr.cold_for_last_step) &&
                               s.i_next >= 0.0f && s.i_cur >= 0.0f;
        if (cold_pair) {  // both interpolations of this step in one forward: s_pair = [I(.., s_next) ; I(.., s)]
            dyf_status r = interp2(step_idx, s.i_next, s.i_cur, e->s_x0hat, e->s_pair);
            if (r != DYF_OK) return r;
            LOG_COPY(1, e->s_pair);
            LOG_COPY(2, e->s_pair + field);
            if (logging) e->log_has_cur[step_idx] = 1;
            // (the new x_s goes to its forecast-stack slot in the same pass when the step emits a prediction)
            HIP_TRY(e, launch_cold_update(e->s_xs, e->s_pair + field, e->s_pair, (long long)field, st,
                                          s.out_slot >= 0 ? e->s_stack + (size_t)s.out_slot * field : nullptr));
            if (&s == &ph.steps.back())  // sample_loop's third return value for a truncated schedule (dyffusion.py:424-425)
                HIP_TRY(e, hipMemcpyAsync(e->s_next, e->s_pair, fbytes, hipMemcpyDeviceToDevice, st));
            ++step_idx;
            continue;
        }
        if (s.i_next >= 0.0f) {
            dyf_status r = interp(s.i_next, e->s_x0hat, e->s_next);
            if (r != DYF_OK) return r;
            x_next = e->s_next;
        }
        LOG_COPY(1, x_next);
        // ---- update of x_s  (dyffusion.py:381-393)
        if (ph.hdr.sampling_cold) {
            if (s.is_last && !ph.hdr.cold_for_last_step) {
                // the reference logs its variable x_interpolated_s as it stands: the previous iteration's value
                if (logging && step_idx > 0 && e->log_has_cur[step_idx - 1]) {
                    HIP_TRY(e, hipMemcpyAsync(log_to(2), log_to(2) - 3 * field, fbytes, hipMemcpyDeviceToDevice, st));
                    e->log_has_cur[step_idx] = 1;
                }
                HIP_TRY(e, hipMemcpyAsync(e->s_xs, e->s_x0hat, fbytes, hipMemcpyDeviceToDevice, st));
            } else if (s.i_cur >= 0.0f) {
                dyf_status r = interp(s.i_cur, e->s_x0hat, e->s_cur);
                if (r != DYF_OK) return r;
                LOG_COPY(2, e->s_cur);
                if (logging) e->log_has_cur[step_idx] = 1;
                HIP_TRY(e, launch_cold_update(e->s_xs, e->s_cur, x_next, (long long)field, st));
            } else {  // s == 0: x_s - x_s + x_next
                LOG_COPY(2, e->s_xs);  // x_interpolated_s = x_s (dyffusion.py:385)
                if (logging) e->log_has_cur[step_idx] = 1;
                HIP_TRY(e, hipMemcpyAsync(e->s_xs, x_next, fbytes, hipMemcpyDeviceToDevice, st));
            }
        } else {
            HIP_TRY(e, hipMemcpyAsync(e->s_xs, x_next, fbytes, hipMemcpyDeviceToDevice, st));
        }
        if (s.out_slot >= 0)
            HIP_TRY(e, hipMemcpyAsync(e->s_stack + (size_t)s.out_slot * field, e->s_xs, fbytes, hipMemcpyDeviceToDevice, st));
        ++step_idx;
    }
    // ---- refinement of the intermediate predictions with the final x0_hat (dyffusion.py:408-422).  The h - 1 interpolator calls
    // share their inputs and do not depend on one another: runs of consecutive output slots go out as ONE forward over k * nb rows
    // (row r: prediction time r / nb, the masks of forward counter + r / nb -- exactly those of k separate forwards), written
    // straight into the contiguous [k][nb][C][H][W] block of the forecast stack.  k is bounded by the workspace (2 max_batch rows):
    // an engine created for 80 rows refines a 10-row call in one 150-row launch instead of eight 20-row ones (the small-batch /
    // ensemble-sharded regime); at nb = max_batch it is the pair it always was.  DYF_REFINE_BATCH caps k.
    const int refine_cap = dyf_form("DYF_REFINE_BATCH") ? std::max(1, atoi(dyf_form("DYF_REFINE_BATCH"))) : 1 << 20;  // read per capture
    const int kmax = can_pair ? std::max(1, std::min(refine_cap, 2 * e->cfg.max_batch / nb)) : 1;
    for (size_t r = 0; r < ph.refine_times.size();) {
        size_t k = 1;
        while ((int)k < kmax && r + k < ph.refine_times.size() && ph.refine_slots[r + k] == ph.refine_slots[r] + (int)k) ++k;
        if (k >= 2) {
            const size_t row = (size_t)2 * I.total_c;
            Source srcs[3] = {{e->s_init, e->wC}, {e->s_x0hat, e->C}, {e->s_static, e->Cs}};
            const int ns = e->Cs > 0 ? 3 : 2;
            FwdOpts o{e->refine_coef + r * row, e->refine_coef + r * row + I.total_c, (int)row, i_mode, nullptr};
            o.src_rows = nb;
            o.coef_div = nb;
            dyf_status rs = net_forward(e, DYF_NET_INTERPOLATOR, srcs, ns, (int)k * nb, o, e->s_stack + (size_t)ph.refine_slots[r] * field, st);
            if (rs != DYF_OK) return rs;
        } else {
            dyf_status rs = interp(ph.refine_times[r], e->s_x0hat, e->s_stack + (size_t)ph.refine_slots[r] * field);
            if (rs != DYF_OK) return rs;
        }
        r += k;
    }
#undef LOG_COPY
    return DYF_OK;
}

}  // namespace

extern "C" {

// dyf_sample without the final copy: the forecast stack of the call stays in e->s_stack, [n_out_slots][nb][C][H][W]
static dyf_status sample_into_stack(dyf_engine* e, const float* initial_dev, const float* static_dev, int32_t nb,
                                    const uint8_t* const* masks_dev, const float* noise_dev, void* stream) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (!e->plan.set) return fail(e, DYF_ERR_STATE, "dyf_set_plan has not been called");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    if (!initial_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "initial condition must not be null");
    if (!e->is_group_child) {
        dyf_status gs = gn_fuse_check(e);
        if (gs != DYF_OK) return gs;
    }
    if ((e->Cs > 0) != (static_dev != nullptr))
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "static condition must be given iff the networks take one");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int H = e->cfg.height, W = e->cfg.width;
    e->last_groups = 0;
    const int G = (int)e->groups.size();
    if (G > 1 && !e->log_on && masks_dev == nullptr && noise_dev == nullptr && nb >= 2 * e->group_min_rows) {
        // rows split over the groups: per = ceil(nb / g) rows each (the last one takes the remainder), every share on its own stream
        // Three concurrent groups + the caller's stream use all four hardware queues a HIP process gets: ONE more stream with work
        // (or a live graph) makes two groups share a queue and the 300-row OISST rollout drops from ~3 850 to ~3 100 fields/s, below
        // what two groups deliver with or without company (~3 700).  An engine that owns a communicator lives in a
        // multi-GPU process -- RCCL and torch.distributed bring streams of their own -- so it keeps to two groups.
        const int g_cap = e->comm ? std::min(G, 2) : G;
        int g_use = std::min(g_cap, nb / e->group_min_rows);
        if ((nb + g_use - 1) / g_use > e->groups[0]->cfg.max_batch) g_use = g_cap;  // ceil(nb / min(G, 2)) always fits a group's max_batch
        const int per = (nb + g_use - 1) / g_use, used = (nb + per - 1) / per;
        const size_t row = (size_t)e->C * H * W, slots = (size_t)e->plan.hdr.n_out_slots;
        for (int g = 0; g < used; ++g)  // same seed and stream position as this engine, batch row 0 = global row offset + g * per
            HIP_TRY(e, launch_rng_clone(e->groups[g]->rng_state, e->rng_state, (uint32_t)(g * per), false, st));
        HIP_TRY(e, hipEventRecord(e->group_fork, st));
        // the parent keeps the call's inputs too (dyf_time_kernel_in_rollout / dyf_time_layer_in_rollout re-run the plan on
        // e->s_init / e->s_static: without this copy they would read whatever hipMalloc left there); 29 KB per OISST row
        HIP_TRY(e, hipMemcpyAsync(e->s_init, initial_dev, (size_t)nb * e->wC * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
        if (static_dev)
            HIP_TRY(e, hipMemcpyAsync(e->s_static, static_dev, (size_t)nb * e->Cs * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
        for (int g = 0; g < used; ++g) {
            dyf_engine* c = e->groups[g];
            const int rows = std::min(per, nb - g * per);
            HIP_TRY(e, hipStreamWaitEvent(c->group_stream, e->group_fork, 0));
            dyf_status r = sample_into_stack(c, initial_dev + (size_t)g * per * e->wC * H * W,
                                             static_dev ? static_dev + (size_t)g * per * e->Cs * H * W : nullptr, rows, nullptr,
                                             nullptr, c->group_stream);
            if (r != DYF_OK) return fail(e, r, "row group: " + c->err);
            // the share's stack [slots][rows][C][H][W] into rows [g per, g per + rows) of this engine's [slots][nb][C][H][W]
            HIP_TRY(e, hipMemcpy2DAsync(e->s_stack + (size_t)g * per * row, (size_t)nb * row * sizeof(float), c->s_stack,
                                        (size_t)rows * row * sizeof(float), (size_t)rows * row * sizeof(float), slots,
                                        hipMemcpyDeviceToDevice, c->group_stream));
            HIP_TRY(e, hipEventRecord(c->group_done, c->group_stream));
        }
        for (int g = 0; g < used; ++g) HIP_TRY(e, hipStreamWaitEvent(st, e->groups[g]->group_done, 0));
        HIP_TRY(e, launch_rng_clone(e->rng_state, e->groups[0]->rng_state, 0u, true, st));  // the counters the rollout advanced
        e->last_groups = used;
        e->last_per = per;
        e->last_nb = nb;
        return DYF_OK;
    }
    HIP_TRY(e, hipMemcpyAsync(e->s_init, initial_dev, (size_t)nb * e->wC * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (static_dev)
        HIP_TRY(e, hipMemcpyAsync(e->s_static, static_dev, (size_t)nb * e->Cs * H * W * sizeof(float), hipMemcpyDeviceToDevice, st));
    if (e->log_on) {  // (re)size the log for the current plan
        const size_t need = e->plan.steps.size() * 3 * (size_t)e->cfg.max_batch * e->C * H * W;
        if (need > e->s_log_floats) {
            HIP_TRY(e, hipDeviceSynchronize());
            if (e->s_log) (void)hipFree(e->s_log);
            e->s_log = nullptr;
            e->s_log_floats = 0;
            HIP_TRY(e, hipMalloc((void**)&e->s_log, need * sizeof(float)));
            e->s_log_floats = need;
        }
    }
    const bool graphable = e->cfg.use_graph && !e->log_on && masks_dev == nullptr && noise_dev == nullptr;
    if (!graphable) {
        dyf_status r = run_plan(e, nb, masks_dev, noise_dev, st);
        if (r != DYF_OK) return r;
    } else {
        GraphEntry& g = e->graphs[nb];
        if (!g.exec) {
            HIP_TRY(e, hipStreamBeginCapture(e->cap_stream, hipStreamCaptureModeThreadLocal));
            dyf_status r = run_plan(e, nb, nullptr, nullptr, e->cap_stream);
            hipGraph_t graph = nullptr;
            hipError_t ce = hipStreamEndCapture(e->cap_stream, &graph);
            if (r != DYF_OK) {
                if (graph) (void)hipGraphDestroy(graph);
                e->graphs.erase(nb);
                return r;
            }
            if (ce != hipSuccess) {
                e->graphs.erase(nb);
                return fail(e, DYF_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(ce));
            }
            g.graph = graph;
            HIP_TRY(e, hipGraphInstantiate(&g.exec, g.graph, nullptr, nullptr, 0));
        }
        HIP_TRY(e, hipGraphLaunch(g.exec, st));
    }
    return DYF_OK;
}

dyf_status dyf_sample(dyf_engine* e, const float* initial_dev, const float* static_dev, float* out_dev, int32_t nb,
                      const uint8_t* const* masks_dev, const float* noise_dev, void* stream) {
    if (e && !out_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "out must not be null");
    if (e) e->poll_stream = (hipStream_t)stream;
    dyf_status r = sample_into_stack(e, initial_dev, static_dev, nb, masks_dev, noise_dev, stream);
    if (r != DYF_OK) return r;
    const size_t field = (size_t)nb * e->C * e->cfg.height * e->cfg.width;
    HIP_TRY(e, hipMemcpyAsync(out_dev, e->s_stack, (size_t)e->plan.hdr.n_out_slots * field * sizeof(float),
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DYF_OK;
}

// ---- engine-owned exchange (SURVEY 8b "Ownership", 8e): RCCL communicator + ONE all-gather of the forecast stack -----------
// librccl is opened lazily (dlopen) the first time a communicator is asked for: the library has no link-time dependency on it
// and loads on hosts without RCCL; a copy already mapped into the process (PyTorch's) is reused.
namespace {
struct RcclApi {
    void* lib = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int*) = nullptr;
    ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    std::string err;
};
RcclApi& rccl_api() {
    static RcclApi api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {getenv("DYF_RCCL_LIB"), "librccl.so", "librccl.so.1", "/opt/rocm/lib/librccl.so.1"};
        for (const char* nm : names) {
            if (!nm || !*nm) continue;
            api.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);  // already mapped (torch's copy)?
            if (!api.lib) api.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
            if (api.lib) break;
        }
        if (!api.lib) { api.err = std::string("librccl not found: ") + (dlerror() ? dlerror() : ""); return; }
#define SYM(field, name) api.field = (decltype(api.field))dlsym(api.lib, name); if (!api.field) api.err = std::string("librccl lacks ") + name;
        SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
        SYM(AllGather, "ncclAllGather") SYM(GetErrorString, "ncclGetErrorString") SYM(CommCount, "ncclCommCount")
#undef SYM
    });
    return api;
}
}  // namespace

dyf_status dyf_comm_unique_id(uint8_t* id_out) {
    if (!id_out) return DYF_ERR_INVALID_ARGUMENT;
    RcclApi& api = rccl_api();
    if (!api.err.empty()) return fail(nullptr, DYF_ERR_UNSUPPORTED, api.err);
    ncclUniqueId id;
    ncclResult_t r = api.GetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, DYF_ERR_HIP, std::string("ncclGetUniqueId: ") + api.GetErrorString(r));
    static_assert(sizeof(id) == DYF_COMM_ID_BYTES, "ncclUniqueId is 128 bytes");
    memcpy(id_out, &id, sizeof(id));
    return DYF_OK;
}

dyf_status dyf_comm_init(dyf_engine* e, const uint8_t* unique_id, int32_t rank, int32_t world) {
    if (!e || !unique_id) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return fail(e, DYF_ERR_INVALID_ARGUMENT, "rank outside [0, world)");
    RcclApi& api = rccl_api();
    if (!api.err.empty()) return fail(e, DYF_ERR_UNSUPPORTED, api.err);
    if (e->comm) return fail(e, DYF_ERR_STATE, "the engine already owns a communicator (dyf_comm_destroy first)");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    ncclUniqueId id;
    memcpy(&id, unique_id, sizeof(id));
    ncclComm_t comm = nullptr;
    ncclResult_t r = api.CommInitRank(&comm, world, id, rank);
    if (r != ncclSuccess) return fail(e, DYF_ERR_HIP, std::string("ncclCommInitRank: ") + api.GetErrorString(r));
    e->comm = comm;
    e->comm_rank = rank;
    e->comm_world = world;
    return DYF_OK;
}

dyf_status dyf_comm_count(const dyf_engine* e, int32_t* ranks_out) {
    if (!e || !ranks_out) return DYF_ERR_INVALID_ARGUMENT;
    *ranks_out = 0;
    if (!e->comm) return DYF_OK;  // no communicator: 0 ranks
    RcclApi& api = rccl_api();
    int n = 0;
    if (!api.CommCount || api.CommCount((ncclComm_t)e->comm, &n) != ncclSuccess) return DYF_ERR_HIP;
    *ranks_out = n;
    return DYF_OK;
}

dyf_status dyf_comm_destroy(dyf_engine* e) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    if (e->comm) {
        (void)hipDeviceSynchronize();
        (void)rccl_api().CommDestroy((ncclComm_t)e->comm);
        e->comm = nullptr;
    }
    if (e->gather_recv) {
        (void)hipFree(e->gather_recv);
        e->gather_recv = nullptr;
        e->gather_recv_floats = 0;
    }
    return DYF_OK;
}

dyf_status dyf_sample_gather(dyf_engine* e, const float* initial_dev, const float* static_dev, float* out_full_dev, int32_t nb,
                             int32_t total_rows, void* stream) {
    if (!e || !out_full_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (!e->comm) return fail(e, DYF_ERR_STATE, "dyf_comm_init has not been called");
    const int world = e->comm_world;
    if (total_rows < 1 || (long long)nb * world < total_rows || nb != (total_rows + world - 1) / world)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "every rank samples nb = ceil(total_rows / world) rows");
    e->poll_stream = (hipStream_t)stream;
    dyf_status r = sample_into_stack(e, initial_dev, static_dev, nb, nullptr, nullptr, stream);
    if (r != DYF_OK) return r;
    hipStream_t st = (hipStream_t)stream;
    const size_t row = (size_t)e->C * e->cfg.height * e->cfg.width, slots = (size_t)e->plan.hdr.n_out_slots;
    const size_t send = slots * nb * row;  // floats: the contiguous local stack
    if (e->gather_recv_floats < send * world) {
        if (e->gather_recv) { HIP_TRY(e, hipDeviceSynchronize()); (void)hipFree(e->gather_recv); e->gather_recv = nullptr; }
        HIP_TRY(e, hipMalloc((void**)&e->gather_recv, send * world * sizeof(float)));
        e->gather_recv_floats = send * world;
    }
    RcclApi& api = rccl_api();
    ncclResult_t nr = api.AllGather(e->s_stack, e->gather_recv, send, ncclFloat, (ncclComm_t)e->comm, st);  // ONE collective
    if (nr != ncclSuccess) return fail(e, DYF_ERR_HIP, std::string("ncclAllGather: ") + api.GetErrorString(nr));
    // [world][slots][nb][row] -> [slots][total_rows][row] in global row order (rank r owns rows shard(r); its padding rows drop)
    HIP_TRY(e, launch_gather_unpack(e->gather_recv, out_full_dev, world, (int)slots, nb, total_rows, (long long)row, st));
    return DYF_OK;
}

dyf_status dyf_get_sampler_state(dyf_engine* e, int32_t what, float* out_dev, int32_t nb, void* stream) {
    if (!e || !out_dev) return DYF_ERR_INVALID_ARGUMENT;
    if (!e->plan.set || !e->s_x0hat) return fail(e, DYF_ERR_STATE, "no sampling call has been made yet");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    if (e->last_groups > 0) {  // the most recent call ran on the row groups: its state lives there, share by share
        if (nb != e->last_nb) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size differs from the sampling call's");
        const size_t row = (size_t)e->C * e->cfg.height * e->cfg.width;
        for (int g = 0; g < e->last_groups; ++g) {
            dyf_engine* c = e->groups[g];
            dyf_status r = dyf_get_sampler_state(c, what, out_dev + (size_t)g * e->last_per * row,
                                                 std::min(e->last_per, nb - g * e->last_per), stream);
            if (r != DYF_OK) return fail(e, r, "row group: " + c->err);
        }
        return DYF_OK;
    }
    const float* src = nullptr;
    if (what == DYF_STATE_X0_HAT) src = e->s_x0hat;
    else if (what == DYF_STATE_X_S) src = e->s_xs;
    else if (what == DYF_STATE_X_NEXT) src = e->plan.steps.back().i_next >= 0.0f ? e->s_next : e->s_x0hat;  // = x0_hat past T-1
    else return fail(e, DYF_ERR_INVALID_ARGUMENT, "what must be a dyf_sampler_state");
    HIP_TRY(e, hipMemcpyAsync(out_dev, src, (size_t)nb * e->C * e->cfg.height * e->cfg.width * sizeof(float),
                              hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DYF_OK;
}

dyf_status dyf_set_log_intermediates(dyf_engine* e, int32_t enable) {
    if (!e) return DYF_ERR_INVALID_ARGUMENT;
    e->log_on = enable != 0;  // the log of the last logged call stays readable after logging is switched off
    return DYF_OK;
}

dyf_status dyf_get_log(dyf_engine* e, int32_t step, int32_t what, float* out_dev, int32_t nb, void* stream) {
    if (!e || !out_dev) return DYF_ERR_INVALID_ARGUMENT;
    if (!e->s_log || e->log_has_cur.size() != e->plan.steps.size())
        return fail(e, DYF_ERR_STATE, "no sampling call has been logged (dyf_set_log_intermediates, then dyf_sample)");
    if (step < 0 || step >= (int)e->plan.steps.size() || what < 0 || what > 2 || nb != e->log_nb)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "step / what / batch size outside the logged call");
    if (what == 2 && !e->log_has_cur[step])
        return fail(e, DYF_ERR_STATE, "x_interpolated_s is not defined at this step (naive sampling, or a first step that is the last)");
    const size_t field = (size_t)nb * e->C * e->cfg.height * e->cfg.width;
    HIP_TRY(e, hipMemcpyAsync(out_dev, e->s_log + ((size_t)step * 3 + what) * field, field * sizeof(float), hipMemcpyDeviceToDevice,
                              (hipStream_t)stream));
    return DYF_OK;
}

dyf_status dyf_time_conv_layer(dyf_engine* e, int32_t which, int32_t layer, int32_t nb, int32_t iters, void* stream,
                               double* avg_ms, double* flops, double* algo_bytes) {
    if (!e || which < 0 || which > 1 || layer < 0 || layer > 11 || iters < 1 || !avg_ms)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "bad argument to dyf_time_conv_layer");
    Net& n = e->net[which];
    if (!n.loaded) return fail(e, DYF_ERR_STATE, "weights not loaded");
    if (n.rn || n.sc) return fail(e, DYF_ERR_UNSUPPORTED, "dyf_time_conv_layer addresses the 12 UNetBlocks of arch unet_simple");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const UBlock& b = n.blk[layer];
    ConvArgs a = block_conv_args(e, n, b, nb);
    // same operands as in net_forward: whatever the last forward left in the workspace (realistic activations)
    a.src0 = b.transposed ? e->ws.up : (layer == 0 ? e->ws.stem : e->ws.enc[layer - 1]);
    a.c0 = b.cin;
    if (layer == 0 && n.stem_fused && e->cfg.enable_mfma && e->fuse_stem && n.cfg.input_dropout == 0.0f) fused_enc0_args(e, n, a);
    if (b.transposed && layer > 6) {  // fused x2-upsample form when net_forward uses it
        ConvArgs f = a;
        const UBlock& skipb = n.blk[11 - layer];
        f.src0 = e->ws.dec[layer - 7]; f.c0 = b.cin - skipb.cout; f.src1 = e->ws.enc[11 - layer]; f.c1 = skipb.cout;
        f.h = b.in_h / 2; f.w = b.in_w / 2; f.up2x = 1; f.wpk_up = b.wpk_up; f.wpk_up_frag = b.wpk_up_frag; f.up_border = e->ws.up_border;
        f.up_cols = b.up_cols; f.up_cbase = b.up_cbase; f.up_cidx = b.up_cidx; f.up_ntiles = b.up_ntiles; f.up_npad = b.up_npad;
        f.up_nvalid0 = b.up_nvalid0; f.up_nvalid1 = b.up_nvalid1; f.up_wo_store = b.up_wo_store;
        f.up_mix[0] = b.up_mix[0]; f.up_mix[1] = b.up_mix[1]; f.up_mix[2] = b.up_mix[2];
        if (use_fused_up(e, b, f)) a = f;
    }
    const float* A = n.tables ? n.tables : e->ws.coef_a;                    // row 0 of the plan's tables, or the
    const float* Cc = n.tables ? n.tables + n.total_c : e->ws.coef_c;       // coefficients of the last forward
    a.coef_a = b.gn ? b.static_a : A + b.film_off;
    a.coef_c = b.gn ? b.static_c : Cc + b.film_off;
    a.coef_stride = 0;
    a.drop = DropSpec{};
    if (b.gn) { a.out_f32 = e->ws.enc5_raw; a.act = ACT_NONE; } else { a.out_el16 = b.transposed ? e->ws.dec[layer - 6] : e->ws.enc[layer]; }
    hipEvent_t ev0, ev1;
    HIP_TRY(e, hipEventCreate(&ev0));
    HIP_TRY(e, hipEventCreate(&ev1));
    for (int i = 0; i < 2; ++i) {
        dyf_status s = run_conv(e, a, st);
        if (s != DYF_OK) return s;
    }
    HIP_TRY(e, hipEventRecord(ev0, st));
    for (int i = 0; i < iters; ++i) {
        dyf_status s = run_conv(e, a, st);
        if (s != DYF_OK) return s;
    }
    HIP_TRY(e, hipEventRecord(ev1, st));
    HIP_TRY(e, hipEventSynchronize(ev1));
    float ms = 0.0f;
    HIP_TRY(e, hipEventElapsedTime(&ms, ev0, ev1));
    (void)hipEventDestroy(ev0);
    (void)hipEventDestroy(ev1);
    *avg_ms = (double)ms / iters;
    // output pixels that are computed: all of them, or (sparse-column form of the last block) the columns the readout reads
    const bool sparse = a.up2x && a.up_cols != nullptr;
    const double M = (double)nb * b.out_h * (sparse ? (double)(a.up_nvalid0 + a.up_nvalid1) : (double)b.out_w);
    if (flops) *flops = 2.0 * M * b.cout * b.cin * b.k * b.k;
    // algorithmic HBM bytes: read the input once, write the output once, read the weights once (bf16)
    if (algo_bytes) {  // the fused x2-upsample form reads its input at low resolution
        const double in_px = a.up2x ? (double)a.h * a.w : (double)b.in_h * b.in_w;
        *algo_bytes = 2.0 * ((double)nb * in_px * b.cin + M * b.cout + (double)b.cout * b.cin * b.k * b.k);
    }
    return DYF_OK;
}

dyf_status dyf_ensemble_metrics(dyf_engine* e, const float* preds_dev, const float* targets_dev, int32_t n_members,
                                int64_t n_points, double* out_host, void* stream) {
    if (!e || !preds_dev || !targets_dev || !out_host) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (n_members < 1 || n_members > 15360) return fail(e, DYF_ERR_INVALID_ARGUMENT, "n_members outside [1, 15360]");
    if (n_points < 1) return fail(e, DYF_ERR_INVALID_ARGUMENT, "n_points must be positive");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    if (!e->metric_sums) {
        dyf_status s = dev_alloc(e, &e->metric_sums, 4);
        if (s != DYF_OK) return s;
    }
    HIP_TRY(e, launch_ensemble_metrics(preds_dev, targets_dev, n_members, n_points, e->metric_sums, st));
    double h[3];
    HIP_TRY(e, hipMemcpyAsync(h, e->metric_sums, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    const double mse = h[0] / (double)n_points;
    out_host[0] = mse;
    out_host[1] = std::sqrt(h[1] / (double)n_points) / std::sqrt(mse);
    out_host[2] = h[2] / (double)n_points;
    return DYF_OK;
}

dyf_status dyf_apply_boundary_conditions(dyf_engine* e, const dyf_bc_args* bc, float* preds_dev, void* stream) {
    if (!e || !bc || !preds_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (bc->kind != DYF_BC_NAVIER_STOKES && bc->kind != DYF_BC_SPRING_MESH)
        return fail(e, DYF_ERR_UNSUPPORTED, "Boundary conditions are implemented for navier-stokes and spring-mesh");  // as the reference
    if (bc->n_fields < 1 || bc->rows < 1 || bc->channels < 1 || bc->height < 1 || bc->width < 1 || bc->n_meta < 1)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "boundary conditions: bad geometry");
    if (!bc->row_meta_dev || !bc->fixed_mask_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "row_meta / fixed_mask must be given");
    if (bc->kind == DYF_BC_NAVIER_STOKES && (!bc->in_velocity_dev || !bc->vertex_y_dev || !bc->time_factor_dev))
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "navier-stokes boundary conditions need in_velocity, vertex_y and time_factor");
    if (bc->kind == DYF_BC_SPRING_MESH && !bc->boundary_dev)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "spring-mesh boundary conditions need the boundary values");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    BcArgs a{};
    a.preds = preds_dev; a.kind = bc->kind; a.n_fields = bc->n_fields; a.rows = bc->rows; a.c = bc->channels; a.h = bc->height;
    a.w = bc->width; a.n_meta = bc->n_meta; a.row_meta = bc->row_meta_dev; a.time_factor = bc->time_factor_dev; a.times_per_meta = bc->times_per_meta;
    a.fixed_mask = bc->fixed_mask_dev; a.in_velocity = bc->in_velocity_dev; a.vertex_y = bc->vertex_y_dev; a.boundary = bc->boundary_dev;
    HIP_TRY(e, launch_boundary_conditions(a, (hipStream_t)stream));
    return DYF_OK;
}

dyf_status dyf_criterion(dyf_engine* e, const float* pred_dev, const float* target_dev, int64_t count, int32_t kind,
                         double* out_host, void* stream) {
    if (!e || !pred_dev || !target_dev || !out_host) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (count < 1 || kind < 0 || kind > 2) return fail(e, DYF_ERR_INVALID_ARGUMENT, "count must be positive, kind in {0 l1, 1 mse, 2 smooth-l1}");
    if (((uintptr_t)pred_dev | (uintptr_t)target_dev) & 15) return fail(e, DYF_ERR_INVALID_ARGUMENT, "tensors must be 16-byte aligned");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    if (!e->metric_sums) {
        dyf_status s = dev_alloc(e, &e->metric_sums, 4);
        if (s != DYF_OK) return s;
    }
    HIP_TRY(e, launch_criterion_sum(pred_dev, target_dev, count, kind, e->metric_sums, st));
    double h = 0.0;
    HIP_TRY(e, hipMemcpyAsync(&h, e->metric_sums, sizeof(h), hipMemcpyDeviceToHost, st));
    HIP_TRY(e, hipStreamSynchronize(st));
    *out_host = h / (double)count;
    return DYF_OK;
}

// eager rollout with events around the launches of the kernel class e->prof_layer names; average per nb-row launch equivalent
static dyf_status time_class_in_rollout(dyf_engine* e, int cls, int nb, hipStream_t st, double* avg_ms, int32_t* launches) {
    e->prof_layer = cls;
    e->prof_ev.clear();
    e->prof_rows.clear();
    dyf_status r = run_plan(e, nb, nullptr, nullptr, st);  // eager launch of the whole rollout, not the captured graph
    e->prof_layer = -1;
    hipError_t se = hipStreamSynchronize(st);
    double tot = 0.0;
    for (auto& pr : e->prof_ev) {
        float ms = 0.0f;
        if (se == hipSuccess && hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) tot += ms;
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    const size_t cnt = e->prof_ev.size();
    // a paired interpolator launch covers 2 nb rows: report time per nb-row launch equivalent
    double units = 0.0;
    for (int rws : e->prof_rows) units += (double)rws / (double)nb;
    e->prof_ev.clear();
    e->prof_rows.clear();
    if (r != DYF_OK) return r;
    if (se != hipSuccess) return fail(e, DYF_ERR_HIP, std::string("rollout: ") + hipGetErrorString(se));
    if (cnt == 0) return fail(e, DYF_ERR_STATE, "kernel class was not launched");
    *avg_ms = tot / units;
    if (launches) *launches = (int32_t)cnt;
    return DYF_OK;
}

dyf_status dyf_time_layer_in_rollout(dyf_engine* e, int32_t layer, int32_t nb, void* stream, double* avg_ms,
                                     int32_t* launches) {
    if (!e || layer < 6 || layer > 11 || !avg_ms) return fail(e, DYF_ERR_INVALID_ARGUMENT, "decoder layer 6..11 expected");
    if (!e->plan.set || !e->s_init) return fail(e, DYF_ERR_STATE, "needs a plan and one earlier dyf_sample call (its inputs are re-used)");
    if (e->net[0].rn || e->net[1].rn || e->net[0].sc || e->net[1].sc) return fail(e, DYF_ERR_UNSUPPORTED, "arch unet_simple only");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    return time_class_in_rollout(e, layer, nb, (hipStream_t)stream, avg_ms, launches);
}

dyf_status dyf_time_kernel_in_rollout(dyf_engine* e, int32_t kind, int32_t nb, void* stream, double* avg_ms, int32_t* launches,
                                      double* flops, double* algorithmic_bytes) {
    if (!e || kind < 0 || kind > 2 || !avg_ms) return fail(e, DYF_ERR_INVALID_ARGUMENT, "kind 0 (level-0 3x3 convs), 1 (attention core) or 2 (level-0 GroupNorm chain) expected");
    if (!e->plan.set || !e->s_init) return fail(e, DYF_ERR_STATE, "needs a plan and one earlier dyf_sample call (its inputs are re-used)");
    if (!e->net[0].rn || !e->net[1].rn) return fail(e, DYF_ERR_UNSUPPORTED, "arch unet (ResNet-UNet) pair only");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const double H = e->cfg.height, W = e->cfg.width, d = e->net[1].cfg.dim;
    int lv = e->net[1].cfg.n_mults - 1;
    const double tok = (H / (1 << lv)) * (W / (1 << lv));  // tokens of the bottleneck attention
    if (flops) *flops = kind == 0 ? 2.0 * nb * H * W * 9.0 * d * d : kind == 1 ? (double)nb * 4 * 2.0 * 2.0 * tok * tok * 32 : 0.0;
    if (algorithmic_bytes)  // 16-bit tensors, every operand once
        *algorithmic_bytes = kind == 0 ? 2.0 * (2.0 * nb * H * W * d + 9.0 * d * d)
                           : kind == 1 ? 2.0 * nb * tok * (384.0 + 128.0)
                                       : 2.0 * nb * H * W * d * 2.0;  // GroupNorm chain, fused ideal: read once, write once
    return time_class_in_rollout(e, DYF_PROF_RESNET_BASE + kind, nb, (hipStream_t)stream, avg_ms, launches);
}

dyf_status dyf_time_named_kernel_in_rollout(dyf_engine* e, const char* kernel, int32_t nb, void* stream, double* total_ms, int32_t* launches,
                                            double* total_bytes) {
    if (!e || !kernel || !total_ms || !launches || !total_bytes) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (!e->plan.set || !e->s_init) return fail(e, DYF_ERR_STATE, "needs a plan and one earlier dyf_sample call (its inputs are re-used)");
    if (nb < 1 || nb > e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, max_batch]");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    dyf_prof_arm(kernel);
    dyf_status r = run_plan(e, nb, nullptr, nullptr, st);  // eager launch of the whole rollout on the engine itself (no graph, no row groups)
    const hipError_t se = hipStreamSynchronize(st);
    int n = 0;
    dyf_prof_collect(total_ms, total_bytes, &n);
    *launches = n;
    if (r != DYF_OK) return r;
    if (se != hipSuccess) return fail(e, DYF_ERR_HIP, std::string("rollout: ") + hipGetErrorString(se));
    return DYF_OK;
}

void dyf_debug_set_form(const char* key, const char* value) { dyf_form_set(key, value); }
int32_t dyf_debug_forms(char* buf, int32_t cap) {
    const std::string t = dyf_form_text();
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(t.size(), (size_t)cap - 1);
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int32_t)t.size();
}

void dyf_debug_form_log(int32_t enable) { dyf_form_log_enable(enable != 0); }
int32_t dyf_debug_form_log_read(char* buf, int32_t cap) {
    const std::string t = dyf_form_log_text();
    if (buf && cap > 0) {
        const size_t n = std::min<size_t>(t.size(), (size_t)cap - 1);
        memcpy(buf, t.data(), n);
        buf[n] = 0;
    }
    return (int32_t)t.size();
}

dyf_status dyf_debug_read_block_output(dyf_engine* e, int32_t which, int32_t layer, int32_t nb, float* out_dev, void* stream) {
    if (!e || !out_dev || which < 0 || which > 1 || layer < 0 || layer > 11)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "bad argument to dyf_debug_read_block_output");
    const Net& n = e->net[which];
    if (n.rn || n.sc) return fail(e, DYF_ERR_UNSUPPORTED, "dyf_debug_read_block_output addresses the 12 UNetBlocks of arch unet_simple");
    if (nb < 1 || nb > 2 * e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "batch size outside [1, 2 max_batch]");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    const UBlock& b = n.blk[layer];
    const el16_t* src = layer < 6 ? e->ws.enc[layer] : e->ws.dec[layer - 6];
    const bool sparse = layer == 11 && e->last_dec5_sparse;
    HIP_TRY(e, launch_nhwc_to_nchw_f32(src, nb, b.out_h, b.out_w, sparse ? b.up_wo_store : b.out_w, b.cout,
                                       sparse ? b.up_col_map : nullptr, out_dev, (hipStream_t)stream));
    return DYF_OK;
}

dyf_status dyf_op_conv2d(dyf_engine* e, const uint16_t* x_dev, const float* w_host, int32_t n, int32_t h, int32_t w,
                         int32_t cin, int32_t cout, int32_t kh, int32_t kw, int32_t stride, int32_t pad,
                         const float* scale_dev, const float* shift_dev, int32_t act, int32_t path, uint16_t* y_dev,
                         void* stream) {
    if (!e || !x_dev || !w_host || !y_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    if (n < 1 || h < 1 || w < 1 || cin < 1 || cout < 1 || kh < 1 || kw < 1 || stride < 1 || pad < 0)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "bad conv geometry");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    const int taps = kh * kw;
    std::vector<el16_t> pk((size_t)cout * taps * cin);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t)
                pk[((size_t)co * taps + t) * cin + ci] = f32_to_el16(w_host[((size_t)co * cin + ci) * taps + t]);
    el16_t* wdev = nullptr;
    float *ones = nullptr, *zeros = nullptr;
    const bool frag = cout % 64 == 0 && cin % 64 == 0 && taps <= 32;
    HIP_TRY(e, hipMalloc((void**)&wdev, 2 * pk.size() * sizeof(el16_t)));
    HIP_TRY(e, hipMemcpy(wdev, pk.data(), pk.size() * sizeof(el16_t), hipMemcpyHostToDevice));
    if (frag) {
        std::vector<el16_t> pf(pk.size());
        pack_conv_frag(pk.data(), cout, taps, cin, pf.data());
        HIP_TRY(e, hipMemcpy(wdev + pk.size(), pf.data(), pf.size() * sizeof(el16_t), hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src0 = x_dev; a.c0 = cin; a.n = n; a.h = h; a.w = w;
    a.ho = (h + 2 * pad - kh) / stride + 1; a.wo = (w + 2 * pad - kw) / stride + 1;
    a.kh = kh; a.kw = kw; a.stride = stride; a.pad = pad; a.cout = cout; a.wpk = wdev;
    a.wpk_frag = frag ? wdev + pk.size() : nullptr;
    el16_t* h3dev = nullptr;  // halo form of plain 3x3 convs (looked up through the registry like the engine's own weights)
    if (((taps == 9 && cout % 64 == 0) || (kh == 4 && kw == 4 && cout % 128 == 0)) && cin % 64 == 0) {
        std::vector<el16_t> pf((size_t)cout * 16 * cin * (taps == 9 ? 1 : 4));
        if (taps == 9 && cout % 256 == 0) pack_halo3_frag(pk.data(), cout, cin, pf.data());
        else if (taps == 9) pack_halo3_frag64(pk.data(), cout, cin, pf.data());
        else pack_halo_s2_frag(pk.data(), cout, cin, pf.data());
        HIP_TRY(e, hipMalloc((void**)&h3dev, pf.size() * sizeof(el16_t)));
        HIP_TRY(e, hipMemcpy(h3dev, pf.data(), pf.size() * sizeof(el16_t), hipMemcpyHostToDevice));
        conv_register_halo3_frag(wdev, h3dev);
    }
    a.act = act; a.out_el16 = y_dev; a.zero_page = e->ws.zero_page;
    a.splitk_ws = e->ws.splitk; a.splitk_cap = e->ws.splitk ? DYF_SPLITK_FLOATS : 0;  // the split-K forms, as in the engine's own launches
    if (scale_dev && shift_dev) {
        a.coef_a = scale_dev; a.coef_c = shift_dev; a.coef_stride = cout;
    } else {
        std::vector<float> o1(cout, 1.0f), z0(cout, 0.0f);
        HIP_TRY(e, hipMalloc((void**)&ones, cout * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&zeros, cout * sizeof(float)));
        HIP_TRY(e, hipMemcpy(ones, o1.data(), cout * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(zeros, z0.data(), cout * sizeof(float), hipMemcpyHostToDevice));
        a.coef_a = ones; a.coef_c = zeros; a.coef_stride = 0;
    }
    dyf_status rs = DYF_OK;
    if (path == 1 && !conv_mfma_supported(a)) {
        rs = fail(e, DYF_ERR_UNSUPPORTED, "MFMA path needs cin % 64 == 0 and cout % 64 == 0");
    } else {
        hipError_t le = launch_conv(a, path, st);
        if (le == hipSuccess) le = hipStreamSynchronize(st);
        if (le != hipSuccess) rs = fail(e, DYF_ERR_HIP, std::string("conv launch: ") + hipGetErrorString(le));
    }
    if (h3dev) {
        conv_unregister_frag(wdev);
        (void)hipFree(h3dev);
    }
    (void)hipFree(wdev);
    if (ones) (void)hipFree(ones);
    if (zeros) (void)hipFree(zeros);
    return rs;
}

dyf_status dyf_op_upconv2d(dyf_engine* e, const uint16_t* x_dev, const float* w_host, int32_t n, int32_t h, int32_t w,
                           int32_t cin, int32_t cout, const float* scale_dev, const float* shift_dev, int32_t act,
                           uint16_t* y_dev, void* stream) {
    if (!e || !x_dev || !w_host || !y_dev) return fail(e, DYF_ERR_INVALID_ARGUMENT, "null argument");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<el16_t> pu((size_t)4 * cout * 16 * cin);
    pack_up2x_weights(w_host, cout, cin, pu.data());
    el16_t* wdev = nullptr;
    float *ones = nullptr, *zeros = nullptr;
    const bool frag = cin % 64 == 0 && cout % 64 == 0;
    HIP_TRY(e, hipMalloc((void**)&wdev, 2 * pu.size() * sizeof(el16_t)));
    HIP_TRY(e, hipMemcpy(wdev, pu.data(), pu.size() * sizeof(el16_t), hipMemcpyHostToDevice));
    if (frag) {
        std::vector<el16_t> pf(pu.size());
        pack_up2x_frag(pu.data(), cout, cin, pf.data());
        HIP_TRY(e, hipMemcpy(wdev + pu.size(), pf.data(), pf.size() * sizeof(el16_t), hipMemcpyHostToDevice));
    }
    ConvArgs a{};
    a.src0 = x_dev; a.c0 = cin; a.n = n; a.h = h; a.w = w; a.ho = 2 * h; a.wo = 2 * w;
    a.kh = 3; a.kw = 3; a.stride = 1; a.pad = 1; a.cout = cout; a.wpk = wdev; a.wpk_up = wdev; a.up2x = 1;
    a.wpk_up_frag = frag ? wdev + pu.size() : nullptr;
    a.act = act; a.out_el16 = y_dev;
    a.splitk_ws = e->ws.splitk; a.splitk_cap = e->ws.splitk ? DYF_SPLITK_FLOATS : 0;
    float* border = nullptr;
    HIP_TRY(e, hipMalloc((void**)&border, conv_up_border_floats(n, h, w, cout) * sizeof(float)));
    a.up_border = border;
    if (scale_dev && shift_dev) {
        a.coef_a = scale_dev; a.coef_c = shift_dev; a.coef_stride = cout;
    } else {
        std::vector<float> o1(cout, 1.0f), z0(cout, 0.0f);
        HIP_TRY(e, hipMalloc((void**)&ones, cout * sizeof(float)));
        HIP_TRY(e, hipMalloc((void**)&zeros, cout * sizeof(float)));
        HIP_TRY(e, hipMemcpy(ones, o1.data(), cout * sizeof(float), hipMemcpyHostToDevice));
        HIP_TRY(e, hipMemcpy(zeros, z0.data(), cout * sizeof(float), hipMemcpyHostToDevice));
        a.coef_a = ones; a.coef_c = zeros; a.coef_stride = 0;
    }
    dyf_status rs = DYF_OK;
    if (!conv_mfma_supported(a)) {
        rs = fail(e, DYF_ERR_UNSUPPORTED, "fused upsample conv needs cin % 64 == 0, cout % 64 == 0, w % 16 == 0, h % 8/16 == 0");
    } else {
        hipError_t le = launch_conv(a, 1, st);
        if (le == hipSuccess) le = hipStreamSynchronize(st);
        if (le != hipSuccess) rs = fail(e, DYF_ERR_HIP, std::string("upconv launch: ") + hipGetErrorString(le));
    }
    (void)hipFree(wdev);
    (void)hipFree(border);
    if (ones) (void)hipFree(ones);
    if (zeros) (void)hipFree(zeros);
    return rs;
}

dyf_status dyf_op_linear_attention(dyf_engine* e, const uint16_t* qkv_dev, int32_t n, int32_t hw, uint16_t* out_dev,
                                   void* stream) {
    if (!e || !qkv_dev || !out_dev || n < 1 || hw < 1) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_op_linear_attention: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    const int heads = 4;
    const size_t nblk = ((size_t)hw + 1023) / 1024;
    float* scratch = nullptr;
    HIP_TRY(e, hipMalloc((void**)&scratch, (size_t)n * heads * (nblk * 1088 + 1024) * sizeof(float)));
    LinAttnArgs l{};
    l.qkv = qkv_dev; l.n = n; l.hw = hw; l.heads = heads; l.out = out_dev; l.scratch = scratch;
    hipError_t err = launch_linear_attention(l, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    (void)hipFree(scratch);
    if (err != hipSuccess) return fail(e, DYF_ERR_HIP, std::string("dyf_op_linear_attention: ") + hipGetErrorString(err));
    return DYF_OK;
}

dyf_status dyf_op_linear_attention_fused(dyf_engine* e, const uint16_t* xn_dev, const uint16_t* xres_dev, int32_t n, int32_t hw,
                                         int32_t c, const float* wqkv_host, const float* wout_host, const float* bout_host,
                                         uint16_t* y_dev, void* stream) {
    if (!e || !xn_dev || !xres_dev || !wqkv_host || !wout_host || !bout_host || !y_dev || n < 1 || hw < 1)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_op_linear_attention_fused: bad arguments");
    if (c != 64 && c != 128) return fail(e, DYF_ERR_UNSUPPORTED, "dyf_op_linear_attention_fused: dim 64 or 128");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    std::vector<el16_t> fq((size_t)384 * c), fo((size_t)c * 128);
    linattn_fused_pack(wqkv_host, wout_host, c, fq.data(), fo.data());
    const size_t nblk = ((size_t)hw + 255) / 256;  // (the smallest workgroups of the fused form: 8 groups of 32 pixels)
    const size_t scratch_floats = (size_t)n * 4 * (nblk * 1088 + 1024);
    float *scratch = nullptr, *bo = nullptr;
    el16_t *dq = nullptr, *dout = nullptr;
    hipError_t err = hipMalloc((void**)&scratch, scratch_floats * sizeof(float));
    if (err == hipSuccess) err = hipMalloc((void**)&bo, (size_t)c * sizeof(float));
    if (err == hipSuccess) err = hipMalloc((void**)&dq, fq.size() * sizeof(el16_t));
    if (err == hipSuccess) err = hipMalloc((void**)&dout, fo.size() * sizeof(el16_t));
    if (err == hipSuccess) err = hipMemcpy(bo, bout_host, (size_t)c * sizeof(float), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(dq, fq.data(), fq.size() * sizeof(el16_t), hipMemcpyHostToDevice);
    if (err == hipSuccess) err = hipMemcpy(dout, fo.data(), fo.size() * sizeof(el16_t), hipMemcpyHostToDevice);
    if (err == hipSuccess) {
        LinAttnFusedArgs f{};
        f.xn = xn_dev; f.xres = xres_dev; f.n = n; f.hw = hw; f.c = c; f.wqkv_frag = dq; f.wout_frag = dout; f.bout = bo;
        f.y = y_dev; f.scratch = scratch; f.scratch_floats = (long long)scratch_floats;
        err = launch_linear_attention_fused(f, st);
    }
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    (void)hipFree(scratch); (void)hipFree(bo); (void)hipFree(dq); (void)hipFree(dout);
    if (err != hipSuccess) return fail(e, DYF_ERR_HIP, std::string("dyf_op_linear_attention_fused: ") + hipGetErrorString(err));
    return DYF_OK;
}

dyf_status dyf_op_attention(dyf_engine* e, const uint16_t* qkv_dev, int32_t n, int32_t hw, uint16_t* out_dev, void* stream) {
    return dyf_op_attention_dropout(e, qkv_dev, n, hw, 0.0f, out_dev, stream);
}

dyf_status dyf_op_attention_dropout(dyf_engine* e, const uint16_t* qkv_dev, int32_t n, int32_t hw, float p, uint16_t* out_dev, void* stream) {
    if (!e || !qkv_dev || !out_dev || n < 1 || hw < 1 || p < 0.0f || p >= 1.0f) return fail(e, DYF_ERR_INVALID_ARGUMENT, "dyf_op_attention: bad arguments");
    HIP_TRY(e, hipSetDevice(e->cfg.device));
    hipStream_t st = (hipStream_t)stream;
    AttnArgs a{};
    a.qkv = qkv_dev; a.n = n; a.hw = hw; a.heads = 4; a.out = out_dev; a.drop = DropSpec{};
    if (p > 0.0f) {  // dropout on the probabilities from the engine's generator (forward counter advances, as in a network forward)
        if (n > 2 * e->cfg.max_batch) return fail(e, DYF_ERR_INVALID_ARGUMENT, "more rows than the engine's row-key table (2 max_batch)");
        HIP_TRY(e, launch_rng_begin_forward(e->rng_state, e->row_keys, n, n, st));
        a.drop.mode = 1;
        a.drop.scale = 1.0f / (1.0f - p);
        a.drop.thresh16 = keep_threshold16(p);
        a.drop.thresh8 = keep_threshold8(p);
        a.drop.scale8 = 256.0f / (float)a.drop.thresh8;
        a.drop.salt = rng_layer_salt(0u);
        a.drop.row_keys = e->row_keys;
    }
    hipError_t err = launch_attention(a, st);
    if (err == hipSuccess) err = hipStreamSynchronize(st);
    if (err != hipSuccess) return fail(e, DYF_ERR_HIP, std::string("dyf_op_attention: ") + hipGetErrorString(err));
    return DYF_OK;
}

}  // extern "C"
