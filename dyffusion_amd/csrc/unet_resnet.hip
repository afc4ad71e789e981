// ResNet-UNet backbone (src/models/unet.py:112-315) on the HIP engine: weight preparation and forward orchestration.
//
//   ResnetBlock  (unet.py:79-109)  = Block(FiLM) -> Block -> + residual_conv(x)
//   Block        (unet.py:58-76)   = WeightStandardizedConv2d 3x3 -> GroupNorm(8) -> FiLM -> SiLU -> Dropout
//   LinearAttention / Attention (modules/attention.py:7-73) under Residual(PreNorm(LayerNorm)) (net_norm.py:18-26)
//
// Mapping to kernels: every conv (3x3 WS-conv, 1x1 qkv/out/residual, 4x4 s2 down, 3x3 up) is the MFMA implicit GEMM of
// conv.hip (weight standardisation is constant at inference and folded at load time); GroupNorm+FiLM+SiLU+
// Dropout(+residual) is one wavefront-reduction kernel per Block (gn_act_kernel); LayerNorm, LinearAttention, Attention,
// the 7x7 stem and the 1x1 head are in unet_kernels.hip.  Skip concatenations are never materialised (two-source convs).
#include "engine_internal.h"
#include "unet_kernels.h"

namespace dyf {

struct RBlockW {  // ResnetBlock
    int cin = 0, cout = 0;
    bool has_res = false;
    bool single = false;  // double_conv_layer=False: no second Block (unet.py:94)
    int film_off = 0;
    el16_t *w1 = nullptr, *w2 = nullptr, *wr = nullptr;
    float *b1 = nullptr, *b2 = nullptr, *br = nullptr;
    float *g1 = nullptr, *be1 = nullptr, *g2 = nullptr, *be2 = nullptr;
};

struct AttnW {
    int dim = 0;
    bool linear = true;
    float* ln_g = nullptr;
    el16_t *wqkv = nullptr, *wout = nullptr;
    el16_t *wqkv_frag = nullptr, *wout_frag = nullptr;  // fused LinearAttention (dim 64 / 128)
    float* bout = nullptr;
};

struct SampW {  // down / up sampling conv
    int cin = 0, cout = 0, k = 3, stride = 1, pad = 1;
    bool nearest_up = false;
    el16_t* w = nullptr;
    float* b = nullptr;
};

struct RNet {
    int nlev = 0;
    int dims[8] = {};          // dims[0] = init_dim, dims[i+1] = dim * mult[i]
    int lev_h[8] = {}, lev_w[8] = {};
    std::vector<RBlockW> blocks;   // downs.l.{0,1} (2*nlev), mid_block1, mid_block2, ups.l.{0,1} (2*nlev), final_res_block
    std::vector<AttnW> attns;      // downs.l.2 (nlev), mid_attn, ups.l.2 (nlev)
    std::vector<SampW> downs, ups;
    float *stem_w = nullptr, *stem_b = nullptr, *head_w = nullptr, *head_b = nullptr;
    el16_t* stem_wfrag = nullptr;  // MFMA stem fragments (dim 64)
    int stem_ksteps = 0;
    float *ones = nullptr, *zeros = nullptr;
    double* gn_stats = nullptr;    // [max_batch][groups][2]
    float* gn_part = nullptr;      // GroupNorm partial sums written by conv epilogues (ConvArgs::gn_part), gn_part_floats
    size_t gn_part_floats = 0;
    // GroupNorm fused into the producing conv (gn_fused.h): granule buffer [max_batch][gn_max_slots][maxc / 8][2] x 8 B (zeroed
    // once: tags never repeat), the forward-epoch word the tags are built from, 0 slots = no level of this grid is served
    unsigned long long* gn_gran = nullptr;
    uint32_t* gn_epoch = nullptr;
    int gn_max_slots = 0;
    float* la_scratch = nullptr;   // LinearAttention partials + context
    size_t la_scratch_floats = 0;
    size_t buf_elems = 0;          // elements of one pool buffer at max_batch
    std::vector<el16_t*> pool;
};

namespace {

constexpr int HEADS = 4, DIM_HEAD = 32, HID = HEADS * DIM_HEAD;

struct Pool {
    std::vector<el16_t*> free_list;
    el16_t* get() {
        el16_t* p = free_list.back();
        free_list.pop_back();
        return p;
    }
    void put(el16_t* p) { free_list.push_back(p); }
};

struct DropCtx {  // walks the dropout sites in execution order (same order as the reference / oracle)
    const dyf_engine* e;
    const FwdOpts* o;
    int site = 0, mask_idx = 0;
    DropSpec next(float p) {
        DropSpec d{};
        if (p <= 0.0f) return d;  // p = 0 layers draw nothing and consume no mask
        d.mode = o->dropout_mode;
        d.scale = 1.0f / (1.0f - p);
        d.thresh16 = keep_threshold16(p);
        d.thresh8 = keep_threshold8(p);
        d.scale8 = 256.0f / (float)d.thresh8;
        d.salt = rng_layer_salt((uint32_t)site++);
        d.row_keys = e->row_keys;
        if (d.mode == 2) {
            d.mask = o->masks ? o->masks[mask_idx++] : nullptr;
            if (!d.mask) d.mode = 0;
        }
        return d;
    }
};

dyf_status rconv(dyf_engine* e, const el16_t* s0, int c0, const el16_t* s1, int c1, int n, int h, int w, int k, int stride,
                 int pad, int cout, const el16_t* wpk, const float* coef_a, const float* coef_c, int coef_stride, int act,
                 const DropSpec& drop, const el16_t* residual, el16_t* out, hipStream_t st, float* gn_part = nullptr,
                 int* gn_slots = nullptr, int up_nearest = 0) {
    ConvArgs a{};
    a.src0 = s0; a.c0 = c0; a.src1 = s1; a.c1 = c1; a.n = n; a.h = h; a.w = w;
    a.up_nearest = up_nearest;  // s0 is the (h / 2) x (w / 2) tensor (only after rconv_nearest_fusable said yes)
    a.ho = (h + 2 * pad - k) / stride + 1; a.wo = (w + 2 * pad - k) / stride + 1;
    a.kh = k; a.kw = k; a.stride = stride; a.pad = pad; a.cout = cout; a.wpk = wpk;
    a.coef_a = coef_a; a.coef_c = coef_c; a.coef_stride = coef_stride; a.act = act; a.drop = drop;
    a.residual = residual; a.out_el16 = out;
    a.splitk_ws = e->ws.splitk; a.splitk_cap = DYF_SPLITK_FLOATS;
    a.n_sel = e->cfg.batch_invariant ? 2 * e->cfg.max_batch : 0;
    // a row group shares the chip with the launches of its sibling groups: choose the kernel form by the tile count of all of them
    if (e->form_rows_scale > 1 && !e->cfg.batch_invariant) a.n_sel = n * e->form_rows_scale;
    const int path = (e->cfg.enable_mfma && conv_mfma_supported(a)) ? 1 : 0;
    ProfScope prof(e, e->prof_layer == DYF_PROF_RESNET_BASE + DYF_PROF_RN_CONV3_L0 && k == 3 && stride == 1 && h == e->cfg.height &&
                          w == e->cfg.width && c0 + c1 == cout && residual == nullptr, n, st);
    if (gn_part && gn_slots) {
        a.gn_part = gn_part;
        HIP_TRY(e, launch_conv_stats(a, path, st, gn_slots));
    } else {
        HIP_TRY(e, launch_conv(a, path, st));
    }
    return DYF_OK;
}

// the plain 3x3 conv behind a nearest x2 upsample (output plane h x w) will run on the one form that folds the upsample into its gather
bool rconv_nearest_fusable(dyf_engine* e, int c0, int n, int h, int w, int cout, const el16_t* wpk) {
    if (dyf_form("DYF_FUSE_NEAREST") && atoi(dyf_form("DYF_FUSE_NEAREST")) == 0) return false;  // A/B + parity test
    if (!e->cfg.enable_mfma || (h & 1) || (w & 1)) return false;
    ConvArgs a{};
    a.src0 = (const el16_t*)wpk;  // (any non-null pointer: the predicates look at shapes)
    a.c0 = c0; a.n = n; a.h = h; a.w = w; a.ho = h; a.wo = w; a.kh = 3; a.kw = 3; a.stride = 1; a.pad = 1; a.cout = cout; a.wpk = wpk;
    a.out_el16 = (el16_t*)wpk;
    a.n_sel = e->cfg.batch_invariant ? 2 * e->cfg.max_batch : 0;
    if (e->form_rows_scale > 1 && !e->cfg.batch_invariant) a.n_sel = n * e->form_rows_scale;
    return conv_plain3x3_takes_halo5(a);
}

std::vector<el16_t> pack_conv(const float* w, int cout, int cin, int k) {
    const int taps = k * k;
    std::vector<el16_t> pk((size_t)cout * taps * cin);
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int t = 0; t < taps; ++t) pk[((size_t)co * taps + t) * cin + ci] = f32_to_el16(w[((size_t)co * cin + ci) * taps + t]);
    return pk;
}

// WeightStandardizedConv2d (unet.py:26-40): (w - mean) * rsqrt(var + 1e-5) per output channel, biased variance
std::vector<float> standardize(const float* w, int cout, int per_out) {
    std::vector<float> r((size_t)cout * per_out);
    for (int co = 0; co < cout; ++co) {
        const float* p = w + (size_t)co * per_out;
        double m = 0.0;
        for (int i = 0; i < per_out; ++i) m += p[i];
        m /= per_out;
        double v = 0.0;
        for (int i = 0; i < per_out; ++i) v += (p[i] - m) * (p[i] - m);
        v /= per_out;
        const double rs = 1.0 / std::sqrt(v + 1e-5);
        for (int i = 0; i < per_out; ++i) r[(size_t)co * per_out + i] = (float)((p[i] - m) * rs);
    }
    return r;
}

}  // namespace

// ------------------------------------------------------------------------------------------------ configuration
std::string rn_configure(dyf_engine* e, Net& n) {
    const dyf_net_config& c = n.cfg;
    if (c.n_mults < 1 || c.n_mults > 6) return "dim_mults must have 1..6 entries";
    if (c.upsample_h != 0 || c.upsample_w != 0) return "unet.Unet with an outer resampler is not implemented";
    if (c.groups < 1) return "resnet_block_groups must be positive";
    if (c.learned_sinusoidal_dim < 0 || c.learned_sinusoidal_dim % 2 != 0 || c.learned_sinusoidal_dim > 256)
        return "learned_sinusoidal_dim must be even and <= 256";
    const bool keep = c.keep_spatial_dims != 0;
    if (c.init_kernel_size < 1 || c.init_kernel_size > 9 || c.init_padding * 2 + 1 != c.init_kernel_size)
        return "init_conv must be a 'same' convolution with an odd kernel <= 9";
    RNet* r = new RNet();
    n.rn = r;
    r->nlev = c.n_mults;
    r->dims[0] = c.dim;
    for (int i = 0; i < c.n_mults; ++i) {
        r->dims[i + 1] = c.dim * c.dim_mults[i];
        if (r->dims[i + 1] % c.groups != 0) return "every level width must be divisible by resnet_block_groups";
    }
    if (c.dim % c.groups != 0) return "dim must be divisible by resnet_block_groups";
    int h = e->cfg.height, w = e->cfg.width;
    for (int l = 0; l < r->nlev; ++l) {
        r->lev_h[l] = h;
        r->lev_w[l] = w;
        if (l < r->nlev - 1 && !keep) {
            if (h % 2 || w % 2) return "grid must be divisible by 2^(levels-1) (unet.py down/up sampling)";
            h /= 2;
            w /= 2;
        }
    }
    // ---- layer table
    int off = 0;
    auto add_block = [&](int cin, int cout) {
        RBlockW b;
        b.cin = cin; b.cout = cout; b.has_res = cin != cout; b.film_off = off; b.single = c.single_conv_layer != 0;
        off += cout;
        r->blocks.push_back(b);
    };
    for (int l = 0; l < r->nlev; ++l) { add_block(r->dims[l], r->dims[l]); add_block(r->dims[l], r->dims[l]); }
    const int mid = r->dims[r->nlev];
    add_block(mid, mid);
    add_block(mid, mid);
    for (int l = r->nlev - 1; l >= 0; --l) { add_block(r->dims[l + 1] + r->dims[l], r->dims[l + 1]); add_block(r->dims[l + 1] + r->dims[l], r->dims[l + 1]); }
    add_block(2 * c.dim, c.dim);
    n.total_c = off;
    n.dim = c.dim;
    n.tdim = 2 * c.dim;
    n.cin_total = c.in_channels + c.cond_channels;
    for (int l = 0; l < r->nlev; ++l) { AttnW a; a.dim = r->dims[l]; a.linear = true; r->attns.push_back(a); }
    { AttnW a; a.dim = mid; a.linear = false; r->attns.push_back(a); }
    for (int l = r->nlev - 1; l >= 0; --l) { AttnW a; a.dim = r->dims[l + 1]; a.linear = true; r->attns.push_back(a); }
    for (int l = 0; l < r->nlev; ++l) {
        SampW s;
        s.cin = r->dims[l]; s.cout = r->dims[l + 1];
        if (l < r->nlev - 1 && !keep) { s.k = 4; s.stride = 2; s.pad = 1; } else { s.k = 3; s.stride = 1; s.pad = 1; }
        r->downs.push_back(s);
    }
    for (int l = r->nlev - 1; l >= 0; --l) {
        SampW s;
        s.cin = r->dims[l + 1]; s.cout = r->dims[l]; s.k = 3; s.stride = 1; s.pad = 1; s.nearest_up = l > 0 && !keep;
        r->ups.push_back(s);
    }
    // ---- 2*MAC of conv / matmul layers per sample (torch flop counter convention)
    double f = 0.0;
    const int ks = c.init_kernel_size;
    f += 2.0 * e->cfg.height * e->cfg.width * (double)n.cin_total * c.dim * ks * ks;
    auto blk_f = [&](const RBlockW& b, int hh, int ww) {
        double px = (double)hh * ww;
        return 2.0 * px * b.cout * (9.0 * b.cin + (b.single ? 0.0 : 9.0 * b.cout) + (b.has_res ? b.cin : 0));
    };
    auto attn_f = [&](const AttnW& a, int hh, int ww) {
        double px = (double)hh * ww;
        double v = 2.0 * px * a.dim * 3 * HID + 2.0 * px * HID * a.dim;
        v += a.linear ? 2.0 * HEADS * (2.0 * px * DIM_HEAD * DIM_HEAD) : 2.0 * HEADS * (2.0 * px * px * DIM_HEAD);
        return v;
    };
    int bi = 0, ai = 0;
    for (int l = 0; l < r->nlev; ++l) {
        f += blk_f(r->blocks[bi++], r->lev_h[l], r->lev_w[l]) + blk_f(r->blocks[bi++], r->lev_h[l], r->lev_w[l]);
        f += attn_f(r->attns[ai++], r->lev_h[l], r->lev_w[l]);
        const SampW& s = r->downs[l];
        const int oh = s.stride == 2 ? r->lev_h[l] / 2 : r->lev_h[l], ow = s.stride == 2 ? r->lev_w[l] / 2 : r->lev_w[l];
        f += 2.0 * oh * ow * (double)s.cout * s.cin * s.k * s.k;
    }
    const int mh = r->lev_h[r->nlev - 1], mw = r->lev_w[r->nlev - 1];
    f += blk_f(r->blocks[bi++], mh, mw) + attn_f(r->attns[ai++], mh, mw) + blk_f(r->blocks[bi++], mh, mw);
    for (int l = r->nlev - 1, u = 0; l >= 0; --l, ++u) {
        f += blk_f(r->blocks[bi++], r->lev_h[l], r->lev_w[l]) + blk_f(r->blocks[bi++], r->lev_h[l], r->lev_w[l]);
        f += attn_f(r->attns[ai++], r->lev_h[l], r->lev_w[l]);
        const SampW& s = r->ups[u];
        const int oh = s.nearest_up ? 2 * r->lev_h[l] : r->lev_h[l], ow = s.nearest_up ? 2 * r->lev_w[l] : r->lev_w[l];
        f += 2.0 * oh * ow * (double)s.cout * s.cin * 9;
    }
    f += blk_f(r->blocks[bi++], e->cfg.height, e->cfg.width);
    f += 2.0 * e->cfg.height * e->cfg.width * (double)c.dim * c.out_channels;
    n.flops_per_sample = f;
    // ---- pool buffer size: the widest tensor is the qkv projection (3*128 channels) or a level's widest activation
    size_t per_sample = 0;
    for (int l = 0; l < r->nlev; ++l) {
        const size_t px = (size_t)r->lev_h[l] * r->lev_w[l];
        const size_t cmax = std::max<size_t>(3 * HID, (size_t)r->dims[l + 1] + r->dims[l]);
        per_sample = std::max(per_sample, px * cmax);
        if (l > 0) per_sample = std::max(per_sample, 4 * px * (size_t)r->dims[l + 1]);  // nearest x2 of the level's output
    }
    r->buf_elems = per_sample * (size_t)e->cfg.max_batch;
    n.n_drop_sites = 0;
    for (size_t i = 0; i < r->blocks.size(); ++i) n.n_drop_sites += (c.block_dropout1 > 0) + (c.dropout > 0 && !c.single_conv_layer);
    if (c.attn_dropout > 0) n.n_drop_sites += (int)r->attns.size();
    if (c.input_dropout > 0) n.n_drop_sites += 2;  // dropout_input_for_residual, dropout_input (unet.py:276-277)
    return "";
}

dyf_status rn_alloc_workspace(dyf_engine* e) {
    for (int w = 0; w < 2; ++w) {
        Net& n = e->net[w];
        if (!n.rn) continue;
        RNet* r = n.rn;
        {
            dyf_status s = dev_alloc(e, &r->gn_stats, gn_stats_doubles((size_t)e->cfg.max_batch, (size_t)n.cfg.groups));
            if (s != DYF_OK) return s;
        }
        {
            int maxc = 0;
            for (auto& b : r->blocks) maxc = std::max(maxc, b.cout);
            r->gn_part_floats = (size_t)e->cfg.max_batch * conv_halo5_gn_slots(e->cfg.height, e->cfg.width) * (maxc / 8 + 1) * 2;
            dyf_status s = dev_alloc(e, &r->gn_part, r->gn_part_floats);
            if (s != DYF_OK) return s;
        }
        {
            int maxc = 0;
            for (auto& b : r->blocks) maxc = std::max(maxc, b.cout);
            for (int l = 0; l < r->nlev; ++l) r->gn_max_slots = std::max(r->gn_max_slots, conv_gn_fused_max_slots(r->lev_h[l], r->lev_w[l]));
            if (r->gn_max_slots > 0) {
                dyf_status s = dev_alloc(e, &r->gn_gran, (size_t)e->cfg.max_batch * r->gn_max_slots * (maxc / 8) * 2);
                if (s != DYF_OK) return s;
                s = dev_alloc(e, &r->gn_epoch, 64);
                if (s != DYF_OK) return s;
            }
        }
        {
            // partials of the smallest workgroups of the fused form (8 groups = 256 pixels each) + ctx fragments
            const size_t nblk = ((size_t)e->cfg.height * e->cfg.width + 255) / 256;
            r->la_scratch_floats = (size_t)e->cfg.max_batch * HEADS * (nblk * 1088 + 1024);
            dyf_status s = dev_alloc(e, &r->la_scratch, r->la_scratch_floats);
            if (s != DYF_OK) return s;
        }
        const int nbuf = 2 * r->nlev + 8;
        // the two networks run back to back: share one pool when both are ResNet-UNets
        if (w == 1 && e->net[0].rn && e->net[0].rn->buf_elems >= r->buf_elems &&
            (int)e->net[0].rn->pool.size() >= nbuf) {
            r->pool = e->net[0].rn->pool;
            continue;
        }
        for (int i = 0; i < nbuf; ++i) {
            el16_t* p = nullptr;
            dyf_status s = dev_alloc(e, &p, r->buf_elems);
            if (s != DYF_OK) return s;
            r->pool.push_back(p);
        }
    }
    return DYF_OK;
}

void rn_destroy(Net& n) {
    delete n.rn;
    n.rn = nullptr;
}

// ------------------------------------------------------------------------------------------------ weights (K11)
dyf_status rn_load_weights(dyf_engine* e, Net& n, std::map<std::string, TensorView>& sd) {
    RNet* r = n.rn;
    const dyf_net_config& c = n.cfg;
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
#define NEED(var, key, ...)                                              \
    const TensorView* var = get(key, std::vector<int64_t>{__VA_ARGS__}); \
    if (!var) return fail(e, DYF_ERR_INVALID_ARGUMENT, missing)
#define UP(dst, hostvec)                                  \
    do {                                                  \
        dyf_status _s = dev_upload(e, &(dst), (hostvec)); \
        if (_s != DYF_OK) return _s;                      \
    } while (0)
    const int64_t d = c.dim, td = n.tdim;
    if (c.with_time_emb) {
        const int64_t tfeat = c.learned_sinusoidal_dim > 0 ? c.learned_sinusoidal_dim + 1 : d;
        NEED(w1, "time_emb_mlp.1.weight", td, tfeat);
        NEED(b1, "time_emb_mlp.1.bias", td);
        if (c.learned_sinusoidal_dim > 0) {
            NEED(lw, "time_emb_mlp.0.weights", (int64_t)c.learned_sinusoidal_dim / 2);
            UP(n.t_learned, vec(lw));
        }
        NEED(w2, "time_emb_mlp.3.weight", td, td);
        NEED(b2, "time_emb_mlp.3.bias", td);
        UP(n.t_w1, vec(w1)); UP(n.t_b1, vec(b1)); UP(n.t_w2, vec(w2)); UP(n.t_b2, vec(b2));
    }
    {
        const int64_t ks = c.init_kernel_size;
        NEED(sw, "init_conv.weight", d, (int64_t)n.cin_total, ks, ks);
        NEED(sb, "init_conv.bias", d);
        std::vector<float> pk((size_t)ks * ks * n.cin_total * d);  // [tap][cin][dim]
        for (int co = 0; co < d; ++co)
            for (int ci = 0; ci < n.cin_total; ++ci)
                for (int t = 0; t < ks * ks; ++t)
                    pk[((size_t)t * n.cin_total + ci) * d + co] = sw->data[((size_t)co * n.cin_total + ci) * ks * ks + t];
        UP(r->stem_w, pk); UP(r->stem_b, vec(sb));
        r->stem_ksteps = d == 64 ? stem_frag_steps((int)ks, n.cin_total) : 0;
        if (r->stem_ksteps > 0) {  // MFMA stem: weights as 16-bit hi/lo A fragments, channel-major K
            std::vector<el16_t> pf((size_t)r->stem_ksteps * 2 * 2 * 64 * 8);
            pack_stem_frag(pk.data(), (int)ks, n.cin_total, (int)d, pf.data());
            UP(r->stem_wfrag, pf);
        }
        NEED(hw, "final_conv.weight", (int64_t)c.out_channels, d, 1, 1);
        NEED(hb, "final_conv.bias", (int64_t)c.out_channels);
        UP(r->head_w, vec(hw)); UP(r->head_b, vec(hb));
    }
    int maxc = 3 * HID;
    for (auto& b : r->blocks) maxc = std::max(maxc, b.cout);
    UP(r->ones, std::vector<float>(maxc, 1.0f));
    UP(r->zeros, std::vector<float>(maxc, 0.0f));
    // block name table in the order rn_configure built them
    std::vector<std::string> bnames, anames;
    for (int l = 0; l < r->nlev; ++l) { bnames.push_back("downs." + std::to_string(l) + ".0"); bnames.push_back("downs." + std::to_string(l) + ".1"); }
    bnames.push_back("mid_block1");
    bnames.push_back("mid_block2");
    for (int u = 0; u < r->nlev; ++u) { bnames.push_back("ups." + std::to_string(u) + ".0"); bnames.push_back("ups." + std::to_string(u) + ".1"); }
    bnames.push_back("final_res_block");
    for (int l = 0; l < r->nlev; ++l) anames.push_back("downs." + std::to_string(l) + ".2");
    anames.push_back("mid_attn");
    for (int u = 0; u < r->nlev; ++u) anames.push_back("ups." + std::to_string(u) + ".2");

    std::vector<float> film_w((size_t)2 * n.total_c * n.tdim, 0.0f), film_b((size_t)2 * n.total_c, 0.0f);
    std::vector<int> blk_of(n.total_c), blk_off(r->blocks.size()), blk_cout(r->blocks.size());
    for (size_t i = 0; i < r->blocks.size(); ++i) {
        RBlockW& b = r->blocks[i];
        const std::string& P = bnames[i];
        NEED(w1, P + ".block1.proj.weight", (int64_t)b.cout, (int64_t)b.cin, 3, 3);
        NEED(b1, P + ".block1.proj.bias", (int64_t)b.cout);
        NEED(g1, P + ".block1.norm.weight", (int64_t)b.cout);
        NEED(e1, P + ".block1.norm.bias", (int64_t)b.cout);
        const TensorView *w2 = nullptr, *b2 = nullptr, *g2 = nullptr, *e2 = nullptr;
        if (!b.single) {
            w2 = get(P + ".block2.proj.weight", std::vector<int64_t>{(int64_t)b.cout, (int64_t)b.cout, 3, 3});
            b2 = get(P + ".block2.proj.bias", std::vector<int64_t>{(int64_t)b.cout});
            g2 = get(P + ".block2.norm.weight", std::vector<int64_t>{(int64_t)b.cout});
            e2 = get(P + ".block2.norm.bias", std::vector<int64_t>{(int64_t)b.cout});
            if (!w2 || !b2 || !g2 || !e2) return fail(e, DYF_ERR_INVALID_ARGUMENT, missing);
        }
#define UPW(dst, hostvec, CO, TAPS, CI)                                                         \
    do {                                                                                       \
        dyf_status _s = upload_conv_weights(e, &(dst), (hostvec), (CO), (TAPS), (CI));         \
        if (_s != DYF_OK) return _s;                                                           \
    } while (0)
        UPW(b.w1, pack_conv(standardize(w1->data, b.cout, b.cin * 9).data(), b.cout, b.cin, 3), b.cout, 9, b.cin);
        UP(b.b1, vec(b1)); UP(b.g1, vec(g1)); UP(b.be1, vec(e1));
        if (!b.single) {
            UPW(b.w2, pack_conv(standardize(w2->data, b.cout, b.cout * 9).data(), b.cout, b.cout, 3), b.cout, 9, b.cout);
            UP(b.b2, vec(b2)); UP(b.g2, vec(g2)); UP(b.be2, vec(e2));
        }
        if (b.has_res) {
            NEED(wr, P + ".residual_conv.weight", (int64_t)b.cout, (int64_t)b.cin, 1, 1);
            NEED(br, P + ".residual_conv.bias", (int64_t)b.cout);
            UPW(b.wr, pack_conv(wr->data, b.cout, b.cin, 1), b.cout, 1, b.cin);
            UP(b.br, vec(br));
        }
        blk_off[i] = b.film_off;
        blk_cout[i] = b.cout;
        for (int ch = 0; ch < b.cout; ++ch) blk_of[b.film_off + ch] = (int)i;
        if (c.with_time_emb) {
            NEED(fw, P + ".mlp.1.weight", (int64_t)2 * b.cout, td);
            NEED(fb, P + ".mlp.1.bias", (int64_t)2 * b.cout);
            std::copy(fw->data, fw->data + fw->numel(), film_w.begin() + (size_t)2 * b.film_off * n.tdim);
            std::copy(fb->data, fb->data + fb->numel(), film_b.begin() + (size_t)2 * b.film_off);
        }
    }
    UP(n.film_w, film_w); UP(n.film_b, film_b);
    UP(n.norm_a, std::vector<float>(n.total_c, 1.0f)); UP(n.norm_c, std::vector<float>(n.total_c, 0.0f));
    UP(n.blk_of, blk_of); UP(n.blk_off, blk_off); UP(n.blk_cout, blk_cout);
    for (size_t i = 0; i < r->attns.size(); ++i) {
        AttnW& a = r->attns[i];
        const std::string& P = anames[i];
        NEED(wq, P + (a.linear ? ".fn.fn.to_qkv.1.weight" : ".fn.fn.to_qkv.weight"), (int64_t)3 * HID, (int64_t)a.dim, 1, 1);
        NEED(wo, P + ".fn.fn.to_out.weight", (int64_t)a.dim, (int64_t)HID, 1, 1);
        NEED(bo, P + ".fn.fn.to_out.bias", (int64_t)a.dim);
        NEED(lg, P + ".fn.norm.g", 1, (int64_t)a.dim, 1, 1);
        UPW(a.wqkv, pack_conv(wq->data, 3 * HID, a.dim, 1), 3 * HID, 1, a.dim);
        UPW(a.wout, pack_conv(wo->data, a.dim, HID, 1), a.dim, 1, HID);
        UP(a.bout, vec(bo));
        UP(a.ln_g, vec(lg));
        if (a.linear && linattn_fused_supported(a.dim)) {
            std::vector<el16_t> fq((size_t)3 * HID * a.dim), fo((size_t)a.dim * HID);
            linattn_fused_pack(wq->data, wo->data, a.dim, fq.data(), fo.data());
            UP(a.wqkv_frag, fq);
            UP(a.wout_frag, fo);
        }
    }
    for (int l = 0; l < r->nlev; ++l) {
        SampW& s = r->downs[l];
        const std::string P = "downs." + std::to_string(l) + ".3";
        NEED(w, P + ".weight", (int64_t)s.cout, (int64_t)s.cin, (int64_t)s.k, (int64_t)s.k);
        NEED(b, P + ".bias", (int64_t)s.cout);
        UPW(s.w, pack_conv(w->data, s.cout, s.cin, s.k), s.cout, s.k * s.k, s.cin);
        UP(s.b, vec(b));
    }
    for (int u = 0; u < r->nlev; ++u) {
        SampW& s = r->ups[u];
        const std::string P = "ups." + std::to_string(u) + (s.nearest_up ? ".3.1" : ".3");
        NEED(w, P + ".weight", (int64_t)s.cout, (int64_t)s.cin, 3, 3);
        NEED(b, P + ".bias", (int64_t)s.cout);
        UPW(s.w, pack_conv(w->data, s.cout, s.cin, 3), s.cout, 9, s.cin);
        UP(s.b, vec(b));
    }
#undef NEED
#undef UP
    return DYF_OK;
}

// ------------------------------------------------------------------------------------------------ forward
dyf_status rn_forward(dyf_engine* e, int which, const Source* srcs, int nsrc, int nb, const FwdOpts& o, float* out_dev,
                      hipStream_t st) {
    Net& n = e->net[which];
    RNet* r = n.rn;
    const dyf_net_config& c = n.cfg;
    const int H = e->cfg.height, W = e->cfg.width;
    Pool pool;
    pool.free_list = r->pool;
    DropCtx dc{e, &o};
    const bool film = c.with_time_emb != 0;
    if (o.dropout_mode == 1 && (c.dropout > 0.0f || c.block_dropout1 > 0.0f || c.attn_dropout > 0.0f || c.input_dropout > 0.0f))
        HIP_TRY(e, launch_rng_begin_forward(e->rng_state, e->row_keys, nb, o.src_rows > 0 ? o.src_rows : nb, st));

#define TRY(expr)                         \
    do {                                  \
        dyf_status _s = (expr);           \
        if (_s != DYF_OK) return _s;      \
    } while (0)

    // GroupNorm fused into the producing conv (gn_fused.h): one epoch per forward, one tag per conv of the forward
    const bool gn_fuse_on = r->gn_gran != nullptr && !e->gn_fuse_disabled && o.dropout_mode != 2;
    int gn_conv_idx = 0;
    if (gn_fuse_on) HIP_TRY(e, launch_gn_epoch_bump(r->gn_epoch, st));
    // conv + GroupNorm(+FiLM) + SiLU + Dropout (+ residual) in ONE launch where a fused form serves the shape; *fused = false:
    // nothing was launched
    auto conv_gn = [&](const el16_t* s0, int c0, const el16_t* s1, int c1, int hh, int ww, int cout, const el16_t* wpk,
                       const float* bias, const float* gamma, const float* beta, bool with_film, int film_off, const DropSpec& drop,
                       const el16_t* residual, el16_t* out, bool* fused) -> dyf_status {
        *fused = false;
        if (!gn_fuse_on || gn_conv_idx >= 255) return DYF_OK;
        ConvArgs a{};
        a.src0 = s0; a.c0 = c0; a.src1 = s1; a.c1 = c1; a.n = nb; a.h = hh; a.w = ww; a.ho = hh; a.wo = ww;
        a.kh = 3; a.kw = 3; a.stride = 1; a.pad = 1; a.cout = cout; a.wpk = wpk;
        a.coef_a = r->ones; a.coef_c = bias; a.coef_stride = 0; a.coef_div = o.coef_div;
        a.act = ACT_SILU; a.drop = drop; a.residual = residual; a.out_el16 = out;
        a.n_sel = e->cfg.batch_invariant ? 2 * e->cfg.max_batch : 0;
        if (e->form_rows_scale > 1 && !e->cfg.batch_invariant) a.n_sel = nb * e->form_rows_scale;
        a.gnf.gran = r->gn_gran; a.gnf.epoch = r->gn_epoch; a.gnf.conv_tag = (uint32_t)(gn_conv_idx + 1);
        a.gnf.max_slots = r->gn_max_slots; a.gnf.groups = c.groups; a.gnf.bias = bias; a.gnf.gamma = gamma; a.gnf.beta = beta;
        if (with_film) { a.gnf.film_a = o.coef_a + film_off; a.gnf.film_c = o.coef_c + film_off; a.gnf.film_stride = o.coef_stride; }
        a.gnf.err = e->gn_err_dev;
        a.gnf.invariant = e->cfg.batch_invariant ? 1 : 0;
        a.gnf.timeout_ticks = e->gn_timeout_ticks; a.gnf.test_tag_xor = e->gn_test_tag_xor;
        const int path = (e->cfg.enable_mfma && conv_mfma_supported(a)) ? 1 : 0;
        ProfScope prof(e, e->prof_layer == DYF_PROF_RESNET_BASE + DYF_PROF_RN_CONV3_L0 && hh == e->cfg.height && ww == e->cfg.width &&
                              c0 + c1 == cout, nb, st);
        HIP_TRY(e, launch_conv_gn_fused(a, path, st, fused));
        if (*fused) ++gn_conv_idx;
        return DYF_OK;
    };

    // ResnetBlock on cat[a0 (c_a0 ch), a1 (c_a1 ch)] at hh x ww; returns the output buffer (cout channels)
    auto resblock = [&](const RBlockW& b, const el16_t* a0, int c_a0, const el16_t* a1, int c_a1, int hh, int ww,
                        el16_t** out) -> dyf_status {
        el16_t* t1 = pool.get();
        // GroupNorm statistics from the conv's fp32 accumulators where the kernel form produces them (conv_up_halo_kernel<5>)
        const bool fuse_stats = !(dyf_form("DYF_GN_CONV_STATS") && atoi(dyf_form("DYF_GN_CONV_STATS")) == 0);
        const bool ask = fuse_stats && gn_part_supported(b.cout, c.groups) &&
                         (size_t)nb * conv_halo5_gn_slots(hh, ww) * (b.cout / 8) * 2 <= r->gn_part_floats;
        int slots1 = 0, slots2 = 0;
        const DropSpec drop1 = dc.next(c.block_dropout1);
        // the shortcut first: the fused second conv adds it in its epilogue
        const el16_t* res = a0;  // identity shortcut (single source, cin == cout)
        el16_t* t3 = nullptr;
        if (b.has_res) {
            t3 = pool.get();
            TRY(rconv(e, a0, c_a0, a1, c_a1, nb, hh, ww, 1, 1, 0, b.cout, b.wr, r->ones, b.br, 0, ACT_NONE, DropSpec{}, nullptr, t3, st));
            res = t3;
        }
        bool fused1 = false;
        TRY(conv_gn(a0, c_a0, a1, c_a1, hh, ww, b.cout, b.w1, b.b1, b.g1, b.be1, film, b.film_off, drop1, b.single ? res : nullptr, t1,
                    &fused1));
        if (!fused1) {
            TRY(rconv(e, a0, c_a0, a1, c_a1, nb, hh, ww, 3, 1, 1, b.cout, b.w1, r->ones, b.b1, 0, ACT_NONE, DropSpec{}, nullptr, t1, st,
                      ask ? r->gn_part : nullptr, &slots1));
            GnActArgs g{};
            g.x = t1; g.n = nb; g.hw = hh * ww; g.c = b.cout; g.groups = c.groups; g.gamma = b.g1; g.beta = b.be1;
            if (film) { g.film_a = o.coef_a + b.film_off; g.film_c = o.coef_c + b.film_off; g.film_stride = o.coef_stride; }
            g.act = ACT_SILU; g.drop = drop1; g.residual = b.single ? res : nullptr; g.out = t1; g.stats = r->gn_stats;
            if (slots1 > 0) { g.part = r->gn_part; g.part_slots = slots1; }
            ProfScope prof(e, e->prof_layer == DYF_PROF_RESNET_BASE + DYF_PROF_RN_GN_L0 && hh == H && b.cout == c.dim, nb, st);
            HIP_TRY(e, launch_gn_act(g, st));
        }
        if (b.single) {  // double_conv_layer=False: h = block1(x); return h + residual_conv(x)
            if (t3) pool.put(t3);
            *out = t1;
            return DYF_OK;
        }
        el16_t* t2 = pool.get();
        const DropSpec drop2 = dc.next(c.dropout);
        bool fused2 = false;
        TRY(conv_gn(t1, b.cout, nullptr, 0, hh, ww, b.cout, b.w2, b.b2, b.g2, b.be2, false, 0, drop2, res, t2, &fused2));
        if (!fused2) {
            TRY(rconv(e, t1, b.cout, nullptr, 0, nb, hh, ww, 3, 1, 1, b.cout, b.w2, r->ones, b.b2, 0, ACT_NONE, DropSpec{}, nullptr, t2, st,
                      ask ? r->gn_part : nullptr, &slots2));
            GnActArgs g2{};
            g2.x = t2; g2.n = nb; g2.hw = hh * ww; g2.c = b.cout; g2.groups = c.groups; g2.gamma = b.g2; g2.beta = b.be2;
            g2.act = ACT_SILU; g2.drop = drop2; g2.residual = res; g2.out = t2; g2.stats = r->gn_stats;
            if (slots2 > 0) { g2.part = r->gn_part; g2.part_slots = slots2; }
            ProfScope prof(e, e->prof_layer == DYF_PROF_RESNET_BASE + DYF_PROF_RN_GN_L0 && hh == H && b.cout == c.dim, nb, st);
            HIP_TRY(e, launch_gn_act(g2, st));
        }
        pool.put(t1);
        if (t3) pool.put(t3);
        *out = t2;
        return DYF_OK;
    };

    // Residual(PreNorm(LayerNorm, [Linear]Attention)) on x (dim channels) at hh x ww
    auto attention = [&](const AttnW& a, const el16_t* x, int hh, int ww, el16_t** out) -> dyf_status {
        const int hw = hh * ww;
        el16_t* ln = pool.get();
        LayerNormArgs la{};
        la.x = x; la.pixels = (long long)nb * hw; la.hw = hw; la.c = a.dim; la.g = a.ln_g; la.out = ln;
        la.drop = a.linear ? dc.next(c.attn_dropout) : DropSpec{};  // LinearAttention drops its (normalised) input
        HIP_TRY(e, launch_layernorm_c(la, st));
        if (a.linear && a.wqkv_frag) {  // to_qkv, both contractions, to_out + residual: two passes over ln
            el16_t* yf = pool.get();
            LinAttnFusedArgs f{};
            f.xn = ln; f.xres = x; f.n = nb; f.hw = hw; f.c = a.dim; f.wqkv_frag = a.wqkv_frag; f.wout_frag = a.wout_frag;
            f.bout = a.bout; f.y = yf; f.scratch = r->la_scratch;
            f.scratch_floats = (long long)r->la_scratch_floats; f.groups_per_block = e->cfg.batch_invariant ? 32 : 0;
            HIP_TRY(e, launch_linear_attention_fused(f, st));
            pool.put(ln);
            *out = yf;
            return DYF_OK;
        }
        el16_t* qkv = pool.get();
        TRY(rconv(e, ln, a.dim, nullptr, 0, nb, hh, ww, 1, 1, 0, 3 * HID, a.wqkv, r->ones, r->zeros, 0, ACT_NONE, DropSpec{}, nullptr, qkv, st));
        pool.put(ln);
        el16_t* ao = pool.get();
        if (a.linear) {
            LinAttnArgs l{};
            l.qkv = qkv; l.n = nb; l.hw = hw; l.heads = HEADS; l.out = ao; l.scratch = r->la_scratch;
            HIP_TRY(e, launch_linear_attention(l, st));
        } else {
            AttnArgs t{};
            t.qkv = qkv; t.n = nb; t.hw = hw; t.heads = HEADS; t.out = ao;
            t.drop = dc.next(c.attn_dropout);  // Attention drops the softmax probabilities
            ProfScope prof(e, e->prof_layer == DYF_PROF_RESNET_BASE + DYF_PROF_RN_ATTENTION, nb, st);
            HIP_TRY(e, launch_attention(t, st));
        }
        pool.put(qkv);
        el16_t* y = pool.get();
        TRY(rconv(e, ao, HID, nullptr, 0, nb, hh, ww, 1, 1, 0, a.dim, a.wout, r->ones, a.bout, 0, ACT_NONE, DropSpec{}, x, y, st));
        pool.put(ao);
        *out = y;
        return DYF_OK;
    };

    // ---- init_conv (condition first: the caller orders the sources, unet.py:269)
    StemConvArgs sa{};
    int ctot = 0;
    for (int i = 0; i < nsrc; ++i) { sa.src[i] = srcs[i].p; sa.ch[i] = srcs[i].ch; ctot += srcs[i].ch; }
    if (ctot != n.cin_total) return fail(e, DYF_ERR_INVALID_ARGUMENT, "channel count of the network inputs does not match its configuration");
    el16_t* rbuf = pool.get();
    sa.nsrc = nsrc; sa.cin = ctot; sa.n = nb; sa.h = H; sa.w = W; sa.k = c.init_kernel_size; sa.pad = c.init_padding;
    sa.wgt = r->stem_w; sa.bias = r->stem_b; sa.dim = c.dim; sa.out = rbuf;
    sa.wfrag = r->stem_wfrag; sa.ksteps = r->stem_ksteps;
    HIP_TRY(e, launch_stem_conv(sa, st));
    el16_t* x0 = rbuf;  // what the first block reads; rbuf stays the copy for the final residual
    if (c.input_dropout > 0.0f) {
        // unet.py:276-277: r = dropout_input_for_residual(x); x = dropout_input(x) -- two sites, the residual's first
        const DropSpec dr = dc.next(c.input_dropout), dx = dc.next(c.input_dropout);
        if (dr.mode != 0 || dx.mode != 0) {
            x0 = pool.get();
            HIP_TRY(e, launch_drop16(rbuf, x0, nb, (long long)H * W * c.dim, dx, st));    // x from the raw stem output ...
            HIP_TRY(e, launch_drop16(rbuf, rbuf, nb, (long long)H * W * c.dim, dr, st));  // ... then the residual copy in place
        }
    }

    std::vector<el16_t*> skips;
    el16_t* x = x0;
    int bi = 0, ai = 0;
    for (int l = 0; l < r->nlev; ++l) {
        const int hh = r->lev_h[l], ww = r->lev_w[l], dl = r->dims[l];
        el16_t *x1, *x2, *x3;
        TRY(resblock(r->blocks[bi++], x, dl, nullptr, 0, hh, ww, &x1));
        if (x != rbuf) pool.put(x);  // (the dropped-out input copy x0 goes back to the pool here, too)
        skips.push_back(x1);
        TRY(resblock(r->blocks[bi++], x1, dl, nullptr, 0, hh, ww, &x2));
        TRY(attention(r->attns[ai++], x2, hh, ww, &x3));
        pool.put(x2);
        skips.push_back(x3);
        const SampW& s = r->downs[l];
        el16_t* xd = pool.get();
        TRY(rconv(e, x3, dl, nullptr, 0, nb, hh, ww, s.k, s.stride, s.pad, s.cout, s.w, r->ones, s.b, 0, ACT_NONE, DropSpec{}, nullptr, xd, st));
        x = xd;
    }
    {
        const int hh = r->lev_h[r->nlev - 1], ww = r->lev_w[r->nlev - 1], dm = r->dims[r->nlev];
        el16_t *m1, *m2, *m3;
        TRY(resblock(r->blocks[bi++], x, dm, nullptr, 0, hh, ww, &m1));
        pool.put(x);
        TRY(attention(r->attns[ai++], m1, hh, ww, &m2));
        pool.put(m1);
        TRY(resblock(r->blocks[bi++], m2, dm, nullptr, 0, hh, ww, &m3));
        pool.put(m2);
        x = m3;
    }
    for (int l = r->nlev - 1, u = 0; l >= 0; --l, ++u) {
        const int hh = r->lev_h[l], ww = r->lev_w[l], dout = r->dims[l + 1], din = r->dims[l];
        el16_t *y1, *y2, *y3;
        el16_t* s1 = skips.back(); skips.pop_back();
        TRY(resblock(r->blocks[bi++], x, dout, s1, din, hh, ww, &y1));
        pool.put(x); pool.put(s1);
        el16_t* s2 = skips.back(); skips.pop_back();
        TRY(resblock(r->blocks[bi++], y1, dout, s2, din, hh, ww, &y2));
        pool.put(y1); pool.put(s2);
        TRY(attention(r->attns[ai++], y2, hh, ww, &y3));
        pool.put(y2);
        const SampW& s = r->ups[u];
        el16_t* xu = pool.get();
        if (s.nearest_up) {
            // Upsample = nearest x2 + 3x3 conv (unet.py): when the conv runs on conv_up_halo_kernel<5> its halo gather reads the low-res
            // tensor at (y >> 1, x >> 1) -- no materialised upsample (a quarter of the halo traffic, one launch less); otherwise through
            // up2x_nearest_vec_kernel as before
            if (rconv_nearest_fusable(e, dout, nb, 2 * hh, 2 * ww, s.cout, s.w)) {
                TRY(rconv(e, y3, dout, nullptr, 0, nb, 2 * hh, 2 * ww, 3, 1, 1, s.cout, s.w, r->ones, s.b, 0, ACT_NONE, DropSpec{}, nullptr, xu, st,
                          nullptr, nullptr, 1));
            } else {
                el16_t* up = pool.get();
                HIP_TRY(e, launch_up2x_nearest(y3, nb, hh, ww, dout, up, st));
                TRY(rconv(e, up, dout, nullptr, 0, nb, 2 * hh, 2 * ww, 3, 1, 1, s.cout, s.w, r->ones, s.b, 0, ACT_NONE, DropSpec{}, nullptr, xu, st));
                pool.put(up);
            }
        } else {
            TRY(rconv(e, y3, dout, nullptr, 0, nb, hh, ww, 3, 1, 1, s.cout, s.w, r->ones, s.b, 0, ACT_NONE, DropSpec{}, nullptr, xu, st));
        }
        pool.put(y3);
        x = xu;
    }
    el16_t* yf;
    TRY(resblock(r->blocks[bi++], x, c.dim, rbuf, c.dim, H, W, &yf));
    pool.put(x);
    pool.put(rbuf);
    HeadArgs ha{};
    ha.x = yf; ha.n = nb; ha.hw = H * W; ha.c = c.dim; ha.cout = c.out_channels; ha.wgt = r->head_w; ha.bias = r->head_b; ha.out = out_dev;
    HIP_TRY(e, launch_head(ha, st));
    pool.put(yf);
#undef TRY
    return DYF_OK;
}

}  // namespace dyf
