// SimpleConvNet backbone (src/models/simple_conv_net.py:39-131) on the HIP engine -- SURVEY.md 8a row B7, the spring-mesh
// configuration (10x10 grid; BASELINE configs[0], the reference's own CPU-runnable case):
//
//   x = cat[inputs, condition] -> for k in kernel_sizes: [Conv2d(k, 'same') -> BatchNorm (eval) -> FiLM -> GELU -> Dropout
//   -> + residual if Cin == Cout] -> Conv2d 1x1 head
//
// Every block is ONE launch of conv_direct_kernel (conv.hip): the conv bias, the folded BatchNorm and FiLM are the (A, C)
// affine of its epilogue (coefficient tables of engine.hip::compute_coefs, same layout as unet_simple), GELU and the
// dropout mask follow, the block input is the residual operand.  The grids are tiny (100 pixels), so nothing here is
// tuned: the point of the arch is that the spring-mesh DYffusion pair runs through the same sampler / graph / ABI.
#include "engine_internal.h"
#include "unet_kernels.h"

namespace dyf {

struct SNet {
    int nk = 0;
    int ks[6] = {};
    el16_t* w[6] = {};          // [dim][k*k][cin] bf16
    float *head_w = nullptr, *head_b = nullptr;
    el16_t* packed = nullptr;   // [max_batch][H][W][cin_total]
    el16_t* buf[2] = {};        // ping-pong activations [max_batch][H][W][dim]
};

namespace {

// cat of up to 4 NCHW fp32 tensors -> NHWC bf16
__global__ void pack_inputs_kernel(const float* s0, const float* s1, const float* s2, const float* s3, int c0, int c1, int c2, int c3,
                                   int n, int hw, el16_t* out) {
    const int ctot = c0 + c1 + c2 + c3;
    const long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (long long)n * hw * ctot) return;
    const int c = (int)(idx % ctot);
    const long long pix = idx / ctot;
    const int p = (int)(pix % hw), b = (int)(pix / hw);
    float v;
    if (c < c0) v = s0[((size_t)b * c0 + c) * hw + p];
    else if (c < c0 + c1) v = s1[((size_t)b * c1 + (c - c0)) * hw + p];
    else if (c < c0 + c1 + c2) v = s2[((size_t)b * c2 + (c - c0 - c1)) * hw + p];
    else v = s3[((size_t)b * c3 + (c - c0 - c1 - c2)) * hw + p];
    out[idx] = f32_to_el16(v);
}

}  // namespace

std::string sc_configure(dyf_engine* e, Net& n) {
    const dyf_net_config& c = n.cfg;
    if (c.n_mults < 1 || c.n_mults > 6) return "kernel_sizes must have 1..6 entries (dyf_net_config.n_mults / dim_mults)";
    if (c.upsample_h != 0 || c.upsample_w != 0) return "SimpleConvNet has no outer resampler";
    SNet* s = new SNet();
    n.sc = s;
    s->nk = c.n_mults;
    double f = 0.0;
    const double px = (double)e->cfg.height * e->cfg.width;
    n.cin_total = c.in_channels + c.cond_channels;
    for (int i = 0; i < s->nk; ++i) {
        s->ks[i] = c.dim_mults[i];
        if (s->ks[i] < 1 || s->ks[i] > 15 || (s->ks[i] & 1) == 0) return "kernel sizes must be odd and <= 15 ('same' padding)";
        f += 2.0 * px * c.dim * (double)(i == 0 ? n.cin_total : c.dim) * s->ks[i] * s->ks[i];
    }
    f += 2.0 * px * c.dim * c.out_channels;
    n.dim = c.dim;
    n.tdim = 2 * c.dim;
    n.total_c = s->nk * c.dim;
    n.n_drop_sites = c.dropout > 0.0f ? s->nk : 0;
    n.flops_per_sample = f;
    return "";
}

dyf_status sc_alloc_workspace(dyf_engine* e) {
    const size_t nb = (size_t)e->cfg.max_batch, px = (size_t)e->cfg.height * e->cfg.width;
    for (int w = 0; w < 2; ++w) {
        Net& n = e->net[w];
        if (!n.sc) continue;
        dyf_status s = dev_alloc(e, &n.sc->packed, nb * px * n.cin_total);
        if (s != DYF_OK) return s;
        for (int i = 0; i < 2; ++i) {
            s = dev_alloc(e, &n.sc->buf[i], nb * px * n.dim);
            if (s != DYF_OK) return s;
        }
    }
    return DYF_OK;
}

void sc_destroy(Net& n) {
    delete n.sc;
    n.sc = nullptr;
}

dyf_status sc_load_weights(dyf_engine* e, Net& n, std::map<std::string, TensorView>& sd) {
    SNet* s = n.sc;
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
        NEED(w1, "time_emb_mlp.1.weight", td, d);
        NEED(b1, "time_emb_mlp.1.bias", td);
        NEED(w2, "time_emb_mlp.3.weight", td, td);
        NEED(b2, "time_emb_mlp.3.bias", td);
        UP(n.t_w1, vec(w1)); UP(n.t_b1, vec(b1)); UP(n.t_w2, vec(w2)); UP(n.t_b2, vec(b2));
    }
    std::vector<float> film_w((size_t)2 * n.total_c * n.tdim, 0.0f), film_b((size_t)2 * n.total_c, 0.0f);
    std::vector<float> norm_a(n.total_c), norm_c(n.total_c);
    std::vector<int> blk_of(n.total_c), blk_off(s->nk), blk_cout(s->nk);
    for (int i = 0; i < s->nk; ++i) {
        const std::string P = "convs." + std::to_string(i);
        const int64_t cin = i == 0 ? n.cin_total : d, k = s->ks[i];
        NEED(cw, P + ".conv.weight", d, cin, k, k);
        NEED(cb, P + ".conv.bias", d);
        NEED(nw, P + ".norm.weight", d);
        NEED(nbias, P + ".norm.bias", d);
        NEED(rm, P + ".norm.running_mean", d);
        NEED(rv, P + ".norm.running_var", d);
        const int taps = (int)(k * k);
        std::vector<el16_t> pk((size_t)d * taps * cin);
        for (int co = 0; co < d; ++co)
            for (int ci = 0; ci < cin; ++ci)
                for (int t = 0; t < taps; ++t)
                    pk[((size_t)co * taps + t) * cin + ci] = f32_to_el16(cw->data[((size_t)co * cin + ci) * taps + t]);
        UP(s->w[i], pk);
        const int off = i * (int)d;
        for (int ch = 0; ch < d; ++ch) {  // eval-mode BatchNorm2d folded with the conv bias: y = conv*a + c
            const double a = (double)nw->data[ch] / std::sqrt((double)rv->data[ch] + 1e-5);
            norm_a[off + ch] = (float)a;
            norm_c[off + ch] = (float)((double)nbias->data[ch] + ((double)cb->data[ch] - (double)rm->data[ch]) * a);
            blk_of[off + ch] = i;
        }
        blk_off[i] = off;
        blk_cout[i] = (int)d;
        if (c.with_time_emb) {
            NEED(fw, P + ".time_mlp.1.weight", (int64_t)2 * d, td);
            NEED(fb, P + ".time_mlp.1.bias", (int64_t)2 * d);
            std::copy(fw->data, fw->data + fw->numel(), film_w.begin() + (size_t)2 * off * n.tdim);
            std::copy(fb->data, fb->data + fb->numel(), film_b.begin() + (size_t)2 * off);
        }
    }
    NEED(hw, "head.weight", (int64_t)c.out_channels, d, 1, 1);
    NEED(hb, "head.bias", (int64_t)c.out_channels);
    UP(s->head_w, vec(hw)); UP(s->head_b, vec(hb));
    UP(n.film_w, film_w); UP(n.film_b, film_b);
    UP(n.norm_a, norm_a); UP(n.norm_c, norm_c);
    UP(n.blk_of, blk_of); UP(n.blk_off, blk_off); UP(n.blk_cout, blk_cout);
#undef NEED
#undef UP
    n.loaded = true;
    n.table_of_time.clear();
    n.ntables = 0;
    e->plan.set = false;
    return DYF_OK;
}

dyf_status sc_forward(dyf_engine* e, int which, const Source* srcs, int nsrc, int nb, const FwdOpts& o, float* out_dev,
                      hipStream_t st) {
    Net& n = e->net[which];
    SNet* s = n.sc;
    const int H = e->cfg.height, W = e->cfg.width, hw = H * W;
    int ctot = 0;
    const float* sp[4] = {nullptr, nullptr, nullptr, nullptr};
    int sc[4] = {0, 0, 0, 0};
    if (nsrc > 4) return fail(e, DYF_ERR_INVALID_ARGUMENT, "too many input tensors");
    for (int i = 0; i < nsrc; ++i) {
        sp[i] = srcs[i].p;
        sc[i] = srcs[i].ch;
        ctot += srcs[i].ch;
    }
    if (ctot != n.cin_total)
        return fail(e, DYF_ERR_INVALID_ARGUMENT, "channel count of the network inputs does not match its configuration");
    if (o.dropout_mode == 1 && n.cfg.dropout > 0.0f)
        HIP_TRY(e, launch_rng_begin_forward(e->rng_state, e->row_keys, nb, o.src_rows > 0 ? o.src_rows : nb, st));
    const long long tot = (long long)nb * hw * ctot;
    hipLaunchKernelGGL(pack_inputs_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, st, sp[0], sp[1], sp[2], sp[3], sc[0],
                       sc[1], sc[2], sc[3], nb, hw, s->packed);
    HIP_TRY(e, hipGetLastError());
    const el16_t* x = s->packed;
    int cin = ctot;
    for (int i = 0; i < s->nk; ++i) {
        ConvArgs a{};
        a.src0 = x; a.c0 = cin; a.n = nb; a.h = H; a.w = W; a.ho = H; a.wo = W;
        a.kh = s->ks[i]; a.kw = s->ks[i]; a.stride = 1; a.pad = (s->ks[i] - 1) / 2; a.cout = n.dim;
        a.wpk = s->w[i];
        a.coef_a = o.coef_a + i * n.dim; a.coef_c = o.coef_c + i * n.dim; a.coef_stride = o.coef_stride;
        a.act = ACT_GELU;
        DropSpec d{};
        const float p = n.cfg.dropout;
        d.mode = p > 0.0f ? o.dropout_mode : 0;
        d.scale = 1.0f / (1.0f - p);
        d.thresh16 = keep_threshold16(p);
        d.salt = rng_layer_salt((uint32_t)i);
        d.row_keys = e->row_keys;
        d.mask = (d.mode == 2 && o.masks) ? o.masks[i] : nullptr;
        if (d.mode == 2 && d.mask == nullptr) d.mode = 0;
        a.drop = d;
        a.residual = (cin == n.dim) ? x : nullptr;  // simple_conv_net.py:52-54 (residual=True)
        a.out_el16 = s->buf[i & 1];
        a.zero_page = e->ws.zero_page;
        HIP_TRY(e, launch_conv(a, 0, st));
        x = s->buf[i & 1];
        cin = n.dim;
    }
    HeadArgs h{};
    h.x = x; h.n = nb; h.hw = hw; h.c = n.dim; h.cout = n.cfg.out_channels; h.wgt = s->head_w; h.bias = s->head_b; h.out = out_dev;
    HIP_TRY(e, launch_head(h, st));
    return DYF_OK;
}

}  // namespace dyf
