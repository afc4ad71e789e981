// Internal types of the engine shared by engine.hip (unet_simple backbone, sampler, C ABI) and unet_resnet.hip.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "../../include/dyffusion_hip.h"
#include "conv.h"
#include "kernels.h"

#define DYF_SPLITK_FLOATS (32ll << 20)  // 128 MB: 16 splits x 2 M outputs (a layer that fills < 256 tiles of 128 x 128)

namespace dyf {

struct TrainState;  // fp32 training copies of the parameters, gradients and forward tapes (train.hip)
struct RNet;  // ResNet-UNet state (unet_resnet.hip)
struct SNet;  // SimpleConvNet state (simple_conv_net.hip)

inline thread_local std::string g_create_error;

struct UBlock {              // one UNetBlock (unet_simple.py:13-82)
    int cin = 0, cout = 0, k = 0, stride = 1, pad = 0;
    bool gn = false;         // GroupNorm(8) instead of BatchNorm (last encoder block)
    bool transposed = false; // decoder block: x2 bilinear upsample in front of the conv
    int act = ACT_NONE;
    int film_off = 0;        // offset of this block's channels in the flattened coefficient table
    int in_h = 0, in_w = 0;  // conv input size (after the x2 upsample for decoder blocks)
    int out_h = 0, out_w = 0;
    el16_t* wpk = nullptr;   // device [cout][k*k][cin]
    el16_t* wpk_up = nullptr;  // decoder 3x3 blocks: phase-decomposed weights of the fused x2-upsample conv
    el16_t* wpk_up_frag = nullptr;  // ... in MFMA fragment order (halo kernel)
    // last decoder block: column lists of the outputs the readout actually reads (plan_up_sparse_columns), or null
    int16_t* up_cols = nullptr;
    int16_t* up_cbase = nullptr;
    int16_t* up_cidx = nullptr;
    int16_t* up_col_map = nullptr;  // [out_w]: output column -> column of the compact tensor, -1 = not stored
    int up_wo_store = 0;
    int up_ntiles = 0, up_npad = 0, up_nvalid0 = 0, up_nvalid1 = 0;
    int up_mix[3] = {0, 0, 0};  // mixed list tiling (ConvArgs::up_mix), all zero = uniform tiles
    float* gamma = nullptr;  // device (GroupNorm only)
    float* beta = nullptr;
    float* static_a = nullptr;  // device [cout]: epilogue of the GroupNorm block's conv (ones / conv bias)
    float* static_c = nullptr;
};

struct Net {
    dyf_net_config cfg{};
    bool loaded = false;
    int cin_total = 0, dim = 0, tdim = 0, total_c = 0;
    int uh = 0, uw = 0;      // resampled grid
    UBlock blk[12];
    float *t_w1 = nullptr, *t_b1 = nullptr, *t_w2 = nullptr, *t_b2 = nullptr;
    float* t_learned = nullptr;  // learned_sinusoidal_cond: time_emb_mlp.0.weights (cfg.learned_sinusoidal_dim / 2 frequencies)
    float *stem_w = nullptr, *stem_b = nullptr;
    float *film_w = nullptr, *film_b = nullptr, *norm_a = nullptr, *norm_c = nullptr;
    int *blk_of = nullptr, *blk_off = nullptr, *blk_cout = nullptr;
    float *ro_w = nullptr, *ro_b = nullptr;
    el16_t* ro_wfrag = nullptr;  // readout weights as MFMA fragments (dim 64, <= 4 output channels)
    uint4 *ro_row_tab = nullptr, *ro_col_tab = nullptr;  // tap tables of the readout (launch_readout_tables), with ro_wfrag
    bool ro_tab_sparse = false;  // ... built for the compact (sparse-column) layout of the last decoder block's output
    el16_t* enc0_fused_w = nullptr;  // [2dim][4][64]: enc0's 4x4 conv composed with init_conv (+ bias channel)
    bool stem_fused = false;
    double flops_per_sample = 0.0;
    int n_drop_sites = 12;   // dropout sites with p > 0 per forward (mask-injection cursor); 12 UNetBlocks for unet_simple
    struct RNet* rn = nullptr;  // arch == DYF_ARCH_UNET_RESNET: all state lives here (unet_resnet.hip)
    struct SNet* sc = nullptr;  // arch == DYF_ARCH_SIMPLE_CONV_NET (simple_conv_net.hip)
    // sampler coefficient tables: one (A, C) row pair per distinct time value
    std::map<float, int> table_of_time;
    float* tables = nullptr;  // device [ntables][2][total_c]
    int ntables = 0;
};

struct Workspace {
    el16_t* stem = nullptr;
    el16_t* stem16 = nullptr;  // fused stem: [nb][uh+2][uw+2][16]
    el16_t* enc[6] = {};
    float* enc5_raw = nullptr;
    el16_t* up = nullptr;
    el16_t* dec[6] = {};
    float* silu = nullptr;
    float* coef_a = nullptr;
    float* coef_c = nullptr;
    el16_t* zero_page = nullptr;
    float* up_border = nullptr;  // border-correction scratch of the fused-upsample halo convs (ConvArgs::up_border)
    float* coef_pair = nullptr;  // [2][2][total_c]: FiLM coefficient rows of a paired interpolator call
    float* splitk = nullptr;     // split-K partials of the small-batch convs (ConvArgs::splitk_ws), DYF_SPLITK_FLOATS floats
    float* ones_f = nullptr;     // [max cout] 1.0f / 0.0f: identity epilogue of the commuted 1 x 1 decoder convs (raw fp32 output)
    float* zeros_f = nullptr;
};

struct PlanHost {
    bool set = false;
    std::vector<dyf_plan_step> steps;
    std::vector<float> refine_times;
    std::vector<int> refine_slots;
    dyf_plan hdr{};
};

struct GraphEntry {
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
};

}  // namespace dyf

struct dyf_engine {
    dyf_engine_config cfg{};
    std::string err;
    dyf::Net net[2];
    dyf::Workspace ws;
    std::vector<void*> allocs;          // lifetime of the engine (workspace, sampler state)
    std::vector<void*> net_allocs[2];   // packed weights of one network: released when dyf_load_weights is called again
    std::vector<void*> plan_allocs;     // coefficient tables of the current plan: released by the next dyf_set_plan
    std::vector<void*>* alloc_sink = nullptr;  // where dev_alloc records (AllocScope), null = allocs
    dyf::PlanHost plan;
    int C = 0, Cs = 0, wC = 0;  // dynamics channels, static-condition channels, window*C
    // sampler state (fp32 NCHW, engine-owned so a captured graph never sees caller pointers)
    float *s_init = nullptr, *s_static = nullptr, *s_xs = nullptr, *s_x0hat = nullptr, *s_next = nullptr,
          *s_cur = nullptr, *s_noisy = nullptr, *s_stack = nullptr;
    float* s_time = nullptr;   // device scalar scratch for time values
    uint32_t* rng_state = nullptr;  // device {seed_lo, seed_hi, forward counter, row offset, noise counter, ...} (kernels.h)
    uint32_t row_offset = 0;        // host copy of rng_state[3] (dyf_set_row_offset skips the device write when unchanged)
    bool row_offset_known = false;
    uint32_t* row_keys = nullptr;   // device [2 max_batch][2]: per-row stream keys of the forward being launched (common.h)
    int stack_slots = 0;
    std::map<int, dyf::GraphEntry> graphs;  // by batch size
    int fuse_min_plane = 32;           // smallest low-res plane side for which the fused form is used
    bool fuse_stem = true;             // DYF_FUSE_STEM=0: separate 1x1 stem kernel + plain enc0
    bool fuse_up2x = true;             // DYF_FUSE_UP2X=0 falls back to the materialised upsample (A/B testing)
    hipStream_t cap_stream = nullptr;  // capture never runs on the caller's (possibly legacy default) stream
    // dyf_time_layer_in_rollout: events recorded around the conv of block prof_layer while a rollout runs eagerly
    double* metric_sums = nullptr;  // dyf_ensemble_metrics accumulators (device)
    int prof_layer = -1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> prof_ev;
    std::vector<int> prof_rows;
    float* s_pair = nullptr;        // [2][max_batch][C][H][W]: outputs of a paired interpolator call
    float* refine_coef = nullptr;   // [n_refine][2][total_c]: FiLM rows of the refinement pass in refine order (plan allocation)
    float* pair_coef = nullptr;     // [n_steps][2 rows][2][total_c]: FiLM rows (i_next, i_cur) of every step's paired interpolator call (plan allocation)
    bool pair_interp = true;        // DYF_PAIR_INTERP=0: one forward per interpolator call (A/B testing)
    // engine-owned exchange (dyf_comm_init / dyf_sample_gather): RCCL communicator (ncclComm_t) and the all-gather receive buffer
    void* comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    float* gather_recv = nullptr;
    size_t gather_recv_floats = 0;
    // row groups (dyf_set_row_groups): child engines with max_batch / G rows each; a dyf_sample call of enough rows is split over
    // them, every share a rollout of its own (own workspace, own captured graph) on its own stream, so kernels of different groups
    // overlap: tails of under-filled launches and launch gaps of one group are covered by the others
    std::vector<dyf_engine*> groups;
    bool is_group_child = false;
    int train_precision = 0;  // dyf_train_set_precision: 0 = DYF_TRAIN_OPERANDS decides (default fp32), 32, 16
    int form_rows_scale = 1;             // child: kernel forms are chosen for this many times the rows of a launch (the siblings' share)
    hipStream_t group_stream = nullptr;  // child: the stream its share runs on
    hipEvent_t group_done = nullptr;     // child: recorded behind its share
    hipEvent_t group_fork = nullptr;     // parent: recorded on the caller's stream in front of the shares
    int group_min_rows = 16;             // a call with fewer rows per group than this runs ungrouped
    int last_groups = 0, last_per = 0, last_nb = 0;  // split of the most recent sampling call (dyf_get_sampler_state)
    // log_every_t (dyf_set_log_intermediates): per sampling step {x0_hat, x_interpolated_s_next, x_interpolated_s} copied out of
    // the rollout, [n_steps][3][max_batch][C][H][W] fp32; such calls run eagerly on the engine itself (no graph, no row groups)
    bool log_on = false;
    float* s_log = nullptr;
    size_t s_log_floats = 0;
    int log_nb = 0;                  // batch rows of the logged call
    std::vector<uint8_t> log_has_cur;  // per step: slot 2 (x_interpolated_s) was defined at that step
    dyf::TrainState* train = nullptr;  // training path (arch unet_simple), created by the first dyf_load_weights
    bool last_dec5_sparse = false;  // the most recent unet_simple forward stored dec5 in the compact sparse-column layout
    bool poison_dec5 = false;       // DYF_POISON_DEC5=1 (test hook, read once at create): NaN-fill dec5's output before its conv
    // GroupNorm fused into the producing conv (gn_fused.h): host-visible error word (pinned, mapped) a timed-out granule sweep raises,
    // its device alias, and the switch that sends later forwards down the three-kernel path once that has happened
    uint32_t* gn_err_host = nullptr;
    uint32_t* gn_err_dev = nullptr;
    bool gn_fuse_disabled = false;
    uint32_t gn_timeout_ticks = 0;   // dyf_debug_gn_fuse: sweep bound in 100 MHz ticks (0 = the 2 s default)
    uint32_t gn_test_tag_xor = 0;    // dyf_debug_gn_fuse: nonzero = every sweep waits for a tag nobody publishes (forced time-out)
    hipStream_t poll_stream = nullptr;  // the stream of the last dyf_net_forward / dyf_sample / dyf_sample_gather: what dyf_poll_errors waits for
    int gn_fuse_downgrades = 0;      // times this engine left the fused path (time-out or slow sweep): dyf_gn_fuse_state
};

namespace dyf {

// dyf_time_layer_in_rollout / dyf_time_kernel_in_rollout: HIP events on the launch stream around the launches of one kernel
// class while a rollout runs eagerly.  `match` = this launch belongs to the class e->prof_layer names.
struct ProfScope {
    dyf_engine* e;
    hipStream_t st;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    ProfScope(dyf_engine* eng, bool match, int rows, hipStream_t s) : e(eng), st(s) {
        if (!match || e->prof_ev.size() >= 4096) return;
        if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, st) != hipSuccess) {
            e0 = e1 = nullptr;
            return;
        }
        e->prof_rows.push_back(rows);
    }
    ~ProfScope() {
        if (!e0) return;
        (void)hipEventRecord(e1, st);
        e->prof_ev.emplace_back(e0, e1);
    }
};
// kernel classes of the ResNet-UNet path (dyf_time_kernel_in_rollout `kind`); e->prof_layer = DYF_PROF_RESNET_BASE + kind
#define DYF_PROF_RESNET_BASE 100
#define DYF_PROF_RN_CONV3_L0 0   // 3x3 WS-convs of the full-resolution level with cin == cout == dim
#define DYF_PROF_RN_ATTENTION 1  // bottleneck Attention core (flash kernel)
#define DYF_PROF_RN_GN_L0 2      // GroupNorm(+FiLM+SiLU+dropout(+residual)) chain of the full-resolution level, dim channels

// ------------------------------------------------------------------------------------------------ error helpers
inline dyf_status fail(dyf_engine* e, dyf_status st, const std::string& msg) {
    if (e) e->err = msg; else g_create_error = msg;
    return st;
}

#define HIP_TRY(e, expr)                                                                                   \
    do {                                                                                                   \
        hipError_t _err = (expr);                                                                          \
        if (_err != hipSuccess)                                                                            \
            return fail(e, DYF_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_err));              \
    } while (0)

template <typename T>
dyf_status dev_alloc(dyf_engine* e, T** out, size_t count) {
    void* p = nullptr;
    size_t bytes = std::max<size_t>(count * sizeof(T), 256);
    HIP_TRY(e, hipMalloc(&p, bytes));
    HIP_TRY(e, hipMemset(p, 0, bytes));
    (e->alloc_sink ? *e->alloc_sink : e->allocs).push_back(p);
    *out = (T*)p;
    return DYF_OK;
}

// device allocations made while the scope is alive are recorded in `list`
struct AllocScope {
    dyf_engine* e;
    std::vector<void*>* prev;
    AllocScope(dyf_engine* eng, std::vector<void*>* list) : e(eng), prev(eng->alloc_sink) { e->alloc_sink = list; }
    ~AllocScope() { e->alloc_sink = prev; }
};

// free every allocation of `list` (and their registered fragment copies); the device must be idle
inline void release_allocs(std::vector<void*>& list) {
    for (void* p : list) {
        conv_unregister_frag(p);
        (void)hipFree(p);
    }
    list.clear();
}

template <typename T>
dyf_status dev_upload(dyf_engine* e, T** out, const std::vector<T>& host) {
    dyf_status st = dev_alloc(e, out, host.size());
    if (st != DYF_OK) return st;
    if (!host.empty()) HIP_TRY(e, hipMemcpy(*out, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice));
    return DYF_OK;
}

// packed conv weights [cout][taps][cin] bf16 -> device; layers the second implicit-GEMM form can run also get their
// fragment-ordered copy, registered under the primary pointer (launch_conv looks it up)
inline dyf_status upload_conv_weights(dyf_engine* e, el16_t** out, const std::vector<el16_t>& pk, int cout, int taps, int cin) {
    dyf_status st = dev_upload(e, out, pk);
    if (st != DYF_OK) return st;
    if (cout % 64 == 0 && cin % 64 == 0 && taps <= 32 && (size_t)cout * taps * cin == pk.size()) {
        std::vector<el16_t> pf(pk.size());
        pack_conv_frag(pk.data(), cout, taps, cin, pf.data());
        el16_t* frag = nullptr;
        st = dev_upload(e, &frag, pf);
        if (st != DYF_OK) return st;
        conv_register_frag(*out, frag);
    }
    if (taps == 16 && cout % 128 == 0 && cin % 64 == 0 && (size_t)cout * taps * cin == pk.size()) {  // halo form of 4x4 / s2
        std::vector<el16_t> pf((size_t)cout * 16 * 4 * cin);
        pack_halo_s2_frag(pk.data(), cout, cin, pf.data());
        el16_t* frag = nullptr;
        st = dev_upload(e, &frag, pf);
        if (st != DYF_OK) return st;
        conv_register_halo3_frag(*out, frag);
    }
    const bool h5_all = dyf_form("DYF_HALO5_ALL") && atoi(dyf_form("DYF_HALO5_ALL")) != 0;  // experiment: SP = 5 for every 3x3
    if (taps == 9 && cout % 64 == 0 && (cout % 256 != 0 || h5_all) && cin % 64 == 0 && (size_t)cout * taps * cin == pk.size()) {
        std::vector<el16_t> pf((size_t)cout * 16 * cin);  // halo form of plain 3x3 convs with 64 / 128 output channels (SP = 5)
        pack_halo3_frag64(pk.data(), cout, cin, pf.data());
        el16_t* frag = nullptr;
        st = dev_upload(e, &frag, pf);
        if (st != DYF_OK) return st;
        conv_register_halo3_frag(*out, frag);
    }
    if (taps == 9 && cout % 256 == 0 && !h5_all && cin % 64 == 0 && (size_t)cout * taps * cin == pk.size()) {  // halo form of plain 3x3
        std::vector<el16_t> pf((size_t)cout * 16 * cin);
        pack_halo3_frag(pk.data(), cout, cin, pf.data());
        el16_t* frag = nullptr;
        st = dev_upload(e, &frag, pf);
        if (st != DYF_OK) return st;
        conv_register_halo3_frag(*out, frag);
        // ... and the 64-channel-block order for conv_gn16_kernel (the fused-GroupNorm 3 x 3 convs of the ResNet-UNet's 256-channel level)
        pack_halo3_frag64(pk.data(), cout, cin, pf.data());
        el16_t* frag64 = nullptr;
        st = dev_upload(e, &frag64, pf);
        if (st != DYF_OK) return st;
        conv_register_frag64(*out, frag64);
    }
    return DYF_OK;
}

// host view of one state_dict tensor (dyf_load_weights)
struct TensorView {
    const float* data;
    std::vector<int64_t> shape;
    int64_t numel() const {
        int64_t n = 1;
        for (auto s : shape) n *= s;
        return n;
    }
};


struct Source {
    const float* p;
    int ch;
};

struct FwdOpts {
    const float* coef_a;      // [rows][total_c]
    const float* coef_c;
    int coef_stride;          // 0: one row for the whole batch
    int dropout_mode;         // 0 off, 1 engine RNG, 2 injected
    const uint8_t* const* masks;  // [12] when dropout_mode == 2
    int src_rows = 0;         // rows of the source tensors (0: = nb); row r of the batch reads source row r % src_rows
    int coef_div = 0;         // batch rows per coefficient row (0/1: coef_stride semantics unchanged)
};


}  // namespace dyf

// ---- training step (train.hip)
namespace dyf {
dyf_status train_store_weights(dyf_engine* e, int which, std::map<std::string, TensorView>& sd);
dyf_status rn_train_store_weights(dyf_engine* e, int which, std::map<std::string, TensorView>& sd);  // arch unet.Unet
void train_destroy(dyf_engine* e);
}  // namespace dyf

// ---- SimpleConvNet backbone (src/models/simple_conv_net.py), implemented in simple_conv_net.hip
namespace dyf {
std::string sc_configure(dyf_engine* e, Net& n);
dyf_status sc_alloc_workspace(dyf_engine* e);
dyf_status sc_load_weights(dyf_engine* e, Net& n, std::map<std::string, TensorView>& sd);
dyf_status sc_forward(dyf_engine* e, int which, const Source* srcs, int nsrc, int nb, const FwdOpts& o, float* out_dev,
                      hipStream_t st);
void sc_destroy(Net& n);
}  // namespace dyf

// ---- ResNet-UNet backbone (src/models/unet.py), implemented in unet_resnet.hip
namespace dyf {
std::string rn_configure(dyf_engine* e, Net& n);   // "" or an error message; fills geometry + flops
dyf_status rn_alloc_workspace(dyf_engine* e);
dyf_status rn_load_weights(dyf_engine* e, Net& n, std::map<std::string, TensorView>& sd);
dyf_status rn_forward(dyf_engine* e, int which, const Source* srcs, int nsrc, int nb, const FwdOpts& o, float* out_dev,
                      hipStream_t st);
void rn_destroy(Net& n);
}  // namespace dyf
