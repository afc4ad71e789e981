/*
 * dyffusion_hip.h -- C ABI of libdyffusion_hip.so: the MI355X (gfx950) DYffusion sampling engine.
 *
 * Drop-in boundary for the reference's hot path (all paths relative to the upstream repo root):
 *   - per-network seam   BaseModel.forward(inputs, time, condition)        src/models/unet_simple.py:181-197
 *   - sampler seam       DYffusion.sample(initial_condition, static_condition=..., num_predictions=N)
 *                                                                          src/diffusion/dyffusion.py:335-431
 *     reached from BaseDiffusion.predict_forward (src/diffusion/_base_diffusion.py:48-68) and
 *     BaseExperiment.predict (src/experiment_types/_base_experiment.py:315-356).
 *
 * Conventions
 *   - plain C types only; every pointer named *_dev is DEVICE memory owned by the caller (e.g. a torch tensor's
 *     data_ptr()), everything else is HOST memory.  Tensors crossing the ABI are fp32, NCHW, contiguous -- the
 *     reference's layout.  Internally the engine keeps activations NHWC in a 16-bit format (bf16, or fp16 in the
 *     libdyffusion_hip_f16.so build) and accumulates in fp32.
 *   - the engine owns packed device weights, its workspace arena and captured hipGraphs; no allocation happens
 *     inside dyf_sample / dyf_net_forward after the first call for a given batch size.
 *   - one engine per (device, stream); calls on one engine are not re-entrant.
 *   - every function returns a dyf_status; dyf_last_error(engine) (or NULL for create-time failures) returns a
 *     message.  The Python shim re-raises ValueError / AssertionError / RuntimeError from these.
 *   - `stream` arguments are hipStream_t passed as void* (0 = the null stream).
 */
#ifndef DYFFUSION_HIP_H
#define DYFFUSION_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DYF_ABI_VERSION 8

typedef struct dyf_engine dyf_engine;

typedef enum dyf_status {
    DYF_OK = 0,
    DYF_ERR_INVALID_ARGUMENT = 1, /* maps to ValueError / AssertionError in the shim */
    DYF_ERR_UNSUPPORTED = 2,      /* valid in the reference, not implemented by this engine (NotImplementedError) */
    DYF_ERR_HIP = 3,              /* a HIP runtime call failed (RuntimeError) */
    DYF_ERR_STATE = 4             /* call order violated, e.g. sampling before weights / plan are set */
} dyf_status;

/* Which network of the DYffusion pair a call refers to (dyffusion.py:448-468: self.model / self.interpolator). */
typedef enum dyf_net_id { DYF_NET_FORECASTER = 0, DYF_NET_INTERPOLATOR = 1 } dyf_net_id;

/* Backbone architectures (src/models/). */
typedef enum dyf_arch {
    DYF_ARCH_UNET_SIMPLE = 0, /* src/models/unet_simple.py:85-197 (Navier-Stokes / spring-mesh backbone) */
    DYF_ARCH_UNET_RESNET = 1, /* src/models/unet.py:112-315 (OISST / synthetic backbone: ResnetBlocks + attention) */
    DYF_ARCH_SIMPLE_CONV_NET = 2 /* src/models/simple_conv_net.py:58-131 (spring-mesh backbone); kernel_sizes travel in
                                  * n_mults / dim_mults, dropout in `dropout` */
} dyf_arch;

/* Hyper-parameters of one backbone: the kwargs of unet_simple.UNet.__init__ (unet_simple.py:86-95) plus the
 * channel bookkeeping of BaseModel (src/models/_base_model.py:43-66). */
typedef struct dyf_net_config {
    int32_t arch;             /* dyf_arch */
    int32_t in_channels;      /* num_input_channels (without the conditional channels) */
    int32_t cond_channels;    /* num_conditional_channels */
    int32_t out_channels;     /* num_output_channels */
    int32_t dim;              /* base width */
    int32_t with_time_emb;    /* 0/1 */
    int32_t upsample_h;       /* upsample_dims[0], 0 = no outer resampling */
    int32_t upsample_w;
    float dropout;            /* unet_simple: UNetBlock dropout p; unet: block_dropout (2nd Block of a ResnetBlock) */
    float input_dropout;      /* Dropout on init_conv's output: unet_simple.py:116,168 (one site, the first of a forward; the engine then
                               * runs the un-fused stem), unet.py:162-163,276-277 (two sites: residual copy, input); 0 in every shipped config */
    /* ---- unet.Unet only (kwargs of Unet.__init__, src/models/unet.py:113-135) ---- */
    int32_t n_mults;          /* len(dim_mults), <= 6 */
    int32_t dim_mults[6];
    float block_dropout1;     /* 1st Block of a ResnetBlock */
    float attn_dropout;       /* LinearAttention input dropout / Attention probability dropout */
    int32_t groups;           /* resnet_block_groups */
    int32_t init_kernel_size; /* 7 */
    int32_t init_padding;     /* 3 */
    int32_t outer_nearest;    /* unet_simple outer_sample_mode: 0 "bilinear", 1 "nearest" (upsampler + final F.interpolate,
                               * unet_simple.py:103,195) */
    /* ---- unet.Unet options no shipped config sets (unet.py:127-135); 0 = the shipped default ---- */
    int32_t keep_spatial_dims;      /* 1: no down / up sampling, plain 3x3 convs between the levels (unet.py:190,214) */
    int32_t single_conv_layer;      /* 1: double_conv_layer=False, a ResnetBlock's second Block is nn.Identity (unet.py:94) */
    int32_t learned_sinusoidal_dim; /* > 0: learned_sinusoidal_cond=True with this (even) dim: time features
                                     * [t, sin(2 pi t w), cos(2 pi t w)] with the parameter time_emb_mlp.0.weights (misc.py:35-59) */
} dyf_net_config;

typedef enum dyf_dtype_id { DYF_DTYPE_BF16 = 0, DYF_DTYPE_F16 = 1 } dyf_dtype_id;

typedef struct dyf_engine_config {
    int32_t abi_version;      /* DYF_ABI_VERSION */
    int32_t device;           /* HIP device ordinal */
    int32_t height, width;    /* native grid (e.g. 221 x 42) */
    int32_t max_batch;        /* largest NB = N_ensemble * B the workspace is sized for */
    int32_t use_graph;        /* capture the rollout in a hipGraph (dyf_sample) */
    int32_t enable_mfma;      /* 1: implicit-GEMM MFMA conv where shapes allow; 0: direct conv everywhere (debug) */
    int32_t dtype;            /* dyf_dtype of activations / weights in HBM and of the MFMA operands; must equal dyf_dtype() of
                               * the library: libdyffusion_hip.so is the bf16 build, libdyffusion_hip_f16.so the fp16 build of the
                               * same sources (-DDYF_F16=1), same ABI */
    int32_t batch_invariant;  /* 1: the kernel form of every layer (implicit-GEMM variant, split-K factor, halo forms) is chosen for
                               * 2 * max_batch rows whatever the batch of a call, so a row's result is bit-identical under any
                               * batching / sharding served by engines of equal max_batch; 0: chosen per call (fastest) */
    dyf_net_config net[2];    /* indexed by dyf_net_id */
} dyf_engine_config;

/* One iteration of the sampling loop, resolved on the host (dyffusion.py:352-397). */
typedef struct dyf_plan_step {
    float forecaster_time;    /* enc(s): value fed to the forecaster's time embedding */
    float tau;                /* s/(T-1), used by forward_conditioning="data+noise" */
    float i_next;             /* interpolation time of s_next, < 0: none (x_next = x0_hat) */
    float i_cur;              /* interpolation time of s,      < 0: none (x_cur  = x_s)    */
    int32_t is_last;          /* s == T-1 */
    int32_t out_slot;         /* index into the forecast stack written after this step, -1: none */
} dyf_plan_step;

typedef enum dyf_fcond { DYF_FCOND_NONE = 0, DYF_FCOND_DATA = 1, DYF_FCOND_DATA_NOISE = 2 } dyf_fcond;

typedef struct dyf_plan {
    int32_t n_steps;
    const dyf_plan_step* steps;
    int32_t sampling_cold;            /* 1 = "cold", 0 = "naive" (dyffusion.py:381-391) */
    int32_t cold_for_last_step;       /* use_cold_sampling_for_last_step */
    int32_t forward_conditioning;     /* dyf_fcond */
    int32_t n_refine;                 /* refinement pass (dyffusion.py:408-422): number of re-predicted times */
    const float* refine_times;        /* [n_refine] interpolation times */
    const int32_t* refine_slots;      /* [n_refine] forecast-stack slots they overwrite */
    int32_t n_out_slots;              /* h: number of (NB,C,H,W) fields in the forecast stack */
    int32_t interpolator_dropout;     /* MC dropout active in the interpolator (enable_interpolator_dropout) */
    int32_t forecaster_dropout;       /* dropout active in the forecaster (module.enable_inference_dropout) */
} dyf_plan;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
dyf_status dyf_engine_create(const dyf_engine_config* cfg, dyf_engine** out_engine);
void dyf_engine_destroy(dyf_engine* engine);
const char* dyf_last_error(const dyf_engine* engine);
int32_t dyf_abi_version(void);
int32_t dyf_dtype(void);   /* dyf_dtype_id this library was built for */

/* ---- weights: the reference's state_dict (names as in UNet.state_dict(), host fp32, contiguous) ---------- */
/* Replaces BaseExperiment.instantiate_model / load_state_dict (_base_experiment.py:173-199).  Folds eval-mode
 * BatchNorm into per-channel scale/shift, repacks conv weights to the MFMA layout and uploads as bf16. */
dyf_status dyf_load_weights(dyf_engine* engine, int32_t net, int32_t n_tensors, const char* const* names,
                            const float* const* data, const int64_t* const* shapes, const int32_t* ndims);

/* ---- per-network seam: BaseModel.forward(inputs, time, condition) (unet_simple.py:181-197) ------------------ */
/* inputs_dev (NB,in_channels,H,W), time_dev (NB) or NULL when with_time_emb == 0, condition_dev
 * (NB,cond_channels,H,W) or NULL, out_dev (NB,out_channels,H,W).  dropout_mode: 0 off (eval), 1 on (engine RNG),
 * 2 on with injected keep-masks: masks_dev[l] is the uint8 keep-mask of the l-th dropout site with p > 0 in
 * execution order (NHWC for activations, (NB,heads,N,N) for attention probabilities), used by the parity tests. */
dyf_status dyf_net_forward(dyf_engine* engine, int32_t net, const float* inputs_dev, const float* time_dev,
                           const float* condition_dev, float* out_dev, int32_t nb, int32_t dropout_mode,
                           const uint8_t* const* masks_dev, void* stream);

/* ---- sampler seam: DYffusion.sample / sample_loop (dyffusion.py:335-431) ------------------------------------- */
dyf_status dyf_set_plan(dyf_engine* engine, const dyf_plan* plan);
/* initial_dev (NB, window*C, H, W); static_dev (NB, Cs, H, W) or NULL; out_dev (n_out_slots, NB, C, H, W): slot i
 * holds t{i+1}_preds.  masks_dev: NULL, or one pointer per (interpolator/forecaster forward, dropout layer) in
 * execution order (12 per forward that has dropout on) for the mask-injection parity mode; noise_dev: NULL or
 * (n_steps, NB, window*C, H, W) standard-normal draws for forward_conditioning="data+noise" parity. */
dyf_status dyf_sample(dyf_engine* engine, const float* initial_dev, const float* static_dev, float* out_dev,
                      int32_t nb, const uint8_t* const* masks_dev, const float* noise_dev, void* stream);
/* Asynchronous failures (ABI 7).  Every entry point only ENQUEUES work on the caller's stream; the one thing that can go wrong
 * after that is a fused GroupNorm convolution of the ResNet-UNet (csrc/gn_fused.h: workgroups of one sample meet inside the launch)
 * whose wait for its sample's statistics times out -- its output is then NaN-poisoned and a host-visible word is raised.
 * dyf_poll_errors reports that: DYF_OK, or DYF_ERR_STATE naming it (the engine has then already dropped its captured graphs and
 * runs the un-fused GroupNorm kernels from now on: repeat the call).  synchronize != 0 first waits for the STREAM of the engine's last
 * dyf_sample / dyf_sample_gather / dyf_net_forward call (not for the device: other streams keep running; a stream under capture is not waited for), so a
 * caller that polls right after dyf_sample / dyf_sample_gather / dyf_net_forward gets the failure from the SAME call (the Python
 * wrappers do).  Engines without a live fused form (unet_simple, SimpleConvNet, after a downgrade) return
 * DYF_OK at once without waiting.  Un-polled failures are still reported at the head of the next entry point.
 * dyf_gn_fuse_state: *live = the fused form is in use, *downgrades = how often the engine left it (time-out, or sweeps slower than
 * 1 ms on a GPU shared with another process -- the latter is not an error; DYF_VERBOSE=1 logs either to stderr). */
dyf_status dyf_poll_errors(dyf_engine* engine, int32_t synchronize);
dyf_status dyf_gn_fuse_state(const dyf_engine* engine, int32_t* live, int32_t* downgrades);
/* log_every_t of sample_loop (dyffusion.py:339-344, 398-406): with logging enabled a dyf_sample call also keeps, per sampling step,
 * x0_hat (what = 0, the reference's `intermediate_{s}_x0hat`), x_interpolated_s_next (1: `xipol_{s}_dmodel`, and `t{k}_preds2` on
 * the steps that emit a forecast) and, for cold sampling, x_interpolated_s (2: `xipol_{s}_dmodel2`; as in the reference the last
 * step without cold sampling reports the previous step's value).  Logged calls run eagerly on the engine's own workspace (no
 * captured graph, no row groups).  dyf_get_log copies one (NB, C, H, W) tensor of the most recent logged call; `step` indexes the
 * plan's steps. */
dyf_status dyf_set_log_intermediates(dyf_engine* engine, int32_t enable);
dyf_status dyf_get_log(dyf_engine* engine, int32_t step, int32_t what, float* out_dev, int32_t nb, void* stream);
/* Re-seed the engine's counter-based dropout / noise generator (forward and noise counters reset to 0).  The keep bit of
 * an element is a function of (seed, forward index, GLOBAL batch row, dropout layer, element index inside the row), see
 * csrc/common.h: a rollout does not depend on how its rows are batched or sharded over GPUs.  dyf_seed and dyf_set_row_offset
 * wait for the device to go idle before they write the generator state (hipDeviceSynchronize), so they are ordered against
 * rollouts in flight on ANY stream. */
dyf_status dyf_seed(dyf_engine* engine, uint64_t seed);
/* Global index of this engine's batch row 0 (default 0).  A rank that samples rows [lo, hi) of an N*B-row ensemble
 * (_base_experiment.py:503-538 tiles them, row = n*B + b) sets lo: its rows then draw exactly the masks / noise they
 * would draw inside the un-sharded batch. */
dyf_status dyf_set_row_offset(dyf_engine* engine, uint32_t first_row);
/* Row groups (ABI 5).  The rows of a sampling call are independent for the whole rollout, so the engine may run them as n_groups
 * concurrent rollouts of ceil(NB / n_groups) rows each -- own workspace, own packed weights, own captured hipGraph, own HIP stream,
 * forked from and joined into the caller's stream -- so that under-filled launches, ragged last rounds of workgroups and launch
 * gaps of one group are covered by the kernels of the others.  Row g*per + i of the call is row i of group g and draws the
 * masks / noise of global row (row offset + g*per + i): the generator streams are those of the ungrouped call.  Default: chosen
 * at dyf_engine_create from the architecture and max_batch (ResNet-UNet on planes <= 128 x 128, not batch_invariant: 3 groups from
 * 432 000 pixels x rows = 120 rows of 60 x 60, 2 from 230 400 = 64 rows; otherwise 1; environment DYF_ROW_GROUPS overrides).  Must be called before dyf_load_weights.
 * Calls with fewer than 32 rows, with injected masks / noise, and every other entry point run on the engine itself.
 * Three groups plus the caller's stream fill the 4 hardware queues of a HIP process: with other busy streams or other live engines
 * in the process two groups are the robust choice.
 * dyf_row_groups returns the number of groups in effect (1 = none). */
dyf_status dyf_set_row_groups(dyf_engine* engine, int32_t n_groups);
int32_t dyf_row_groups(const dyf_engine* engine);

/* ---- engine-owned exchange of the ensemble-sharded path (one process per GPU; the reference has no inference collective) ------- */
/* Ensemble members / batch items are independent rows for the whole rollout (_base_experiment.py:503-538 tiles them, row = n*B + b),
 * so rank r of `world` samples the contiguous block of global rows [lo_r, hi_r) of a total_rows-row ensemble -- balanced split, the
 * first total_rows % world ranks own one row more -- with NO collective inside the rollout, and ONE RCCL all-gather (xGMI) of the
 * local forecast stack at the end.  The engine owns the communicator: a host binding this ABI needs no torch.distributed.
 *   dyf_comm_unique_id   rank 0 creates the 128-byte RCCL unique id; the caller distributes it (MPI, a file, torch.distributed ...)
 *   dyf_comm_init        every rank, same id: ncclCommInitRank on the engine's device (librccl is dlopen'ed here, on first use)
 *   dyf_sample_gather    dyf_sample of this rank's nb = ceil(total_rows / world) rows (ranks owning fewer repeat a row; call
 *                        dyf_set_row_offset(lo_r) first so the rows draw the masks of the un-sharded batch), then ncclAllGather of the
 *                        contiguous [n_out_slots][nb][C][H][W] stack and one unpack pass, all enqueued on `stream`:
 *                        out_full_dev (n_out_slots, total_rows, C, H, W) holds t{i+1}_preds of ALL rows, global order, on every rank. */
#define DYF_COMM_ID_BYTES 128
dyf_status dyf_comm_unique_id(uint8_t* id_out /* [DYF_COMM_ID_BYTES], host */);
dyf_status dyf_comm_init(dyf_engine* engine, const uint8_t* unique_id, int32_t rank, int32_t world);
dyf_status dyf_comm_destroy(dyf_engine* engine);
/* ranks of the engine's communicator as RCCL itself reports them (ncclCommCount); 0 = the engine owns none (ABI 6) */
dyf_status dyf_comm_count(const dyf_engine* engine, int32_t* ranks_out);
dyf_status dyf_sample_gather(dyf_engine* engine, const float* initial_dev, const float* static_dev, float* out_full_dev, int32_t nb,
                             int32_t total_rows, void* stream);

/* On-device ensemble metrics, replaces evaluate_ensemble_prediction (src/utilities/evaluation.py:10-118) and the
 * .cpu().numpy() round trip in front of it (_base_experiment.py:617-640).  preds_dev: (n_members, n_points) fp32 with
 * n_points = B*C*H*W (the "(N, B, C, H, W)" ensemble stack of one horizon step), targets_dev: (n_points) fp32.
 * out_host[3] receives {mse of the ensemble mean, spread-skill ratio sqrt(mean var)/sqrt(mse), CRPS}, all averaged over
 * every point (mean_over_samples=True).  Synchronises `stream`. */
dyf_status dyf_ensemble_metrics(dyf_engine* engine, const float* preds_dev, const float* targets_dev, int32_t n_members,
                                int64_t n_points, double* out_host, void* stream);   /* n_members <= 15 360: up to 64 members a workgroup stages
                                * [N][256 points] in LDS; beyond that fewer points per workgroup, several threads per point */
/* Copy one (NB,C,H,W) field of the sampler's state after the most recent dyf_sample call: what sample_loop returns
 * beside the intermediates (dyffusion.py:424-426): (x0_hat, ., x_s), or (x_s, ., x_interpolated_s_next) when the sampling
 * schedule stops before T-1. */
typedef enum dyf_sampler_state { DYF_STATE_X0_HAT = 0, DYF_STATE_X_S = 1, DYF_STATE_X_NEXT = 2 } dyf_sampler_state;
dyf_status dyf_get_sampler_state(dyf_engine* engine, int32_t what, float* out_dev, int32_t nb, void* stream);

/* ---- training step: DYffusion.p_losses in training mode (dyffusion.py:496-567; arch unet_simple) ---------------------------- */
/* What the reference gets from torch.autograd over unet_simple.py.  A forward is RECORDED in one of four tape slots (the
 * objective runs up to two interpolator and two forecaster forwards); dyf_train_backward consumes a slot: gradient of a
 * scalar loss w.r.t. the network's parameters (accumulated into the engine's gradient buffers when param_grads != 0) and,
 * when dinputs_dev != NULL, w.r.t. its `inputs` (NB,in_channels,H,W) -- the frozen interpolator is differentiated through.
 * flags: DYF_TRAIN_BATCH_STATS = BatchNorm with batch statistics (+ running-statistics update, momentum 0.1; module.train()),
 * otherwise running statistics (frozen / eval network); DYF_TRAIN_DROPOUT = Dropout layers active (engine generator, same
 * streams as sampling; the backward re-derives the masks).  All arithmetic fp32. */
#define DYF_TRAIN_BATCH_STATS 1
#define DYF_TRAIN_DROPOUT 2
dyf_status dyf_train_forward(dyf_engine* engine, int32_t net, int32_t slot, const float* inputs_dev, const float* time_dev,
                             const float* condition_dev, float* out_dev, int32_t nb, int32_t flags, void* stream);
dyf_status dyf_train_backward(dyf_engine* engine, int32_t slot, const float* dout_dev, float* dinputs_dev, int32_t param_grads,
                              void* stream);
dyf_status dyf_train_zero_grads(dyf_engine* engine, int32_t net);
/* Operand precision of the training convolutions (ABI 8) -- the reference's `trainer.precision` (Lightning; src/configs/trainer/default.yaml:14
 * "precision: 32   # 32 or 16"): 32 = fp32 operands on the fp32 matrix cores (the default; gradients at the 1e-6 level of autograd over the oracle),
 * 16 = "bf16-mixed": activations, gradients, master weights and accumulation stay fp32, the conv operands are rounded to bf16 -- in
 * BOTH builds of the library, whatever dyf_engine_config.dtype is -- while they are staged (csrc/train_halo16.hip,
 * csrc/train_gemm.hip).  Lightning's precision=16 is fp16 + a GradScaler; this engine answers the same request with bf16 operands,
 * which need no loss scale (with mean-reduced losses dL/dout is ~1e-6..1e-7 at real batch sizes, below fp16's normal range).
 * 0 = not set (fp32 operands; the tests' kernel-form switch DYF_TRAIN_OPERANDS of dyffusion_hip_testing.h can select 16-bit operands
 * for such engines).  Applies to the dyf_train_forward / dyf_train_backward calls that follow; a recorded forward and its backward
 * should run under the same setting. */
dyf_status dyf_train_set_precision(dyf_engine* engine, int32_t bits);
int32_t dyf_train_precision(const dyf_engine* engine);
/* Copy gradients out by the reference's state_dict names (PyTorch layouts), HOST fp32 buffers; "*.running_mean/var" return the
 * updated BatchNorm statistics. */
/* Refresh only the TRAINING copy of a network's parameters (after optimizer.step()): same arguments as dyf_load_weights, which
 * must have loaded the network once.  The sampling copy (folded / packed 16-bit weights, FiLM tables) is NOT updated: call
 * dyf_load_weights again before sampling the network. */
dyf_status dyf_train_load_weights(dyf_engine* engine, int32_t net, int32_t n_tensors, const char* const* names,
                                  const float* const* data, const int64_t* const* shapes, const int32_t* ndims);
/* dyf_train_load_weights / dyf_train_export with DEVICE pointers (contiguous fp32 tensors on the engine's GPU -- the parameters /
 * gradients of a module that lives on the GPU): device-to-device copies and repack kernels, no host round trip. */
dyf_status dyf_train_load_weights_dev(dyf_engine* engine, int32_t net, int32_t n_tensors, const char* const* names,
                                      const float* const* data_dev, const int64_t* const* shapes, const int32_t* ndims);
dyf_status dyf_train_export_dev(dyf_engine* engine, int32_t net, int32_t n_tensors, const char* const* names, float* const* out_dev);
dyf_status dyf_train_export(dyf_engine* engine, int32_t net, int32_t n_tensors, const char* const* names, float* const* out_host);
/* d(scale * mean criterion)/d pred, kinds as dyf_criterion (the loss terms of p_losses, dyffusion.py:531,557). */
dyf_status dyf_criterion_grad(dyf_engine* engine, const float* pred_dev, const float* target_dev, int64_t count, int32_t kind,
                              float scale, float* dpred_dev, void* stream);

/* ---- boundary conditions of the physical-systems benchmark -------------------------------------------------------------- */
/* Replaces PhysicalSystemsBenchmarkDataModule.boundary_conditions (src/datamodules/physical_systems_benchmark.py:245-297:
 * a Python loop over batch elements with boolean-mask writes), which _evaluation_step applies to every predicted field
 * before it is returned / fed back (src/experiment_types/forecasting_multi_horizon.py:175-182), by ONE masked write over a
 * (n_fields, rows, C, H, W) fp32 stack, in place.  row_meta_dev[row] = batch element whose metadata applies to the row, -1 =
 * untouched (the host resolves the reference's indexing of ensemble stacks).  navier-stokes: preds[fixed_mask] = 0, then
 * channel 0 of grid row 0 = in_velocity*4*y*(0.41-y)/0.41^2*(1-exp(-5t)); spring-mesh: preds = where(fixed_mask, boundary). */
typedef enum dyf_bc_kind { DYF_BC_NAVIER_STOKES = 0, DYF_BC_SPRING_MESH = 1 } dyf_bc_kind;
typedef struct dyf_bc_args {
    int32_t kind;                  /* dyf_bc_kind */
    int32_t n_fields, rows, channels, height, width;
    int32_t n_meta;                /* batch elements the metadata tensors hold */
    const int32_t* row_meta_dev;   /* [rows] */
    const float* time_factor_dev;  /* navier-stokes: 1 - exp(-5 t) of every field (t = its physical time; evaluated by the host
                                    * in double like the reference's math.exp), [n_fields] or [n_fields][n_meta] */
    int32_t times_per_meta;        /* 0 / 1 */
    const uint8_t* fixed_mask_dev; /* [n_meta][C][H][W], non-zero = fixed */
    const float* in_velocity_dev;  /* navier-stokes [n_meta] */
    const float* vertex_y_dev;     /* navier-stokes [n_meta][W]: metadata["vertices"][:, 1, 0, :] */
    const float* boundary_dev;     /* spring-mesh [n_meta][C][H][W]: cat[zeros (p), features[:, 0, 2:] (q)] */
} dyf_bc_args;
dyf_status dyf_apply_boundary_conditions(dyf_engine* engine, const dyf_bc_args* args, float* preds_dev, void* stream);

/* ---- introspection (bench.py FLOP accounting); timing / op-level test seams live in dyffusion_hip_testing.h ---------- */
/* Number of network forwards one dyf_sample call performs under the current plan. */
dyf_status dyf_plan_forward_counts(const dyf_engine* engine, int32_t* n_forecaster, int32_t* n_interpolator);
/* 2*MAC of conv/linear layers of one forward of `net` for one sample (elementwise work excluded). */
dyf_status dyf_net_flops(const dyf_engine* engine, int32_t net, double* flops_per_sample);
/* The same count for the contractions the engine actually EXECUTES after dyf_load_weights: arch unet_simple skips the output
 * columns of the last decoder block that the final resample never reads, contracts init_conv into the first encoder conv, and
 * evaluates the readout only at the 4 neighbours the final bilinear resample reads (<= dyf_net_flops; equal for other archs). */
dyf_status dyf_net_flops_executed(const dyf_engine* engine, int32_t net, double* flops_per_sample);
/* Mean of the training criterion over `count` fp32 elements (src/utilities/utils.py:201-212 `get_loss`, reduction "mean"):
 * kind 0 = L1, 1 = MSE, 2 = smooth-L1 (beta 1).  The reduction the forecaster objective `DYffusion.p_losses`
 * (dyffusion.py:531,557) applies to (prediction, target); out_host[0] receives the scalar. */
dyf_status dyf_criterion(dyf_engine* engine, const float* pred_dev, const float* target_dev, int64_t count, int32_t kind,
                         double* out_host, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DYFFUSION_HIP_H */
