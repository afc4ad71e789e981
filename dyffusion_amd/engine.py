"""Thin object wrapper over the C ABI (one HipEngine = one dyf_engine).  PyTorch is only used for device memory
and streams here: every tensor crosses the boundary as a raw device pointer (`tensor.data_ptr()`)."""
import ctypes as C
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import _lib as L


class EngineError(RuntimeError):
    pass


def _raise(status: int, msg: str):
    if status == L.DYF_ERR_INVALID_ARGUMENT:
        raise ValueError(msg)
    if status == L.DYF_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise EngineError(msg)


def net_config(*, in_channels: int, cond_channels: int, out_channels: int, dim: int, with_time_emb: bool = True,
               upsample_dims: Optional[Sequence[int]] = (256, 256), dropout: float = 0.0,
               input_dropout: float = 0.0, outer_sample_mode: str = "bilinear") -> L.NetConfig:
    uh, uw = (0, 0) if upsample_dims is None else (int(upsample_dims[0]), int(upsample_dims[1]))
    cfg = L.NetConfig()
    cfg.arch, cfg.in_channels, cfg.cond_channels, cfg.out_channels = L.ARCH_UNET_SIMPLE, in_channels, cond_channels, out_channels
    cfg.dim, cfg.with_time_emb, cfg.upsample_h, cfg.upsample_w = dim, int(bool(with_time_emb)), uh, uw
    cfg.dropout, cfg.input_dropout = float(dropout), float(input_dropout)
    if outer_sample_mode not in ("bilinear", "nearest"):
        raise NotImplementedError(f"outer_sample_mode={outer_sample_mode!r}: the HIP engine implements 'bilinear' and 'nearest'")
    cfg.outer_nearest = int(outer_sample_mode == "nearest")
    return cfg


def resnet_net_config(*, in_channels: int, cond_channels: int, out_channels: int, dim: int, dim_mults=(1, 2, 4),
                      with_time_emb: bool = True, block_dropout: float = 0.0, block_dropout1: float = 0.0,
                      attn_dropout: float = 0.0, input_dropout: float = 0.0, groups: int = 8, init_kernel_size: int = 7,
                      init_padding: int = 3, keep_spatial_dims: bool = False, double_conv_layer: bool = True,
                      learned_sinusoidal_dim: int = 0) -> L.NetConfig:
    """dyf_net_config for src.models.unet.Unet (no outer resampling)."""
    cfg = L.NetConfig()
    cfg.arch, cfg.in_channels, cfg.cond_channels, cfg.out_channels = L.ARCH_UNET_RESNET, in_channels, cond_channels, out_channels
    cfg.dim, cfg.with_time_emb = dim, int(bool(with_time_emb))
    cfg.dropout, cfg.input_dropout = float(block_dropout), float(input_dropout)
    cfg.n_mults = len(dim_mults)
    for i, m in enumerate(dim_mults):
        cfg.dim_mults[i] = int(m)
    cfg.block_dropout1, cfg.attn_dropout, cfg.groups = float(block_dropout1), float(attn_dropout), int(groups)
    cfg.init_kernel_size, cfg.init_padding = int(init_kernel_size), int(init_padding)
    cfg.keep_spatial_dims, cfg.single_conv_layer = int(bool(keep_spatial_dims)), int(not double_conv_layer)
    cfg.learned_sinusoidal_dim = int(learned_sinusoidal_dim)
    return cfg


def default_dtype_for(net) -> str:
    """16-bit storage / MFMA operand format an engine is built with when the caller names none: "bf16" for the unet_simple /
    SimpleConvNet backbones (BASELINE configs[1]: "bf16"), "fp16" for the ResNet-UNet `unet.Unet` (OISST, 512^2: ~60 16-bit
    roundings per forward and 93 chained forwards per rollout -- bf16's 8 mantissa bits give 4e-2 .. 5e-2 per field over the
    OISST rollout, fp16's 11 bits 6e-3 at the same MFMA rate and the same bytes).  A network class states
    its default as `default_engine_dtype`; `net.engine_dtype = "bf16" | "fp16"` or `DYffusion(dtype=...)` override it."""
    return getattr(net, "engine_dtype", None) or getattr(net, "default_engine_dtype", "bf16")


def _f32c(t: torch.Tensor, name: str) -> torch.Tensor:
    if not t.is_cuda:
        raise ValueError(f"{name} must live on the GPU (got {t.device}); the HIP engine has no CPU path")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def parse_train_precision(precision) -> int:
    """Lightning-style precision value -> bits for dyf_train_set_precision (0 = not set)."""
    if precision is None:
        return 0
    key = str(precision).lower()
    if key in ("32", "32-true", "fp32", "float32"):
        return 32
    if key in ("16", "16-mixed", "bf16", "bf16-mixed", "fp16", "fp16-mixed"):
        # every 16-bit request runs as bf16-mixed (csrc/train_internal.h): the engine has no GradScaler, and fp16 gradient operands
        # without one lose the gradient at real batch sizes (dL/dout ~ 1 / numel).  An explicit fp16 request is told so once.
        if key in ("fp16", "fp16-mixed"):
            import warnings
            warnings.warn("train precision 'fp16-mixed' runs as bf16-mixed: the engine rounds its training conv operands to bf16 "
                          "(no loss scaling needed); fp16 operands are not offered", stacklevel=2)
        return 16
    raise ValueError(f"train precision {precision!r}: expected 32 / '32-true' or 16 / '16-mixed' / 'bf16-mixed'")


class HipEngine:
    def __init__(self, forecaster: L.NetConfig, interpolator: L.NetConfig, height: int, width: int, max_batch: int,
                 device: Optional[int] = None, use_graph: bool = True, enable_mfma: bool = True, dtype: str = "bf16",
                 batch_invariant: bool = False, row_groups: Optional[int] = None, train_precision=None):
        if not torch.cuda.is_available():
            raise EngineError("no GPU visible: the DYffusion HIP engine needs an MI355X (gfx950); there is no CPU fallback")
        self.dtype = dtype
        self._lib = L.lib(dtype)
        self.device = torch.cuda.current_device() if device is None else int(device)
        self.height, self.width, self.max_batch = int(height), int(width), int(max_batch)
        cfg = L.EngineConfig(L.DYF_ABI_VERSION, self.device, self.height, self.width, self.max_batch, int(use_graph),
                             int(enable_mfma), L.DTYPES[dtype], int(batch_invariant),
                             (L.NetConfig * 2)(forecaster, interpolator))
        self.cfg = cfg
        h = C.c_void_p()
        st = self._lib.dyf_engine_create(C.byref(cfg), C.byref(h))
        if st != L.DYF_OK:
            _raise(st, self._lib.dyf_last_error(None).decode())
        self._h = h
        if row_groups is not None:  # None: the engine's own default (dyf_engine_create, DYF_ROW_GROUPS)
            self._check(self._lib.dyf_set_row_groups(self._h, int(row_groups)))
        if train_precision is not None:
            self.train_set_precision(train_precision)
        self._plan_keepalive = None
        self.n_out_slots = 0
        self._tape_net = {}       # tape slot -> network of the recorded training forward
        self._tape_nb = {}        # tape slot -> batch rows of the recorded forward (train_backward validates dout against it)
        self.comm_rank, self.comm_world = 0, 1
        self.synchronize_errors = True  # poll_errors() after every call waits for the call's stream (ResNet-UNet engines only)
        self.train_step_id = 0    # bumped by every p_losses / get_loss training forward: a loss of an older step cannot run backward
        self.plan_valid = False   # cleared by load_weights: the plan's FiLM tables are functions of the weights
        self.weights_version = 0

    # ------------------------------------------------------------------ plumbing
    def _check(self, st: int):
        if st != L.DYF_OK:
            _raise(st, self._lib.dyf_last_error(self._h).decode())

    def poll_errors(self, synchronize: Optional[bool] = None):
        """dyf_poll_errors: asynchronous failures of the work just submitted (a fused GroupNorm convolution whose wait for its
        sample's statistics timed out; csrc/gn_fused.h) fail THIS call instead of the next one.  Engines without a live fused form
        (unet_simple, SimpleConvNet, after a downgrade) return at once; ResNet-UNet engines first wait for the STREAM the call was
        enqueued on (never for the device: other streams keep running; a stream under capture is not waited for).  A caller that
        wants net_forward / sample / sample_gather to stay asynchronous sets `engine.synchronize_errors = False`: a failure is then
        reported by the next entry point instead."""
        if synchronize is None:
            synchronize = self.synchronize_errors
        self._check(self._lib.dyf_poll_errors(self._h, int(synchronize)))

    def gn_fuse_state(self):
        """(live, downgrades): the fused GroupNorm form is in use / how often this engine left it (dyf_gn_fuse_state)."""
        live, down = C.c_int32(0), C.c_int32(0)
        self._check(self._lib.dyf_gn_fuse_state(self._h, C.byref(live), C.byref(down)))
        return bool(live.value), down.value

    def debug_gn_fuse(self, timeout_ticks: int = 0, force_timeout: bool = False):
        """Test hook (dyf_debug_gn_fuse): sweep bound in 100 MHz ticks, forced time-out of every granule sweep."""
        self._check(self._lib.dyf_debug_gn_fuse(self._h, int(timeout_ticks), int(force_timeout)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.dyf_engine_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _stream() -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream().cuda_stream)

    # ------------------------------------------------------------------ weights
    @staticmethod
    def _marshal_state_dict(state_dict: Dict[str, torch.Tensor]):
        items = [(k, v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")]
        arrs = [np.ascontiguousarray(v.detach().to("cpu", torch.float32).numpy()) for _, v in items]
        n = len(items)
        names = (C.c_char_p * n)(*[k.encode() for k, _ in items])
        data = (C.c_void_p * n)(*[a.ctypes.data for a in arrs])
        shp_store = [(C.c_int64 * max(1, a.ndim))(*a.shape) for a in arrs]
        shapes = (C.c_void_p * n)(*[C.addressof(s) for s in shp_store])
        ndims = (C.c_int32 * n)(*[a.ndim for a in arrs])
        return n, names, data, shapes, ndims, (arrs, shp_store)  # the last element keeps the buffers alive

    def load_weights(self, net: int, state_dict: Dict[str, torch.Tensor]):
        """state_dict: the reference's `UNet.state_dict()` (same key names)."""
        n, names, data, shapes, ndims, _keep = self._marshal_state_dict(state_dict)
        self.plan_valid = False
        self.weights_version += 1
        self._check(self._lib.dyf_load_weights(self._h, net, n, names, data, shapes, ndims))

    def train_load_weights(self, net: int, state_dict: Dict[str, torch.Tensor]):
        """Refresh only the training copy of `net`'s parameters (after optimizer.step()); the sampling copy keeps the weights of
        the last `load_weights` until that is called again.  Tensors that all live on this engine's GPU are read in place
        (dyf_train_load_weights_dev)."""
        items = [(k, v) for k, v in state_dict.items() if not k.endswith("num_batches_tracked")]
        if items and all(v.is_cuda and v.device.index == self.device for _, v in items):
            ts = [v.detach().to(torch.float32).contiguous() for _, v in items]
            n = len(items)
            names = (C.c_char_p * n)(*[k.encode() for k, _ in items])
            data = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
            shp_store = [(C.c_int64 * max(1, t.dim()))(*t.shape) for t in ts]
            shapes = (C.c_void_p * n)(*[C.addressof(s_) for s_ in shp_store])
            ndims = (C.c_int32 * n)(*[t.dim() for t in ts])
            torch.cuda.current_stream().synchronize()  # the optimizer's kernels have written these tensors
            self._check(self._lib.dyf_train_load_weights_dev(self._h, net, n, names, data, shapes, ndims))
            return
        n, names, data, shapes, ndims, _keep = self._marshal_state_dict(state_dict)
        self._check(self._lib.dyf_train_load_weights(self._h, net, n, names, data, shapes, ndims))

    # ------------------------------------------------------------------ per-network seam
    def net_forward(self, net: int, inputs: torch.Tensor, time: Optional[torch.Tensor] = None,
                    condition: Optional[torch.Tensor] = None, dropout_mode: int = 0,
                    masks: Optional[Sequence[torch.Tensor]] = None) -> torch.Tensor:
        inputs = _f32c(inputs, "inputs")
        nb = inputs.shape[0]
        ncfg = self.cfg.net[net]
        if inputs.dim() != 4 or inputs.shape[1] != ncfg.in_channels or tuple(inputs.shape[2:]) != (self.height, self.width):
            raise ValueError(f"inputs must be (NB, {ncfg.in_channels}, {self.height}, {self.width}), got {tuple(inputs.shape)}")
        if condition is not None:
            condition = _f32c(condition, "condition")
            if tuple(condition.shape) != (nb, ncfg.cond_channels, self.height, self.width):
                raise ValueError(f"condition has shape {tuple(condition.shape)}")
        if time is not None:
            time = _f32c(time.reshape(-1), "time")
            if time.numel() != nb:
                raise ValueError("time must have one entry per batch row")
        out = torch.empty((nb, ncfg.out_channels, self.height, self.width), dtype=torch.float32, device=inputs.device)
        mptr = None
        if masks is not None:
            keep = [m.contiguous() for m in masks]
            mptr = (C.c_void_p * len(keep))(*[m.data_ptr() for m in keep])
        self._check(self._lib.dyf_net_forward(
            self._h, net, inputs.data_ptr(), None if time is None else time.data_ptr(),
            None if condition is None else condition.data_ptr(), out.data_ptr(), nb, dropout_mode, mptr, self._stream()))
        self.poll_errors()
        return out

    # ------------------------------------------------------------------ sampler seam
    def set_plan(self, steps: List[dict], *, sampling_cold: bool, cold_for_last_step: bool, forward_conditioning: str,
                 refine: Sequence[tuple], n_out_slots: int, interpolator_dropout: bool, forecaster_dropout: bool):
        n = len(steps)
        arr = (L.PlanStep * n)()
        for i, s in enumerate(steps):
            arr[i] = L.PlanStep(float(s["forecaster_time"]), float(s["tau"]),
                                -1.0 if s["i_next"] is None else float(s["i_next"]),
                                -1.0 if s["i_cur"] is None else float(s["i_cur"]), int(s["is_last"]),
                                -1 if s["out_slot"] is None else int(s["out_slot"]))
        nr = len(refine)
        rt = (C.c_float * max(1, nr))(*[float(t) for t, _ in refine])
        rs = (C.c_int32 * max(1, nr))(*[int(sl) for _, sl in refine])
        plan = L.Plan(n, arr, int(sampling_cold), int(cold_for_last_step), L.FCOND[forward_conditioning], nr, rt, rs,
                      int(n_out_slots), int(interpolator_dropout), int(forecaster_dropout))
        self._plan_keepalive = (arr, rt, rs, plan)
        self._check(self._lib.dyf_set_plan(self._h, C.byref(plan)))
        self.n_out_slots = int(n_out_slots)
        self.plan_valid = True

    def sample(self, initial: torch.Tensor, static: Optional[torch.Tensor] = None,
               masks: Optional[Sequence[torch.Tensor]] = None, noise: Optional[torch.Tensor] = None,
               out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Returns the forecast stack (n_out_slots, NB, C, H, W); slot i = t{i+1}_preds."""
        initial = _f32c(initial, "initial_condition")
        if initial.dim() != 4:
            raise AssertionError(f"condition.shape: {tuple(initial.shape)} (should be 4D)")
        nb = initial.shape[0]
        # the C ABI copies nb * channels * H * W floats from raw pointers: validate the shapes here
        icfg = self.cfg.net[L.NET_INTERPOLATOR]
        c_out = self.cfg.net[L.NET_FORECASTER].out_channels
        if tuple(initial.shape[2:]) != (self.height, self.width) or initial.shape[1] != icfg.in_channels - c_out:
            raise ValueError(f"initial_condition must be (NB, {icfg.in_channels - c_out}, {self.height}, {self.width}), "
                             f"got {tuple(initial.shape)}")
        if (static is None) != (icfg.cond_channels == 0):
            raise ValueError("static_condition must be given iff the networks take conditional channels")
        if static is not None:
            static = _f32c(static, "static_condition")
            if tuple(static.shape) != (nb, icfg.cond_channels, self.height, self.width):
                raise ValueError(f"static_condition must be ({nb}, {icfg.cond_channels}, {self.height}, {self.width}), "
                                 f"got {tuple(static.shape)}")
        if out is None:
            out = torch.empty((self.n_out_slots, nb, c_out, self.height, self.width), dtype=torch.float32,
                              device=initial.device)
        mptr = None
        if masks is not None:
            keep = [m.contiguous() for m in masks]
            mptr = (C.c_void_p * len(keep))(*[m.data_ptr() for m in keep])
        if noise is not None:
            noise = _f32c(noise, "noise")
        if tuple(out.shape) != (self.n_out_slots, nb, c_out, self.height, self.width) or not out.is_contiguous() \
                or out.dtype != torch.float32:
            raise ValueError("out must be a contiguous fp32 (n_out_slots, NB, C, H, W) tensor")
        self._check(self._lib.dyf_sample(self._h, initial.data_ptr(), None if static is None else static.data_ptr(),
                                         out.data_ptr(), nb, mptr, None if noise is None else noise.data_ptr(),
                                         self._stream()))
        self.poll_errors()
        return out

    # ------------------------------------------------------------------ engine-owned exchange (ensemble sharding over GPUs)
    @staticmethod
    def comm_unique_id(dtype: str = "bf16") -> bytes:
        """128-byte RCCL unique id (dyf_comm_unique_id): created on ONE rank, handed to every rank's `comm_init`."""
        lib = L.lib(dtype)
        buf = (C.c_uint8 * L.COMM_ID_BYTES)()
        st = lib.dyf_comm_unique_id(buf)
        if st != L.DYF_OK:
            _raise(st, lib.dyf_last_error(None).decode())
        return bytes(buf)

    def comm_init(self, unique_id: bytes, rank: int, world: int):
        """This engine's RCCL communicator over `world` ranks (dyf_comm_init); needed by `sample_gather`."""
        if len(unique_id) != L.COMM_ID_BYTES:
            raise ValueError(f"unique_id must be {L.COMM_ID_BYTES} bytes")
        buf = (C.c_uint8 * L.COMM_ID_BYTES).from_buffer_copy(unique_id)
        self._check(self._lib.dyf_comm_init(self._h, buf, int(rank), int(world)))
        self.comm_rank, self.comm_world = int(rank), int(world)

    def comm_destroy(self):
        self._check(self._lib.dyf_comm_destroy(self._h))
        self.comm_world = 1

    def comm_count(self) -> int:
        """Ranks of this engine's communicator as RCCL reports them (ncclCommCount); 0 when it owns none."""
        n = C.c_int32(0)
        self._check(self._lib.dyf_comm_count(self._h, C.byref(n)))
        return n.value

    def sample_gather(self, initial: torch.Tensor, static: Optional[torch.Tensor], total_rows: int) -> torch.Tensor:
        """dyf_sample_gather: rollout of this rank's rows, ONE all-gather of the forecast stack, unpack -> (n_out_slots, total_rows,
        C, H, W) with every rank's rows in global order."""
        initial = _f32c(initial, "initial_condition")
        nb = initial.shape[0]
        icfg = self.cfg.net[L.NET_INTERPOLATOR]
        c_out = self.cfg.net[L.NET_FORECASTER].out_channels
        if initial.dim() != 4 or tuple(initial.shape[2:]) != (self.height, self.width) or initial.shape[1] != icfg.in_channels - c_out:
            raise ValueError(f"initial_condition must be (NB, {icfg.in_channels - c_out}, {self.height}, {self.width}), got {tuple(initial.shape)}")
        if (static is None) != (icfg.cond_channels == 0):
            raise ValueError("static_condition must be given iff the networks take conditional channels")
        if static is not None:
            static = _f32c(static, "static_condition")
            if tuple(static.shape) != (nb, icfg.cond_channels, self.height, self.width):
                raise ValueError(f"static_condition has shape {tuple(static.shape)}")
        out = torch.empty((self.n_out_slots, int(total_rows), c_out, self.height, self.width), dtype=torch.float32, device=initial.device)
        self._check(self._lib.dyf_sample_gather(self._h, initial.data_ptr(), None if static is None else static.data_ptr(),
                                                out.data_ptr(), nb, int(total_rows), self._stream()))
        self.poll_errors()
        return out

    def sampler_state(self, what: int, nb: int) -> torch.Tensor:
        """what: 0 = last x0_hat, 1 = x_s, 2 = x_interpolated_s_next of the final step (dyf_sampler_state)."""
        c_out = self.cfg.net[L.NET_FORECASTER].out_channels
        out = torch.empty((nb, c_out, self.height, self.width), dtype=torch.float32, device=f"cuda:{self.device}")
        self._check(self._lib.dyf_get_sampler_state(self._h, what, out.data_ptr(), nb, self._stream()))
        return out

    def last_x0hat(self, nb: int) -> torch.Tensor:
        return self.sampler_state(0, nb)

    def seed(self, seed: int):
        """Re-seed the dropout / noise generator; forward and noise counters restart at 0."""
        self._check(self._lib.dyf_seed(self._h, C.c_uint64(int(seed) & (2 ** 64 - 1))))

    def set_log_intermediates(self, enable: bool):
        """log_every_t: keep x0_hat / x_interpolated_s_next / x_interpolated_s of every sampling step of the next `sample` calls."""
        self._check(self._lib.dyf_set_log_intermediates(self._h, int(bool(enable))))

    def get_log(self, step: int, what: int, nb: int) -> torch.Tensor:
        c_out = self.cfg.net[L.NET_FORECASTER].out_channels
        out = torch.empty((nb, c_out, self.height, self.width), dtype=torch.float32, device=f"cuda:{self.device}")
        self._check(self._lib.dyf_get_log(self._h, int(step), int(what), out.data_ptr(), nb, self._stream()))
        return out

    @property
    def row_groups(self) -> int:
        """Number of concurrent row groups a large enough sampling call is split over (dyf_set_row_groups; 1 = none)."""
        return int(self._lib.dyf_row_groups(self._h))

    def set_row_offset(self, first_row: int):
        """Global index of this engine's batch row 0 (ensemble sharding): rows draw the masks of the un-sharded batch."""
        self._check(self._lib.dyf_set_row_offset(self._h, C.c_uint32(int(first_row))))

    def read_block_output(self, net: int, layer: int, nb: int) -> torch.Tensor:
        """Test seam: output of UNetBlock `layer` (0..11) of the most recent unet_simple forward, fp32 NCHW."""
        d = self.cfg.net[net].dim
        ch = [2 * d, 2 * d, 4 * d, 8 * d, 8 * d, 8 * d, 8 * d, 8 * d, 4 * d, 2 * d, 2 * d, d][layer]
        uh = self.cfg.net[net].upsample_h or self.height
        uw = self.cfg.net[net].upsample_w or self.width
        sh = layer + 1 if layer < 6 else 11 - layer   # log2 of the block output's down-sampling factor
        shape = (nb, ch, uh >> sh, uw >> sh)
        out = torch.empty(shape, dtype=torch.float32, device=f"cuda:{self.device}")
        self._check(self._lib.dyf_debug_read_block_output(self._h, net, layer, nb, out.data_ptr(), self._stream()))
        return out

    def form_log(self, enable: bool) -> None:
        """Test seam: clear and enable (or disable) the process-wide kernel-form log (dyf_debug_form_log)."""
        self._lib.dyf_debug_form_log(int(bool(enable)))

    def form_log_read(self) -> Dict[str, Dict[int, int]]:
        """{kernel form: {batch rows of the launch: launches noted}} since the log was enabled."""
        n = self._lib.dyf_debug_form_log_read(None, 0)
        buf = C.create_string_buffer(n + 1)
        self._lib.dyf_debug_form_log_read(buf, n + 1)
        out: Dict[str, Dict[int, int]] = {}
        for item in buf.value.decode().split(";"):
            if item:
                key, cnt = item.rsplit("=", 1)
                form, rows = key.rsplit("@", 1)
                out.setdefault(form, {})[int(rows)] = int(cnt)
        return out

    # ------------------------------------------------------------------ introspection
    def forward_counts(self):
        a, b = C.c_int32(), C.c_int32()
        self._check(self._lib.dyf_plan_forward_counts(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def net_flops(self, net: int) -> float:
        f = C.c_double()
        self._check(self._lib.dyf_net_flops(self._h, net, C.byref(f)))
        return f.value

    def net_flops_executed(self, net: int) -> float:
        f = C.c_double()
        self._check(self._lib.dyf_net_flops_executed(self._h, net, C.byref(f)))
        return f.value

    def time_conv_layer(self, net: int, layer: int, nb: int, iters: int = 20):
        ms, fl, by = C.c_double(), C.c_double(), C.c_double()
        self._check(self._lib.dyf_time_conv_layer(self._h, net, layer, nb, iters, self._stream(), C.byref(ms),
                                                  C.byref(fl), C.byref(by)))
        return ms.value, fl.value, by.value

    def ensemble_metrics(self, preds: torch.Tensor, targets: torch.Tensor):
        """preds (N, B, ...) fp32 on the GPU, targets (B, ...) -> (mse, ssr, crps) floats (dyf_ensemble_metrics)."""
        n = preds.shape[0]
        preds = preds.contiguous().float()
        targets = targets.contiguous().float()
        if preds.shape[1:] != targets.shape:
            raise ValueError(f"predictions.shape[1:] ({tuple(preds.shape[1:])}) != targets.shape ({tuple(targets.shape)})")
        out = (C.c_double * 3)()
        self._check(self._lib.dyf_ensemble_metrics(self._h, preds.data_ptr(), targets.data_ptr(), n, targets.numel(), out,
                                                   self._stream()))
        return out[0], out[1], out[2]

    def time_layer_in_rollout(self, layer: int, nb: int):
        """(average ms, launches) of decoder block `layer`'s conv over one eagerly launched rollout of the current plan."""
        ms, cnt = C.c_double(), C.c_int32()
        self._check(self._lib.dyf_time_layer_in_rollout(self._h, layer, nb, self._stream(), C.byref(ms), C.byref(cnt)))
        return ms.value, cnt.value

    def time_kernel_in_rollout(self, kind: int, nb: int):
        """ResNet-UNet pair: (average ms, launches, flops per launch, algorithmic bytes per launch) of kernel class `kind` (0 level-0
        3x3 convs, 1 bottleneck attention core, 2 level-0 GroupNorm chain) over one eagerly launched rollout of the current plan."""
        ms, cnt, fl, by = C.c_double(), C.c_int32(), C.c_double(), C.c_double()
        self._check(self._lib.dyf_time_kernel_in_rollout(self._h, kind, nb, self._stream(), C.byref(ms), C.byref(cnt), C.byref(fl),
                                                         C.byref(by)))
        return ms.value, cnt.value, fl.value, by.value

    def time_named_kernel_in_rollout(self, kernel: str, nb: int):
        """(total ms, launches, total algorithmic bytes) of every launch of `kernel` in one eager rollout (dyffusion_hip_testing.h)."""
        ms, by, n = C.c_double(0.0), C.c_double(0.0), C.c_int32(0)
        self._check(self._lib.dyf_time_named_kernel_in_rollout(self._h, kernel.encode(), nb, self._stream(), C.byref(ms), C.byref(n),
                                                               C.byref(by)))
        return ms.value, n.value, by.value

    def op_conv2d(self, x_nhwc_bf16: torch.Tensor, weight: torch.Tensor, stride: int, pad: int,
                  scale: Optional[torch.Tensor] = None, shift: Optional[torch.Tensor] = None, act: int = 0,
                  path: int = 1) -> torch.Tensor:
        """Test seam: x (N,H,W,Cin) bf16 on the GPU, weight (Cout,Cin,kh,kw) fp32 (any device) -> (N,Ho,Wo,Cout) bf16."""
        assert x_nhwc_bf16.dtype == torch.bfloat16 and x_nhwc_bf16.is_cuda and x_nhwc_bf16.is_contiguous()
        n, h, w, cin = x_nhwc_bf16.shape
        cout, _, kh, kw = weight.shape
        wh = np.ascontiguousarray(weight.detach().to("cpu", torch.float32).numpy())
        ho, wo = (h + 2 * pad - kh) // stride + 1, (w + 2 * pad - kw) // stride + 1
        y = torch.empty((n, ho, wo, cout), dtype=torch.bfloat16, device=x_nhwc_bf16.device)
        if scale is not None:
            scale, shift = _f32c(scale, "scale"), _f32c(shift, "shift")
        self._check(self._lib.dyf_op_conv2d(self._h, x_nhwc_bf16.data_ptr(), wh.ctypes.data, n, h, w, cin, cout, kh, kw,
                                            stride, pad, None if scale is None else scale.data_ptr(),
                                            None if shift is None else shift.data_ptr(), act, path, y.data_ptr(),
                                            self._stream()))
        return y

    def criterion(self, pred: torch.Tensor, target: torch.Tensor, kind: str = "l1") -> float:
        """Mean L1 / MSE / smooth-L1 between two fp32 device tensors (get_loss, src/utilities/utils.py:201-212)."""
        name = kind.lower().strip().replace("-", "_")
        code = 0 if name in ("l1", "mae", "mean_absolute_error") else 1 if name in ("l2", "mse", "mean_squared_error") else \
            2 if name in ("smoothl1", "smooth") else None
        if code is None:
            raise ValueError(f"Unknown loss function {kind}")
        pred, target = _f32c(pred, "pred"), _f32c(target, "target")
        # the kernel reads 16 bytes per lane: a contiguous view at an odd element offset is copied to an aligned buffer
        if pred.data_ptr() % 16:
            pred = pred.clone()
        if target.data_ptr() % 16:
            target = target.clone()
        if pred.shape != target.shape:
            raise ValueError(f"shape mismatch: {tuple(pred.shape)} vs {tuple(target.shape)}")
        out = C.c_double(0.0)
        self._check(self._lib.dyf_criterion(self._h, pred.data_ptr(), target.data_ptr(), pred.numel(), code, C.byref(out),
                                            self._stream()))
        return out.value

    # ------------------------------------------------------------------ training step (arch unet_simple, fp32)
    def train_forward(self, net: int, slot: int, inputs: torch.Tensor, time: Optional[torch.Tensor],
                      condition: Optional[torch.Tensor], batch_stats: bool, dropout: bool) -> torch.Tensor:
        """One RECORDED forward (tape `slot`, 0..3): batch-statistics BatchNorm iff `batch_stats`, Dropout iff `dropout`."""
        inputs = _f32c(inputs, "inputs")
        nb = inputs.shape[0]
        ncfg = self.cfg.net[net]
        if inputs.dim() != 4 or inputs.shape[1] != ncfg.in_channels or tuple(inputs.shape[2:]) != (self.height, self.width):
            raise ValueError(f"inputs must be (NB, {ncfg.in_channels}, {self.height}, {self.width}), got {tuple(inputs.shape)}")
        if condition is not None:
            condition = _f32c(condition, "condition")
            if tuple(condition.shape) != (nb, ncfg.cond_channels, self.height, self.width):
                raise ValueError(f"condition has shape {tuple(condition.shape)}")
        if time is not None:
            time = _f32c(time.reshape(-1), "time")
        out = torch.empty((nb, ncfg.out_channels, self.height, self.width), dtype=torch.float32, device=inputs.device)
        flags = (L.TRAIN_BATCH_STATS if batch_stats else 0) | (L.TRAIN_DROPOUT if dropout else 0)
        self._tape_net[slot] = net
        self._tape_nb[slot] = nb
        self._check(self._lib.dyf_train_forward(self._h, net, slot, inputs.data_ptr(), None if time is None else time.data_ptr(),
                                                None if condition is None else condition.data_ptr(), out.data_ptr(), nb, flags,
                                                self._stream()))
        return out

    def train_backward(self, slot: int, dout: torch.Tensor, want_dinputs: bool, param_grads: bool) -> Optional[torch.Tensor]:
        dout = _f32c(dout, "dout")
        if slot not in self._tape_net:
            raise EngineError(f"tape slot {slot} holds no recorded forward")
        net = self._tape_net[slot]
        want = (self._tape_nb[slot], self.cfg.net[net].out_channels, self.height, self.width)
        if tuple(dout.shape) != want:  # the C ABI reads / writes nb rows from raw pointers
            raise ValueError(f"dout must have the recorded forward's output shape {want}, got {tuple(dout.shape)}")
        din = None
        if want_dinputs:
            din = torch.empty((dout.shape[0], self.cfg.net[net].in_channels, self.height, self.width), dtype=torch.float32,
                              device=dout.device)
        self._check(self._lib.dyf_train_backward(self._h, slot, dout.data_ptr(), None if din is None else din.data_ptr(),
                                                 int(param_grads), self._stream()))
        return din

    def train_set_precision(self, precision) -> None:
        """Operand precision of the training convolutions (dyf_train_set_precision; the reference's Lightning `trainer.precision`):
        32 / "32" / "32-true" = fp32 operands (default), 16 / "16-mixed" / "bf16-mixed" = operands rounded to bf16 while staged (fp32
        tensors, master weights and accumulation) -- bf16 on fp16 engines too: the reference's precision=16 is fp16 + GradScaler, and
        fp16 gradient operands without a loss scale underflow at real batch sizes (`parse_train_precision`); None = not set (fp32)."""
        self._check(self._lib.dyf_train_set_precision(self._h, parse_train_precision(precision)))

    @property
    def train_precision(self) -> int:
        return int(self._lib.dyf_train_precision(self._h))

    def train_zero_grads(self, net: int):
        self._check(self._lib.dyf_train_zero_grads(self._h, net))

    def train_export(self, net: int, names_shapes: Dict[str, tuple], device: Optional[torch.device] = None) -> Dict[str, torch.Tensor]:
        """Gradients (or updated BatchNorm running statistics) by state_dict name -> fp32 tensors: on the CPU, or -- `device` =
        this engine's GPU -- written there directly (device-to-device, dyf_train_export_dev)."""
        names = list(names_shapes)
        if device is not None and device.type == "cuda":
            outs = [torch.empty(names_shapes[k], dtype=torch.float32, device=device) for k in names]
            cn = (C.c_char_p * len(names))(*[k.encode() for k in names])
            cp = (C.c_void_p * len(names))(*[o.data_ptr() for o in outs])
            torch.cuda.current_stream().synchronize()
            self._check(self._lib.dyf_train_export_dev(self._h, net, len(names), cn, cp))
            return dict(zip(names, outs))
        bufs = [np.empty(names_shapes[k], dtype=np.float32) for k in names]
        cn = (C.c_char_p * len(names))(*[k.encode() for k in names])
        cp = (C.c_void_p * len(names))(*[b.ctypes.data for b in bufs])
        self._check(self._lib.dyf_train_export(self._h, net, len(names), cn, cp))
        return {k: torch.from_numpy(b) for k, b in zip(names, bufs)}

    def criterion_grad(self, pred: torch.Tensor, target: torch.Tensor, kind: str, scale: float) -> torch.Tensor:
        name = kind.lower().strip().replace("-", "_")
        code = 0 if name in ("l1", "mae", "mean_absolute_error") else 1 if name in ("l2", "mse", "mean_squared_error") else 2
        pred, target = _f32c(pred, "pred"), _f32c(target, "target")
        d = torch.empty_like(pred)
        self._check(self._lib.dyf_criterion_grad(self._h, pred.data_ptr(), target.data_ptr(), pred.numel(), code, float(scale),
                                                 d.data_ptr(), self._stream()))
        return d

    def train_conv_check(self, kind: int, n: int, h: int, w: int, cin: int, cout: int, k: int, s: int, p: int, seed: int = 1):
        """Test seam: one training convolution (0 forward, 1 dgrad, 2 wgrad on the matrix-core launchers; 3 wgrad, 4 forward through the
        step's dispatchers) vs the plain VALU kernel.
        Returns (relative max error with split-K workspace, without, whether the matrix-core form took the shape)."""
        out = (C.c_float * 3)()
        self._check(self._lib.dyf_train_conv_check(self._h, kind, n, h, w, cin, cout, k, s, p, C.c_uint32(seed), out))
        return float(out[0]), float(out[1]), bool(out[2])

    def op_linear_attention(self, qkv_bf16: torch.Tensor) -> torch.Tensor:
        """Test seam: LinearAttention core.  qkv (N,HW,384) bf16 (to_qkv output, 4 heads x 32) -> (N,HW,128) bf16."""
        assert qkv_bf16.dtype == self.torch_dtype and qkv_bf16.is_cuda and qkv_bf16.is_contiguous()
        n, hw, c3 = qkv_bf16.shape
        assert c3 == 384
        y = torch.empty((n, hw, 128), dtype=qkv_bf16.dtype, device=qkv_bf16.device)
        self._check(self._lib.dyf_op_linear_attention(self._h, qkv_bf16.data_ptr(), n, hw, y.data_ptr(), self._stream()))
        return y

    def op_linear_attention_fused(self, xn: torch.Tensor, xres: torch.Tensor, wqkv: torch.Tensor, wout: torch.Tensor,
                                  bout: torch.Tensor) -> torch.Tensor:
        """Test seam: the fused LinearAttention block (to_qkv + core + to_out + bias + residual).  xn, xres (N,HW,C) in the engine's
        16-bit dtype on the device, C in {64, 128}; wqkv (384,C), wout (C,128), bout (C) fp32 on the host -> (N,HW,C)."""
        assert xn.dtype == self.torch_dtype and xn.is_cuda and xn.is_contiguous() and xres.shape == xn.shape and xres.is_contiguous()
        n, hw, c = xn.shape
        wq, wo, bo = (t.detach().to(torch.float32).cpu().contiguous() for t in (wqkv, wout, bout))
        assert wq.shape == (384, c) and wo.shape == (c, 128) and bo.shape == (c,)
        y = torch.empty_like(xn)
        self._check(self._lib.dyf_op_linear_attention_fused(self._h, xn.data_ptr(), xres.data_ptr(), n, hw, c, wq.data_ptr(),
                                                            wo.data_ptr(), bo.data_ptr(), y.data_ptr(), self._stream()))
        return y

    @property
    def torch_dtype(self) -> torch.dtype:
        return torch.float16 if L.DTYPES[self.dtype] else torch.bfloat16

    def op_attention(self, qkv: torch.Tensor, p_drop: float = 0.0) -> torch.Tensor:
        """Test seam: Attention core.  qkv (N,HW,384) in the engine's 16-bit dtype (to_qkv output) -> (N,HW,128); p_drop > 0:
        dropout on the probabilities from the engine's generator."""
        assert qkv.dtype == self.torch_dtype and qkv.is_cuda and qkv.is_contiguous() and qkv.shape[2] == 384
        n, hw, _ = qkv.shape
        y = torch.empty((n, hw, 128), dtype=qkv.dtype, device=qkv.device)
        self._check(self._lib.dyf_op_attention_dropout(self._h, qkv.data_ptr(), n, hw, float(p_drop), y.data_ptr(), self._stream()))
        return y

    def op_upconv2d(self, x_nhwc_bf16: torch.Tensor, weight: torch.Tensor, scale: Optional[torch.Tensor] = None,
                    shift: Optional[torch.Tensor] = None, act: int = 0) -> torch.Tensor:
        """Test seam: fused Upsample(x2, bilinear) + Conv2d(3x3, pad 1).  x (N,H,W,Cin) bf16 -> (N,2H,2W,Cout) bf16."""
        assert x_nhwc_bf16.dtype == torch.bfloat16 and x_nhwc_bf16.is_cuda and x_nhwc_bf16.is_contiguous()
        n, h, w, cin = x_nhwc_bf16.shape
        cout = weight.shape[0]
        wh = np.ascontiguousarray(weight.detach().to("cpu", torch.float32).numpy())
        y = torch.empty((n, 2 * h, 2 * w, cout), dtype=torch.bfloat16, device=x_nhwc_bf16.device)
        if scale is not None:
            scale, shift = _f32c(scale, "scale"), _f32c(shift, "shift")
        self._check(self._lib.dyf_op_upconv2d(self._h, x_nhwc_bf16.data_ptr(), wh.ctypes.data, n, h, w, cin, cout,
                                              None if scale is None else scale.data_ptr(),
                                              None if shift is None else shift.data_ptr(), act, y.data_ptr(),
                                              self._stream()))
        return y


class EngineLoss(torch.autograd.Function):
    """Scalar loss whose backward is the engine's backward pass: `owner._train_backward(upstream)` runs dyf_train_backward and
    accumulates into `param.grad` (so `loss.backward()` / torch.optim / Lightning's training_step work unchanged)."""

    @staticmethod
    def forward(ctx, anchor, owner, value):
        ctx.owner = owner
        ctx.step_id = owner._train_state["step_id"]  # the engine's tapes are single-slot: they belong to the LATEST forward
        return anchor.new_tensor(value)

    @staticmethod
    def backward(ctx, grad_out):
        st = getattr(ctx.owner, "_train_state", None)
        # the owner's own record AND the engine-wide counter: two owners can share one engine (a DYffusion and its attached
        # forecaster's get_loss both record into the same tape slots), and a training forward by EITHER overwrites the tapes
        eng = None if st is None else st.get("eng")
        if st is None or st.get("step_id") != ctx.step_id or st.get("consumed") or \
                (eng is not None and eng.train_step_id != ctx.step_id):
            raise RuntimeError("this loss belongs to an earlier training forward (or has already been back-propagated): the "
                               "engine records one step at a time -- call .backward() on a loss before the next p_losses() / "
                               "get_loss() of the same engine, and only once")
        st["consumed"] = True
        ctx.owner._train_backward(float(grad_out))
        return None, None, None


def state_version(net) -> int:
    """Identifies the state of a module's parameters and buffers: the sum of their in-place modification counters
    (`Tensor._version`: optimizer steps, `p.mul_()`, `load_state_dict`, ...) and of their storage addresses (`p.data = other`
    swaps).  NOT visible to it: in-place writes through an alias, e.g. `p.data.copy_(...)` -- call `mark_weights_modified()`
    on the module (or `load_state_dict`) after such an edit.  The tensor list is cached on the module (the mirrors drop it in
    `_apply`, i.e. on .cuda() / .to() / .float(), which replace buffer objects): ~20 us per call."""
    vt = net.__dict__.get("_version_tensors")
    if vt is None:
        vt = list(net.parameters()) + list(net.buffers())
        net.__dict__["_version_tensors"] = vt
    v = net.__dict__.get("_manual_version", 0)
    for t in vt:
        v += t._version + t.data_ptr()
    return v


def mark_weights_modified(net) -> None:
    """Force the next engine use of `net` to re-upload its weights (after edits the version counters cannot see)."""
    net.__dict__["_manual_version"] = net.__dict__.get("_manual_version", 0) + 1
    net.__dict__.pop("_version_tensors", None)


def upload_weights(net, eng: "HipEngine", slot: int) -> None:
    """dyf_load_weights of `net`'s state_dict (sampling copy AND training copy), remembering which version of the parameters
    the engine now holds."""
    eng.load_weights(slot, net.state_dict())
    net._uploaded_version = net._train_version = (id(eng), state_version(net))


def sync_weights(net, eng: "HipEngine", slot: int) -> None:
    """Before SAMPLING / inference: re-upload a network whose parameters or buffers were modified since the last full upload
    (optimizer.step(), `p.data = ...` swaps; see `state_version` for what is detected)."""
    if getattr(net, "_uploaded_version", None) != (id(eng), state_version(net)):
        upload_weights(net, eng, slot)


def sync_train_weights(net, eng: "HipEngine", slot: int) -> None:
    """Before a TRAINING step: refresh only the engine's fp32 training copy (dyf_train_load_weights, ~10x cheaper than the full
    upload, which re-derives every packed 16-bit layout of the sampling path); the sampling copy is brought up to date by
    `sync_weights` when the network is next sampled."""
    ver = (id(eng), state_version(net))
    if getattr(net, "_train_version", None) == ver:
        return
    if getattr(net, "_uploaded_version", None) is None or net._uploaded_version[0] != id(eng):
        upload_weights(net, eng, slot)  # this engine has never seen the network: full load first
    else:
        eng.train_load_weights(slot, net.state_dict())
        net._train_version = ver


def collect_train_results(net, eng: "HipEngine", slot: int, n_forwards: int) -> None:
    """After dyf_train_backward: add the engine's parameter gradients of network `slot` to `param.grad`, clear them in the
    engine, and bring the BatchNorm buffers to what module.train() leaves (running statistics after `n_forwards` momentum
    updates, num_batches_tracked)."""
    sd = net.state_dict(keep_vars=True)
    shapes = {k: tuple(v.shape) for k, v in sd.items() if isinstance(v, torch.nn.Parameter)}
    on_gpu = all(v.is_cuda and v.device.index == eng.device for v in sd.values() if torch.is_tensor(v) and v.is_floating_point())
    dev = next(iter(sd.values())).device if on_gpu else None
    for k, g in eng.train_export(slot, shapes, device=dev).items():
        p = sd[k]
        g = g.to(device=p.device, dtype=p.dtype)
        p.grad = g if p.grad is None else p.grad + g
    eng.train_zero_grads(slot)
    bufs = {k: tuple(v.shape) for k, v in sd.items() if k.endswith("running_mean") or k.endswith("running_var")}
    with torch.no_grad():
        for k, v in eng.train_export(slot, bufs, device=dev).items():
            sd[k].copy_(v)
        for k, v in sd.items():
            if k.endswith("num_batches_tracked"):
                v += n_forwards
    # the engine's TRAINING copy holds exactly these buffers; its sampling copy still has the old running statistics folded in
    net._train_version = (id(eng), state_version(net))

