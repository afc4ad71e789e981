"""Host-side mirror of `src/diffusion/dyffusion.py` (BaseDYffusion :17-431, DYffusion :439-494) on the HIP engine.

`DYffusion.sample()` keeps the reference's signature and return value (dict `t{i}_preds`), but the loop itself
(forecaster / interpolator forwards, cold-sampling update, refinement pass) runs inside libdyffusion_hip.so as one
captured hipGraph.  What stays in Python is the scalar bookkeeping that the reference also does in Python: the
diffusion-step <-> interpolation-time tables and the sampling-schedule parser; it is resolved once into a plan.
"""
import math
import os
from typing import Dict, List, Optional, Sequence, Union

import numpy as np
import torch
from torch import Tensor, nn

from . import _lib as L
from .engine import EngineLoss, HipEngine, collect_train_results, default_dtype_for, sync_train_weights, sync_weights
from .unet_simple import UNet, _AttrDict  # noqa: F401

Step = Union[int, float]
BF16_RESNET_MAX_FORWARDS = 32  # longest sampling plan (network forwards) a ResNet-UNet pair is served in bf16 storage (_ensure_plan)


class DYffusion(nn.Module):
    def __init__(self, model: UNet, interpolator, timesteps: int, forward_conditioning: str = "data",
                 schedule: str = "before_t1_only", additional_interpolation_steps: int = 0,
                 additional_interpolation_steps_factor: int = 0, interpolate_before_t1: bool = False,
                 sampling_type: str = "cold", sampling_schedule: Union[List[float], str] = None,
                 time_encoding: str = "dynamics", refine_intermediate_predictions: bool = False,
                 prediction_timesteps: Optional[Sequence[float]] = None,
                 enable_interpolator_dropout: Union[bool, str] = True, use_cold_sampling_for_last_step: bool = False,
                 log_every_t=None, lambda_reconstruction: float = 1.0, lambda_reconstruction2: float = 0.0,
                 interpolator_horizon: Optional[int] = None, interpolator_window: int = 1,
                 enable_forecaster_dropout: bool = False, max_batch: int = 64, use_graph: bool = True,
                 enable_mfma: bool = True, loss_function: str = "mean_squared_error", dtype: Optional[str] = None,
                 batch_invariant: bool = False, row_groups: Optional[int] = None, allow_bf16_long_rollout: bool = False,
                 train_precision=None, **kwargs):
        super().__init__()
        if model is None:
            raise ValueError("Arg ``model`` is missing... Please provide a backbone model for the diffusion model (e.g. a Unet)")
        if forward_conditioning not in ("data", "none", "data+noise"):
            raise ValueError(f"Invalid forward_conditioning: {forward_conditioning}")
        if enable_interpolator_dropout not in (True, False):
            raise ValueError(f"Invalid enable_interpolator_dropout: {enable_interpolator_dropout}")
        sampling_schedule = None if sampling_schedule == "None" else sampling_schedule
        self.hparams = _AttrDict(
            timesteps=timesteps, forward_conditioning=forward_conditioning, schedule=schedule,
            additional_interpolation_steps=additional_interpolation_steps,
            additional_interpolation_steps_factor=additional_interpolation_steps_factor,
            interpolate_before_t1=interpolate_before_t1, sampling_type=sampling_type,
            sampling_schedule=sampling_schedule, time_encoding=time_encoding,
            refine_intermediate_predictions=refine_intermediate_predictions, prediction_timesteps=prediction_timesteps,
            enable_interpolator_dropout=enable_interpolator_dropout,
            use_cold_sampling_for_last_step=use_cold_sampling_for_last_step, log_every_t=log_every_t,
            lambda_reconstruction=lambda_reconstruction, lambda_reconstruction2=lambda_reconstruction2,
            loss_function=loss_function)  # BaseModel default (_base_model.py:48)
        self.model = model
        # the reference stores an InterpolationExperiment here (dyffusion.py:461-468); accept that duck type or a bare net
        self.interpolator = interpolator
        self._ipol_net: UNet = getattr(interpolator, "model", interpolator)
        self.interpolator_window = getattr(interpolator, "window", interpolator_window)
        self.interpolator_horizon = getattr(interpolator, "true_horizon", interpolator_horizon)
        self.num_input_channels = model.num_input_channels
        self.num_output_channels = model.num_output_channels
        self.num_conditional_channels = model.num_conditional_channels
        self.spatial_shape = model.spatial_shape
        self.enable_interpolator_dropout = enable_interpolator_dropout
        self.enable_forecaster_dropout = enable_forecaster_dropout

        # ---- diffusion-step bookkeeping (dyffusion.py:40-95)
        horizon = timesteps
        assert horizon > 1, f"horizon must be > 1, but got {horizon}. Please use datamodule.horizon with > 1"
        self.di_to_ti_add = 0
        self.additional_interpolation_steps_fac = 0
        if schedule == "linear":
            assert additional_interpolation_steps == 0, "additional_interpolation_steps must be 0 when using linear schedule"
            self.additional_interpolation_steps_fac = additional_interpolation_steps_factor
            between = horizon - 1 if interpolate_before_t1 else horizon - 2
            self.di_to_ti_add = 0 if interpolate_before_t1 else additional_interpolation_steps_factor
            self.additional_diffusion_steps = additional_interpolation_steps_factor * between
        elif schedule == "before_t1_only":
            assert additional_interpolation_steps_factor == 0, \
                "additional_interpolation_steps_factor must be 0 when using before_t1_only schedule"
            assert interpolate_before_t1, "interpolate_before_t1 must be True when using before_t1_only schedule"
            self.additional_diffusion_steps = additional_interpolation_steps
        else:
            raise ValueError(f"Invalid schedule: {schedule}")
        self.num_timesteps = horizon + self.additional_diffusion_steps
        d2i = {d: self.diffusion_step_to_interpolation_step(d) for d in range(1, self.num_timesteps)}
        self.dynamical_steps = {d: i for d, i in d2i.items() if float(i).is_integer()}
        self.artificial_interpolation_steps = {d: i for d, i in d2i.items() if not float(i).is_integer()}
        self.i_to_diffusion_step = {i: d for d, i in d2i.items()}
        last_i = self.diffusion_step_to_interpolation_step(self.num_timesteps - 1)
        if self.interpolator_horizon is None:
            self.interpolator_horizon = int(last_i + 1)
        if self.interpolator_horizon != last_i + 1:  # dyffusion.py:472-478
            raise ValueError(f"interpolator horizon {self.interpolator_horizon} must be equal to the "
                             f"last interpolation step+1=i_N=i_{self.num_timesteps - 1}={last_i + 1}")
        self.full_sampling_schedule = list(range(0, self.num_timesteps))
        self.sampling_schedule = sampling_schedule or self.full_sampling_schedule

        # ---- engine: forecaster + interpolator in one dyf_engine
        # dtype: 16-bit storage / MFMA operand format of the engine: "bf16" or "fp16"; None = the forecaster class's default
        # (engine.default_dtype_for: bf16 for unet_simple -- BASELINE configs[1] --, fp16 for the ResNet-UNet -- configs[2], [4])
        dtype = dtype or default_dtype_for(model)
        self._engine_opts = dict(max_batch=max_batch, use_graph=use_graph, enable_mfma=enable_mfma, dtype=dtype,
                                 batch_invariant=batch_invariant,  # batch_invariant: bit-identical rows under any batching / sharding
                                 row_groups=row_groups,  # concurrent row groups of a sampling call (None = engine default)
                                 # operand precision of the training step's convolutions: the reference's `trainer.precision`
                                 # (32 default; 16 / "16-mixed" / "bf16-mixed": HipEngine.train_set_precision)
                                 train_precision=train_precision)
        self.allow_bf16_long_rollout = bool(allow_bf16_long_rollout)
        self._engine: Optional[HipEngine] = None
        self._plan_key = None
        self._seed: Optional[int] = None
        self._row_offset = 0
        self.requires_grad_(False)
        self.eval()

    # ------------------------------------------------------------------ step tables (dyffusion.py:97-138)
    @property
    def diffusion_steps(self) -> List[int]:
        return list(range(0, self.num_timesteps))

    def diffusion_step_to_interpolation_step(self, diffusion_step):
        T = self.num_timesteps
        is_t = torch.is_tensor(diffusion_step)
        ok = bool(((0 <= diffusion_step) & (diffusion_step <= T - 1)).all()) if is_t else 0 <= diffusion_step <= T - 1
        assert ok, f"diffusion_step must be in [1, num_timesteps-1]=[1, {T - 1}], but got {diffusion_step}"
        if self.hparams.schedule == "linear":
            return (diffusion_step + self.di_to_ti_add) / (self.additional_interpolation_steps_fac + 1)
        k = self.additional_diffusion_steps
        if is_t:
            return torch.where(diffusion_step >= k + 1, (diffusion_step - k).float(), diffusion_step / (k + 1))
        return diffusion_step - k if diffusion_step >= k + 1 else diffusion_step / (k + 1)

    # ------------------------------------------------------------------ sampling schedule (dyffusion.py:241-333)
    @property
    def sampling_schedule(self) -> List[Step]:
        return self._sampling_schedule

    @sampling_schedule.setter
    def sampling_schedule(self, schedule):
        T = self.num_timesteps
        name = schedule
        if isinstance(schedule, str):
            dyn = [0] + list(self.dynamical_steps.keys())
            art = list(self.artificial_interpolation_steps.keys())
            if "only_dynamics" in name:
                extra: List[Step] = []
                if "only_dynamics_plus" in name:
                    n_plus = int(name.replace("only_dynamics_plus", "").replace("_discrete", ""))
                    extra = list(np.linspace(0, dyn[1], n_plus + 1, endpoint=False))
                    if "_discrete" in name:
                        extra = [int(np.floor(s)) for s in extra]
                else:
                    assert name == "only_dynamics", f"Invalid sampling schedule: {name}"
            elif name.startswith("every"):
                nth = int(name.replace("every", "").replace("th", "").replace("nd", "").replace("rd", ""))
                assert 1 <= nth <= T, f"Invalid sampling schedule: {name}"
                extra = art[::nth]
            elif name.startswith("first"):
                first = float(name.replace("first", "").replace("v2", ""))
                if first < 1:
                    assert 0 < first < 1, f"Invalid sampling schedule: {name}, must end with number/float > 0"
                    extra = art[: int(np.ceil(first * len(art)))]
                else:
                    assert first.is_integer(), f"If first_n >= 1, it must be an integer, but got {first}"
                    assert 1 <= first <= T, f"Invalid sampling schedule: {name}"
                    extra = art[: int(first)]
            else:
                raise ValueError(f"Invalid sampling schedule: ``{name}``. ")
            schedule = sorted(set(list(extra) + dyn))
        schedule = list(schedule)
        assert 1 <= schedule[-1] <= T, f"Invalid sampling schedule: {schedule}, must end with number/float <= {T}"
        if schedule[0] != 0:
            schedule = [0] + schedule
        for prev, nxt in zip(schedule[:-1], schedule[1:]):
            assert nxt > prev, f"Invalid sampling schedule not monotonically increasing: {schedule}"
        if all(float(s).is_integer() for s in schedule):
            schedule = [int(s) for s in schedule]
        self._sampling_schedule = schedule
        self._plan_key = None

    # ------------------------------------------------------------------ plan (host resolution of dyffusion.py:352-422)
    def _build_plan(self):
        T, hp = self.num_timesteps, self.hparams
        sched = self.sampling_schedule
        steps, out_step = [], 0
        for j, s in enumerate(sched):
            s_next = sched[j + 1] if j + 1 < len(sched) else sched[-1] + 1
            last = s == T - 1
            if hp.time_encoding == "discrete":
                ftime = float(s)
            elif hp.time_encoding == "normalized":
                ftime = s / T
            elif hp.time_encoding == "dynamics":
                ftime = float(self.diffusion_step_to_interpolation_step(s))
            else:
                raise ValueError(f"Invalid time_encoding: {hp.time_encoding}")
            i_next_raw = math.inf if last else self.diffusion_step_to_interpolation_step(s_next)
            emits = last or float(i_next_raw).is_integer()
            out_step = int(i_next_raw) if s < T - 1 else out_step + 1
            i_next = float(self.diffusion_step_to_interpolation_step(s_next)) if s_next <= T - 1 else None
            i_cur = float(self.diffusion_step_to_interpolation_step(s)) if s > 0 else None
            for t in (i_next, i_cur):
                assert t is None or 0 < t < self.interpolator_horizon, \
                    f"interpolate time must be in (0, {self.interpolator_horizon}), got {t}"
            steps.append(dict(forecaster_time=ftime, tau=float(s) / (T - 1), i_next=i_next, i_cur=i_cur,
                              is_last=last, out_slot=(out_step - 1) if emits else None))
        slots = [st["out_slot"] for st in steps if st["out_slot"] is not None]
        if any(sl < 0 for sl in slots):
            raise NotImplementedError("sampling schedules that emit a t0 prediction are not supported")
        n_slots = max(slots) + 1
        refine, extra_keys = [], {}
        if hp.refine_intermediate_predictions:
            times = hp.prediction_timesteps or list(self.dynamical_steps.values())
            emitted = set(slots)
            for i_n in [t for t in times if t < T]:
                if float(i_n).is_integer():
                    assert int(i_n) - 1 in emitted, f"t{int(i_n)}_preds not in intermediates"
                    refine.append((float(i_n), int(i_n) - 1))
                else:  # fractional prediction time (dyffusion.py:414-421): a NEW key "t{i_n}_preds", in a slot of its own
                    extra_keys[n_slots] = f"t{i_n}_preds"
                    refine.append((float(i_n), n_slots))
                    n_slots += 1
        self._extra_keys = extra_keys
        return steps, refine, n_slots

    def _ensure_engine(self, hw, nb: int, sync: bool = True) -> HipEngine:
        if self._engine is None or (self._engine.height, self._engine.width) != tuple(hw) or self._engine.max_batch < nb:
            opts = dict(self._engine_opts)
            opts["max_batch"] = max(opts["max_batch"], nb)
            if self._engine is not None:
                # the engine being replaced is destroyed NOW, not when its last Python reference goes: its captured graphs keep a
                # hardware queue of the process (row-grouped rollouts of another engine then share the remaining ones: OISST 3 990 ->
                # 3 330 fields/s in a bench.py that still held the old NS engine), and the networks are attached to the new one anyway
                self._engine.close()
            self._engine = HipEngine(self.model.engine_net_config(), self._ipol_net.engine_net_config(), hw[0], hw[1], **opts)
            self.model.attach_engine(self._engine, L.NET_FORECASTER)
            self._ipol_net.attach_engine(self._engine, L.NET_INTERPOLATOR)
            self._plan_key = None
            if self._seed is not None:  # a re-created engine (batch growth, new grid) keeps the caller's stream
                self._engine.seed(self._seed)
            self._engine.set_row_offset(self._row_offset)
        # an optimizer / EMA swap may have modified a network in place since its last upload (version counters: ~0.1 ms);
        # the training step refreshes only the engine's training copy instead (sync=False, _p_losses_train)
        if sync:
            self._sync_engine_weights(self._engine)
        return self._engine

    # ------------------------------------------------------------------ stochastic stream (no reference counterpart: the
    # reference draws MC-dropout masks / noise from torch's global generator, i.e. `pl.seed_everything`)
    def seed(self, seed: int):
        """Seed the engine's dropout / noise generator (restarts its stream); survives engine re-creation."""
        self._seed = int(seed)
        if self._engine is not None:
            self._engine.seed(self._seed)

    def set_row_offset(self, first_row: int):
        """Global index of batch row 0 of the tensors this object is given (ensemble sharding, distributed.py)."""
        if int(first_row) == self._row_offset:
            return  # (a blocking 4-byte upload otherwise: keep it out of steady-state sampling loops)
        self._row_offset = int(first_row)
        if self._engine is not None:
            self._engine.set_row_offset(self._row_offset)

    def _ensure_plan(self, eng: HipEngine):
        hp = self.hparams
        key = (tuple(self.sampling_schedule), hp.sampling_type, hp.use_cold_sampling_for_last_step,
               hp.forward_conditioning, hp.refine_intermediate_predictions, hp.time_encoding,
               None if hp.prediction_timesteps is None else tuple(hp.prediction_timesteps),
               bool(self.enable_interpolator_dropout), bool(self.enable_forecaster_dropout), id(eng))
        if key == self._plan_key and eng.plan_valid:  # reloading weights invalidates the engine's plan (FiLM tables)
            return
        if hp.sampling_type not in ("cold", "naive"):
            raise ValueError(f"unknown sampling type {hp.sampling_type}")
        steps, refine, n_slots = self._build_plan()
        # bf16 storage and the ResNet-UNet: ~1e-2 per forward, 4.6e-2 - 5.2e-2 per field over the T = 32 OISST plan (93 chained
        # forwards; tests/test_gpu_unet_resnet.py) -- twice the 2.5e-2 this engine holds its 16-bit rollouts to.  Such a plan is
        # REFUSED in bf16 rather than served at a tolerance nobody asked for: fp16 (the backbone's default) carries 6e-3 over the
        # same rollout at the same speed.  DYF_ALLOW_BF16_LONG_ROLLOUT=1 / allow_bf16_long_rollout overrides (timing experiments).
        # Either network of the pair counts (the interpolator runs most of the chained forwards), and the check comes BEFORE
        # dyf_set_plan: a refused plan leaves the engine's current plan, its FiLM tables and its captured graphs untouched.
        cold = hp.sampling_type == "cold"
        nf = len(steps)  # the count dyf_plan_forward_counts reports for this plan
        ni = len(refine) + sum(1 for st in steps if st["i_next"] is not None) + \
            sum(1 for st in steps if cold and st["i_cur"] is not None and not (st["is_last"] and not hp.use_cold_sampling_for_last_step))
        resnet = any(eng.cfg.net[k].arch == L.ARCH_UNET_RESNET for k in (L.NET_FORECASTER, L.NET_INTERPOLATOR))
        if eng.dtype == "bf16" and resnet and nf + ni > BF16_RESNET_MAX_FORWARDS \
                and not (self.allow_bf16_long_rollout or os.environ.get("DYF_ALLOW_BF16_LONG_ROLLOUT") == "1"):
            raise NotImplementedError(
                f"a rollout of {nf + ni} network forwards of unet.Unet in bf16 storage drifts ~5e-2 from the fp32 reference "
                f"(limit here: {BF16_RESNET_MAX_FORWARDS} forwards, <= 2.5e-2); use dtype='fp16' (the default for this backbone, "
                f"6e-3 over the same rollout) or pass allow_bf16_long_rollout=True to accept the drift")
        eng.set_plan(steps, sampling_cold=cold,
                     cold_for_last_step=hp.use_cold_sampling_for_last_step,
                     forward_conditioning=hp.forward_conditioning, refine=refine, n_out_slots=n_slots,
                     interpolator_dropout=bool(self.enable_interpolator_dropout or self.training),
                     forecaster_dropout=bool(self.enable_forecaster_dropout))
        if (nf, ni) != tuple(eng.forward_counts()):
            # the count above (it gates the bf16 refusal BEFORE dyf_set_plan) and the engine's own disagree: a bug in one of the two.
            # The engine now holds a plan this object did not vet: forget it so the next call re-plans instead of re-failing.
            self._plan_key = None
            eng.plan_valid = False
            raise RuntimeError(f"plan forward counts: host mirror {(nf, ni)} vs dyf_plan_forward_counts {tuple(eng.forward_counts())}")
        self._plan_key = key
        self._plan_steps = steps
        self._emitted_slots = sorted({st["out_slot"] for st in steps if st["out_slot"] is not None})
        self._slot_keys = {sl: f"t{sl + 1}_preds" for sl in self._emitted_slots}
        self._slot_keys.update(self._extra_keys)

    # ------------------------------------------------------------------ reference API
    def sample_loop(self, initial_condition: Tensor, static_condition: Optional[Tensor] = None, log_every_t=None,
                    num_predictions: int = None, _masks=None, _noise=None):
        assert len(initial_condition.shape) == 4, f"condition.shape: {initial_condition.shape} (should be 4D)"
        nb = initial_condition.shape[0]
        eng = self._ensure_engine(initial_condition.shape[-2:], nb)
        self._ensure_plan(eng)
        log_every_t = log_every_t or self.hparams.log_every_t  # dyffusion.py:343-344 ("auto" = 1: any value logs every step)
        if log_every_t is not None:
            eng.set_log_intermediates(True)
        try:
            stack = eng.sample(initial_condition, static_condition, masks=_masks, noise=_noise)
        finally:
            if log_every_t is not None:
                eng.set_log_intermediates(False)
        intermediates = {key: stack[slot] for slot, key in self._slot_keys.items()}
        if log_every_t is not None:  # dyffusion.py:396-406
            cold = self.hparams.sampling_type == "cold"
            for j, (s, st) in enumerate(zip(self.sampling_schedule, self._plan_steps)):
                x_next = eng.get_log(j, 1, nb)
                if st["out_slot"] is not None:
                    intermediates[f"t{st['out_slot'] + 1}_preds2"] = x_next
                intermediates[f"intermediate_{s}_x0hat"] = eng.get_log(j, 0, nb)
                intermediates[f"xipol_{s}_dmodel"] = x_next
                if cold:
                    intermediates[f"xipol_{s}_dmodel2"] = eng.get_log(j, 2, nb)
        x_s = eng.sampler_state(1, nb)
        if self.sampling_schedule[-1] < self.num_timesteps - 1:
            # dyffusion.py:424-425: a schedule that stops before T-1 returns (x_s, intermediates, x_interpolated_s_next)
            return x_s, intermediates, eng.sampler_state(2, nb)
        return eng.sampler_state(0, nb), intermediates, x_s

    # ------------------------------------------------------------------ ensemble sharding over GPUs (distributed.py)
    def sample_stack(self, initial_condition: Tensor, static_condition: Optional[Tensor] = None):
        """The forecast stack of one rollout as ONE contiguous tensor (n_out_slots, NB, C, H, W) plus the {slot: key} map -- what
        `sample` slices its dict from; `distributed.sample_sharded` exchanges it with a single collective."""
        nb = initial_condition.shape[0]
        eng = self._ensure_engine(initial_condition.shape[-2:], nb)
        self._ensure_plan(eng)
        return eng.sample(initial_condition, static_condition), dict(self._slot_keys)

    def comm_init(self, unique_id: bytes, rank: int, world: int, hw, rows_per_rank: int):
        """Give this model's engine its own RCCL communicator (dyf_comm_init) for `sample_gathered`."""
        eng = self._ensure_engine(hw, rows_per_rank)
        if eng.comm_world != 1:
            eng.comm_destroy()
        eng.comm_init(unique_id, rank, world)

    def comm_destroy(self):
        """Drop the live engine's communicator, if it has one (no-op otherwise)."""
        if self._engine is not None and self._engine.comm_world != 1:
            self._engine.comm_destroy()

    def engine_comm_world(self) -> int:
        """World size of the LIVE engine's RCCL communicator (1 = none).  A communicator is owned by the engine it was created on:
        when `_ensure_engine` replaces the engine (new grid, larger batch) it is gone, `sample_sharded` then falls back to the
        torch.distributed exchange until `distributed.init_engine_comm` runs again (an RCCL unique id is single-use, so the new
        engine cannot re-join by itself)."""
        return 1 if self._engine is None else self._engine.comm_world

    @torch.no_grad()
    def sample_gathered(self, initial_condition: Tensor, static_condition: Optional[Tensor], total_rows: int) -> Dict[str, Tensor]:
        """Engine-owned exchange (dyf_sample_gather): sample this rank's rows, all-gather the stack over RCCL inside the engine,
        return the FULL `t{i}_preds` dict ((total_rows, C, H, W) each, global row order)."""
        nb = initial_condition.shape[0]
        eng = self._ensure_engine(initial_condition.shape[-2:], nb)
        self._ensure_plan(eng)
        full = eng.sample_gather(initial_condition, static_condition, total_rows)
        return {key: full[slot] for slot, key in self._slot_keys.items()}

    @torch.no_grad()
    def sample(self, initial_condition: Tensor, num_samples: int = 1, **kwargs) -> Dict[str, Tensor]:
        _, intermediates, _ = self.sample_loop(initial_condition, **kwargs)
        return intermediates

    def predict_forward(self, inputs: Tensor, condition: Tensor = None, metadata=None, **kwargs):
        """_base_diffusion.py:48-68: `condition` is routed to `static_condition`."""
        if inputs is not None and condition is not None:
            kwargs["static_condition"] = condition
        return self.sample(inputs, **kwargs)

    def q_sample(self, x0: Tensor, x_end: Tensor, t: Optional[Tensor], interpolation_time: Optional[Tensor] = None,
                 static_condition: Optional[Tensor] = None, **kwargs) -> Tensor:
        """dyffusion.py:140-163 + :480-494: one interpolator forward I(x_end, x0, i)."""
        assert t is None or interpolation_time is None, "Either t or interpolation_time must be None."
        i_time = interpolation_time if t is None else self.diffusion_step_to_interpolation_step(t)
        assert bool((0 < i_time).all()) and bool((i_time < self.interpolator_horizon).all()), \
            f"interpolate time must be in (0, {self.interpolator_horizon}), got {i_time}"
        return self._interpolate(initial_condition=x_end, x_last=x0, t=i_time, static_condition=static_condition)

    def _interpolate(self, initial_condition: Tensor, x_last: Tensor, t: Tensor, static_condition: Optional[Tensor] = None,
                     **kwargs) -> Tensor:
        """dyffusion.py:480-494: interpolator inputs = cat[initial_condition, x_last] on channels, time in (0, horizon)."""
        assert bool((0 < t).all()) and bool((t < self.interpolator_horizon).all()), \
            f"interpolate time must be in (0, {self.interpolator_horizon}), got {t}"
        eng = self._ensure_engine(x_last.shape[-2:], x_last.shape[0])
        mode = 1 if (self.training or self.enable_interpolator_dropout) else 0
        return eng.net_forward(L.NET_INTERPOLATOR, torch.cat([initial_condition, x_last], dim=1), t.float(), static_condition,
                               dropout_mode=mode if getattr(self._ipol_net, 'has_dropout', True) else 0)

    def get_condition(self, condition: Optional[Tensor], x_last: Optional[Tensor] = None, prediction_type: str = "forward",
                      static_condition: Optional[Tensor] = None, shape=None) -> Optional[Tensor]:
        """dyffusion.py:177-190: the static condition is concatenated behind the dynamical one on the channel axis."""
        if static_condition is None:
            return condition
        if condition is None:
            return static_condition
        return torch.cat([condition, static_condition], dim=1)

    def _predict_last_dynamics(self, forward_condition: Optional[Tensor], x_t: Tensor, t: Tensor) -> Tensor:
        """dyffusion.py:192-203: time encoding + one forecaster forward."""
        enc = self.hparams.time_encoding
        if enc == "discrete":
            time = t
        elif enc == "normalized":
            time = t / self.num_timesteps
        elif enc == "dynamics":
            time = self.diffusion_step_to_interpolation_step(t)
        else:
            raise ValueError(f"Invalid time_encoding: {enc}")
        eng = self._ensure_engine(x_t.shape[-2:], x_t.shape[0])
        return eng.net_forward(L.NET_FORECASTER, x_t, time.float(), forward_condition,
                               dropout_mode=1 if (self.enable_forecaster_dropout and getattr(self.model, 'has_dropout', True)) else 0)

    def predict_x_last(self, condition: Tensor, x_t: Tensor, t: Tensor, is_sampling: bool = False,
                       static_condition: Optional[Tensor] = None) -> Tensor:
        """dyffusion.py:205-239: one forecaster forward F(x_t, enc(t); cond)."""
        assert bool((0 <= t).all()) and bool((t <= self.num_timesteps - 1).all()), f"Invalid timestep: {t}"
        fc = self.hparams.forward_conditioning
        if fc == "data":
            cond = condition
        elif fc == "none":
            cond = None
        elif "data+noise" in fc:
            tf = (t / (self.num_timesteps - 1)).view(condition.shape[0], *[1] * (condition.ndim - 1))
            cond = tf * condition + (1 - tf) * torch.randn_like(condition)
        else:
            raise ValueError(f"Invalid forward conditioning type: {fc}")
        cond = self.get_condition(condition=cond, x_last=None, prediction_type="forward", static_condition=static_condition,
                                  shape=condition.shape)
        return self._predict_last_dynamics(x_t=x_t, forward_condition=cond, t=t)

    def p_losses(self, xt_last: Tensor, condition: Tensor, t: Tensor, static_condition: Optional[Tensor] = None):
        """dyffusion.py:496-567.  Eval mode (`self.training` false): the forecaster objective as the reference evaluates it in
        validation (eval-mode normalisation; the interpolator keeps MC dropout when `enable_interpolator_dropout`), on the
        sampling kernels; returns the reference's loss dict with the "val/" prefix, values are python floats (reduced on the
        GPU by `dyf_criterion`).  Training mode: see `_p_losses_train` (forward with batch-statistics BatchNorm and the backward
        pass on the engine; `out["loss"].backward()` fills the forecaster's `.grad`s)."""
        if self.training:
            return self._p_losses_train(xt_last, condition, t, static_condition)
        lam1, lam2 = self.hparams.lambda_reconstruction, self.hparams.lambda_reconstruction2
        kind = self.hparams.loss_function

        def sub(x, m):
            return None if x is None else x[m]

        eng = self._ensure_engine(xt_last.shape[-2:], xt_last.shape[0])
        x_t = condition.clone()
        nz = t > 0
        if bool(nz.any()):
            x_t[nz] = self.q_sample(x_end=condition[nz], x0=xt_last[nz], t=t[nz], static_condition=sub(static_condition, nz),
                                    num_predictions=1).to(x_t.dtype)
        pred = self.predict_x_last(condition=condition, x_t=x_t, t=t, static_condition=static_condition)
        loss_forward = eng.criterion(pred, xt_last, kind)
        not_last = t <= self.num_timesteps - 2
        loss_forward2 = 0.0
        if lam2 > 0 and bool(not_last.any()):
            t2 = t[not_last] + 1
            sc2 = sub(static_condition, not_last)
            x_i2 = self.q_sample(x_end=condition[not_last], x0=pred[not_last], t=t2, static_condition=sc2, num_predictions=1)
            pred2 = self.predict_x_last(condition=condition[not_last], x_t=x_i2, t=t2, static_condition=sc2)
            loss_forward2 = eng.criterion(pred2, xt_last[not_last].contiguous(), kind)
        return {"loss": lam1 * loss_forward + lam2 * loss_forward2, "val/loss_forward": loss_forward,
                "val/loss_forward2": loss_forward2}

    # ------------------------------------------------------------------ training step
    def train(self, mode: bool = True):
        super().train(mode)
        self.model.eval()  # the parameter containers never run torch ops; the engine is told the mode per call
        return self

    def _sync_engine_weights(self, eng: HipEngine):
        """Re-upload a network whose parameters were modified in place since the last upload (optimizer.step())."""
        sync_weights(self.model, eng, L.NET_FORECASTER)
        sync_weights(self._ipol_net, eng, L.NET_INTERPOLATOR)

    def _p_losses_train(self, xt_last: Tensor, condition: Tensor, t: Tensor, static_condition: Optional[Tensor] = None):
        """`p_losses` with `self.training` (dyffusion.py:496-567 under torch.autograd in the reference).  The forecaster runs in
        train mode (batch-statistics BatchNorm with running-statistics update, Dropout active), the frozen interpolator in
        eval mode with its Dropout active (:154-160: `self.training or enable_interpolator_dropout`); both loss terms; the
        second term is differentiated THROUGH the interpolator.  Returns the reference's dict ("train/" prefix); `out["loss"]`
        is a scalar tensor whose `.backward()` runs the engine's backward pass and ACCUMULATES into `param.grad` of the
        forecaster's parameters (so `torch.optim` / Lightning's `training_step` work unchanged).  arch unet_simple, fp32."""
        lam1, lam2 = self.hparams.lambda_reconstruction, self.hparams.lambda_reconstruction2
        kind = self.hparams.loss_function
        eng = self._ensure_engine(xt_last.shape[-2:], xt_last.shape[0], sync=False)
        sync_train_weights(self.model, eng, L.NET_FORECASTER)        # fp32 training copy only: no re-packing of the sampling path
        sync_train_weights(self._ipol_net, eng, L.NET_INTERPOLATOR)  # frozen: a no-op after the first step
        ipol_drop = bool(getattr(self._ipol_net, "has_dropout", True))
        f_drop = bool(getattr(self.model, "has_dropout", True))
        T = self.num_timesteps

        def sub(x, m):
            return None if x is None else x[m].contiguous()

        def f_inputs(cond_data, tt, sc):  # predict_x_last (dyffusion.py:205-239): conditioning + time encoding
            fc = self.hparams.forward_conditioning
            if fc == "data":
                cond = cond_data
            elif fc == "none":
                cond = None
            else:
                tf = (tt / (T - 1)).view(cond_data.shape[0], *[1] * (cond_data.ndim - 1))
                cond = tf * cond_data + (1 - tf) * torch.randn_like(cond_data)
            if sc is not None:
                cond = sc if cond is None else torch.cat([cond, sc], dim=1)
            enc = self.hparams.time_encoding
            time = tt.float() if enc == "discrete" else tt / T if enc == "normalized" else self.diffusion_step_to_interpolation_step(tt)
            return time.float(), cond

        x_t = condition.clone()
        nz = t > 0
        if bool(nz.any()):
            it = self.diffusion_step_to_interpolation_step(t[nz]).float()
            x_t[nz] = eng.train_forward(L.NET_INTERPOLATOR, 0, torch.cat([condition[nz], xt_last[nz]], 1), it, sub(static_condition, nz),
                                        batch_stats=False, dropout=ipol_drop)
        time1, cond1 = f_inputs(condition, t, static_condition)
        pred = eng.train_forward(L.NET_FORECASTER, 1, x_t, time1, cond1, batch_stats=True, dropout=f_drop)
        loss_forward = eng.criterion(pred, xt_last, kind)
        not_last = t <= T - 2
        eng.train_step_id += 1
        state = dict(eng=eng, pred=pred, target=xt_last, kind=kind, lam1=lam1, lam2=lam2, not_last=None, n_fwd=1,
                     step_id=eng.train_step_id)
        loss_forward2 = 0.0
        if lam2 > 0 and bool(not_last.any()):
            t2 = t[not_last] + 1
            sc2 = sub(static_condition, not_last)
            cond_nl = condition[not_last].contiguous()
            it2 = self.diffusion_step_to_interpolation_step(t2).float()
            x_i2 = eng.train_forward(L.NET_INTERPOLATOR, 2, torch.cat([cond_nl, pred[not_last]], 1), it2, sc2, batch_stats=False,
                                     dropout=ipol_drop)
            time2, cond2 = f_inputs(cond_nl, t2, sc2)
            pred2 = eng.train_forward(L.NET_FORECASTER, 3, x_i2, time2, cond2, batch_stats=True, dropout=f_drop)
            target2 = xt_last[not_last].contiguous()
            loss_forward2 = eng.criterion(pred2, target2, kind)
            state.update(not_last=not_last, pred2=pred2, target2=target2, n_fwd=2)
        total = lam1 * loss_forward + lam2 * loss_forward2
        self._train_state = state
        if not hasattr(self, "_grad_anchor"):
            self._grad_anchor = torch.zeros((), requires_grad=True)
        loss = EngineLoss.apply(self._grad_anchor, self, float(total))
        return {"loss": loss, "train/loss_forward": loss_forward, "train/loss_forward2": loss_forward2}

    def _train_backward(self, upstream: float):
        st = self._train_state
        eng, C = st["eng"], self.num_output_channels  # the x_last part of the interpolator's inputs cat[x_0 window, x_last]
        d_pred = eng.criterion_grad(st["pred"], st["target"], st["kind"], st["lam1"] * upstream)
        if st["not_last"] is not None:
            d_pred2 = eng.criterion_grad(st["pred2"], st["target2"], st["kind"], st["lam2"] * upstream)
            d_xi2 = eng.train_backward(3, d_pred2, want_dinputs=True, param_grads=True)    # second forecaster pass
            d_ipol = eng.train_backward(2, d_xi2, want_dinputs=True, param_grads=False)    # through the frozen interpolator
            d_pred[st["not_last"]] += d_ipol[:, -C:]                                        # inputs = cat[x_0 window, x_last = pred]
        eng.train_backward(1, d_pred, want_dinputs=False, param_grads=True)
        collect_train_results(self.model, eng, L.NET_FORECASTER, st["n_fwd"])

    def forward(self, inputs: Tensor, targets: Tensor = None, condition: Tensor = None, time: Tensor = None):
        """`BaseDiffusion.forward` (_base_diffusion.py:81-106): draws one diffusion step per batch item unless `time` is given
        and evaluates `p_losses(targets, condition=inputs, t, static_condition=condition)`."""
        b = (targets if targets is not None else inputs).shape[0]
        t = time if time is not None else torch.randint(0, self.num_timesteps, (b,), device=inputs.device, dtype=torch.long)
        return self.p_losses(targets, condition=inputs, t=t, static_condition=condition)

    def get_loss(self, inputs: Tensor, targets: Tensor, metadata=None, **kwargs):
        """`BaseDiffusion.get_loss` (_base_diffusion.py:108-117)."""
        return self(inputs, targets, **kwargs)
