"""ORACLE (test infrastructure, CPU fp32) -- DYffusion sampling loop restated as a flat function.

Restates /root/reference/src/diffusion/dyffusion.py:335-426 (sample_loop), :205-239 (predict_x_last),
:140-163 + :480-494 (q_sample/_interpolate), given two callables for the networks.  The scalar
bookkeeping is resolved ahead of time by oracle.schedule.build_sampling_plan.
Parity: pinned against tests/golden/sample_*.npz (outputs of the imported reference's
`MultiHorizonForecastingDYffusion.predict`) in tests/test_oracle_sampler.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from typing import Callable, Dict, Optional, Sequence

import torch
from torch import Tensor

from .schedule import build_sampling_plan, build_step_tables, parse_sampling_schedule, refine_times

NetFn = Callable[[Tensor, Tensor, Optional[Tensor]], Tensor]  # (inputs, time, condition) -> prediction


def sample_loop(forecaster: NetFn, interpolator: NetFn, x_init: Tensor, static_condition: Optional[Tensor],
                cfg: dict, noise_fn: Optional[Callable[[Tensor], Tensor]] = None) -> Dict[str, Tensor]:
    """cfg keys (reference kwarg names): timesteps, schedule, additional_interpolation_steps,
    additional_interpolation_steps_factor, interpolate_before_t1, sampling_type, sampling_schedule, time_encoding,
    refine_intermediate_predictions, prediction_timesteps, use_cold_sampling_for_last_step, forward_conditioning,
    log_every_t, num_input_channels (C of the dynamics).

    `forecaster(x_s, time, cond)` and `interpolator(cat[x_init, x_last], time, static)` are net forwards; the
    interpolator callable owns its dropout source (MC dropout on/off is the caller's choice).
    """
    tab = build_step_tables(cfg["timesteps"], cfg.get("schedule", "before_t1_only"),
                            cfg.get("additional_interpolation_steps", 0),
                            cfg.get("additional_interpolation_steps_factor", 0),
                            cfg.get("interpolate_before_t1", False))
    sched = parse_sampling_schedule(tab, cfg.get("sampling_schedule"))
    plan = build_sampling_plan(tab, sched, cfg.get("time_encoding", "dynamics"))
    kind = cfg.get("sampling_type", "cold")
    fcond = cfg.get("forward_conditioning", "data")
    cold_last = cfg.get("use_cold_sampling_for_last_step", False)
    noise_fn = noise_fn or torch.randn_like
    assert x_init.dim() == 4, f"condition.shape: {x_init.shape} (should be 4D)"
    nb = x_init.shape[0]
    C = cfg.get("num_input_channels") or x_init.shape[1]  # window == 1 unless stated

    def full(v):
        return torch.full((nb,), float(v), dtype=torch.float32)

    def interp(x_last, i_time):
        assert 0 < i_time < tab.horizon, f"interpolate time must be in (0, {tab.horizon}), got {i_time}"
        return interpolator(torch.cat([x_init, x_last], dim=1), full(i_time), static_condition)

    def forecast(x_s, step):
        if fcond == "data":
            cond = x_init
        elif fcond == "none":
            cond = None
        elif "data+noise" in fcond:
            cond = step.tau * x_init + (1.0 - step.tau) * noise_fn(x_init)
        else:
            raise ValueError(f"Invalid forward conditioning type: {fcond}")
        if static_condition is not None:
            cond = static_condition if cond is None else torch.cat([cond, static_condition], dim=1)
        return forecaster(x_s, full(step.forecaster_time), cond)

    x_s = x_init[:, -C:]
    out: Dict[str, Tensor] = {}
    x_last_hat = None
    log = cfg.get("log_every_t") is not None  # dyffusion.py:343-344, 398-406: per-step intermediates next to the forecasts
    x_cur = None
    for st in plan:
        x_last_hat = forecast(x_s, st)
        x_next = interp(x_last_hat, st.i_next) if st.i_next is not None else x_last_hat
        if kind == "cold":
            if st.is_last and not cold_last:
                x_s = x_last_hat
            else:
                x_cur = interp(x_last_hat, st.i_cur) if st.i_cur is not None else x_s
                x_s = x_s - x_cur + x_next
        elif kind == "naive":
            x_s = x_next
        else:
            raise ValueError(f"unknown sampling type {kind}")
        if st.out_step is not None:
            out[f"t{st.out_step}_preds"] = x_s
            if log:
                out[f"t{st.out_step}_preds2"] = x_next
        if log:
            out[f"intermediate_{st.s}_x0hat"] = x_last_hat
            out[f"xipol_{st.s}_dmodel"] = x_next
            if kind == "cold":  # the reference logs the variable as it stands: on a last step without cold sampling, the previous step's
                out[f"xipol_{st.s}_dmodel2"] = x_cur
    if cfg.get("refine_intermediate_predictions", False):
        for i_n in refine_times(tab, cfg.get("prediction_timesteps")):
            key = int(i_n) if float(i_n).is_integer() else i_n
            assert not float(i_n).is_integer() or f"t{key}_preds" in out, f"t{key}_preds not in intermediates"
            out[f"t{key}_preds"] = interp(x_last_hat, i_n)
    return out


def reshape_ensemble(preds: Dict[str, Tensor], num_predictions: int) -> Dict[str, Tensor]:
    """(N*B, ...) -> (N, B, ...), ensemble-major rows (row = n*B + b); _base_experiment.py:358-379,540-567."""
    res = {}
    for k, v in preds.items():
        b = v.shape[0]
        if num_predictions > 1 and b > 1 and b % num_predictions == 0:
            v = v.reshape(num_predictions, max(1, b // num_predictions), *v.shape[1:])
        res[k] = v
    return res


def count_net_evals(cfg: dict) -> Dict[str, int]:
    """#forecaster / #interpolator forwards of one rollout: used by bench.py for FLOP accounting."""
    tab = build_step_tables(cfg["timesteps"], cfg.get("schedule", "before_t1_only"),
                            cfg.get("additional_interpolation_steps", 0),
                            cfg.get("additional_interpolation_steps_factor", 0),
                            cfg.get("interpolate_before_t1", False))
    plan = build_sampling_plan(tab, parse_sampling_schedule(tab, cfg.get("sampling_schedule")),
                               cfg.get("time_encoding", "dynamics"))
    cold = cfg.get("sampling_type", "cold") == "cold"
    cold_last = cfg.get("use_cold_sampling_for_last_step", False)
    n_f = len(plan)
    n_i = sum(1 for st in plan if st.i_next is not None)
    if cold:
        n_i += sum(1 for st in plan if st.i_cur is not None and not (st.is_last and not cold_last))
    if cfg.get("refine_intermediate_predictions", False):
        n_i += len(refine_times(tab, cfg.get("prediction_timesteps")))
    return {"forecaster": n_f, "interpolator": n_i}
