"""ORACLE (test infrastructure, CPU only) -- DYffusion step bookkeeping restated from the reference.

Restates, as plain functions over ints/floats, the integer/float schedule logic of
  /root/reference/src/diffusion/dyffusion.py:44-95    (table construction in BaseDYffusion.__init__)
  /root/reference/src/diffusion/dyffusion.py:101-138  (diffusion_step_to_interpolation_step)
  /root/reference/src/diffusion/dyffusion.py:245-333  (sampling_schedule setter)
Parity: pinned against tests/golden/schedules.json (generated from the imported reference by
tests/golden/make_golden.py) in tests/test_oracle_schedule.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Union

Number = Union[int, float]


@dataclass
class StepTables:
    """Everything the sampler needs to know about the diffusion-step <-> interpolation-time mapping."""

    horizon: int
    schedule: str
    extra_steps: int            # additional diffusion steps on top of `horizon` (k)
    linear_factor: int          # additional_interpolation_steps_factor (linear schedule only)
    linear_offset: int          # di_to_ti_add of the reference (linear schedule only)
    num_timesteps: int          # T = horizon + extra_steps
    d_to_i: Dict[int, Number] = field(default_factory=dict)
    dynamical_steps: Dict[int, Number] = field(default_factory=dict)
    artificial_steps: Dict[int, Number] = field(default_factory=dict)
    i_to_d: Dict[Number, int] = field(default_factory=dict)

    def interpolation_time(self, d: Number) -> Number:
        """d (diffusion step, possibly fractional) -> interpolation time i(d).  dyffusion.py:101-138."""
        if not (0 <= d <= self.num_timesteps - 1):
            raise AssertionError(
                f"diffusion_step must be in [1, num_timesteps-1]=[1, {self.num_timesteps - 1}], but got {d}")
        if self.schedule == "linear":
            return (d + self.linear_offset) / (self.linear_factor + 1)
        # before_t1_only: the last h-1 diffusion steps are the dynamical times 1..h-1; the first k ones are
        # spread uniformly over (0, 1).
        k = self.extra_steps
        if d >= k + 1:
            return d - k
        return d / (k + 1)


def build_step_tables(horizon: int, schedule: str = "before_t1_only", additional_interpolation_steps: int = 0,
                      additional_interpolation_steps_factor: int = 0,
                      interpolate_before_t1: bool = False) -> StepTables:
    """dyffusion.py:44-95."""
    if not horizon > 1:
        raise AssertionError(f"horizon must be > 1, but got {horizon}. Please use datamodule.horizon with > 1")
    lin_fac, lin_off = 0, 0
    if schedule == "linear":
        if additional_interpolation_steps != 0:
            raise AssertionError("additional_interpolation_steps must be 0 when using linear schedule")
        lin_fac = additional_interpolation_steps_factor
        if interpolate_before_t1:
            n_between, lin_off = horizon - 1, 0
        else:
            n_between, lin_off = horizon - 2, additional_interpolation_steps_factor
        extra = additional_interpolation_steps_factor * n_between
    elif schedule == "before_t1_only":
        if additional_interpolation_steps_factor != 0:
            raise AssertionError(
                "additional_interpolation_steps_factor must be 0 when using before_t1_only schedule")
        if not interpolate_before_t1:
            raise AssertionError("interpolate_before_t1 must be True when using before_t1_only schedule")
        extra = additional_interpolation_steps
    else:
        raise ValueError(f"Invalid schedule: {schedule}")
    tab = StepTables(horizon=horizon, schedule=schedule, extra_steps=extra, linear_factor=lin_fac,
                     linear_offset=lin_off, num_timesteps=horizon + extra)
    for d in range(1, tab.num_timesteps):
        i = tab.interpolation_time(d)
        tab.d_to_i[d] = i
        tab.i_to_d[i] = d
        if float(i).is_integer():
            tab.dynamical_steps[d] = i
        else:
            tab.artificial_steps[d] = i
    return tab


def parse_sampling_schedule(tab: StepTables, spec: Union[None, str, Sequence[Number]]) -> List[Number]:
    """dyffusion.py:245-333.  Returns the ascending list of diffusion steps visited while sampling."""
    T = tab.num_timesteps
    if spec is None or spec == "None":
        sched: List[Number] = list(range(0, T))
    elif isinstance(spec, str):
        base = [0] + list(tab.dynamical_steps.keys())
        artificial = list(tab.artificial_steps.keys())
        if "only_dynamics" in spec:
            picked: List[Number] = []
            if "only_dynamics_plus" in spec:
                n_plus = int(spec.replace("only_dynamics_plus", "").replace("_discrete", ""))
                first_dyn = base[1]
                # numpy.linspace(0, first_dyn, n_plus + 1, endpoint=False)
                picked = [first_dyn * j / (n_plus + 1) for j in range(n_plus + 1)]
                if "_discrete" in spec:
                    picked = [int(math.floor(s)) for s in picked]
            elif spec != "only_dynamics":
                raise AssertionError(f"Invalid sampling schedule: {spec}")
        elif spec.startswith("every"):
            nth = int(spec.replace("every", "").replace("th", "").replace("nd", "").replace("rd", ""))
            if not 1 <= nth <= T:
                raise AssertionError(f"Invalid sampling schedule: {spec}")
            picked = artificial[::nth]
        elif spec.startswith("first"):
            first_n = float(spec.replace("first", "").replace("v2", ""))
            if first_n < 1:
                if not 0 < first_n < 1:
                    raise AssertionError(f"Invalid sampling schedule: {spec}, must end with number/float > 0")
                picked = artificial[: int(math.ceil(first_n * len(artificial)))]
            else:
                if not first_n.is_integer():
                    raise AssertionError(f"If first_n >= 1, it must be an integer, but got {first_n}")
                if not 1 <= first_n <= T:
                    raise AssertionError(f"Invalid sampling schedule: {spec}")
                picked = artificial[: int(first_n)]
        else:
            raise ValueError(f"Invalid sampling schedule: ``{spec}``. ")
        sched = sorted(set(list(picked) + base))
    else:
        sched = list(spec)

    if not 1 <= sched[-1] <= T:
        raise AssertionError(f"Invalid sampling schedule: {sched}, must end with number/float <= {T}")
    if sched[0] != 0:
        sched = [0] + sched
    for a, b in zip(sched[:-1], sched[1:]):
        if not b > a:
            raise AssertionError(f"Invalid sampling schedule not monotonically increasing: {sched}")
    if all(float(s).is_integer() for s in sched):
        sched = [int(s) for s in sched]
    return sched


@dataclass
class PlanStep:
    """One iteration of the sampling loop, fully resolved on the host (no device-side control flow)."""

    s: Number                       # diffusion step fed to the forecaster
    s_next: Number
    forecaster_time: float          # enc(s) according to time_encoding
    tau: float                      # s / (T-1): mixing factor for forward_conditioning="data+noise"
    i_cur: Optional[float]          # interpolation time of s (None when s == 0 -> uses x_s itself)
    i_next: Optional[float]         # interpolation time of s_next (None when s_next > T-1 -> uses x0_hat)
    is_last: bool
    out_step: Optional[int]         # dynamics index t{out_step}_preds written after this iteration (or None)


def build_sampling_plan(tab: StepTables, sched: Sequence[Number], time_encoding: str = "dynamics") -> List[PlanStep]:
    """Resolves dyffusion.py:335-397 (the per-iteration scalar bookkeeping of sample_loop) ahead of time."""
    T = tab.num_timesteps
    plan: List[PlanStep] = []
    after_last = sched[-1] + 1
    out_idx = 0
    for j, s in enumerate(sched):
        s_next = sched[j + 1] if j + 1 < len(sched) else after_last
        is_last = s == T - 1
        if time_encoding == "discrete":
            ftime = float(s)
        elif time_encoding == "normalized":
            ftime = s / T
        elif time_encoding == "dynamics":
            ftime = float(tab.interpolation_time(s))
        else:
            raise ValueError(f"Invalid time_encoding: {time_encoding}")
        i_next_val = math.inf if is_last else tab.interpolation_time(s_next)
        emits = is_last or float(i_next_val).is_integer()
        i_next = float(tab.interpolation_time(s_next)) if s_next <= T - 1 else None
        i_cur = float(tab.interpolation_time(s)) if s > 0 else None
        out_idx = int(i_next_val) if s < T - 1 else out_idx + 1
        plan.append(PlanStep(s=s, s_next=s_next, forecaster_time=ftime, tau=float(s) / (T - 1), i_cur=i_cur,
                             i_next=i_next, is_last=is_last, out_step=out_idx if emits else None))
    return plan


def refine_times(tab: StepTables, prediction_timesteps: Optional[Sequence[Number]] = None) -> List[Number]:
    """Interpolation times re-predicted by the refinement pass, dyffusion.py:408-422."""
    times = list(prediction_timesteps) if prediction_timesteps else list(tab.dynamical_steps.values())
    return [i for i in times if i < tab.num_timesteps]
