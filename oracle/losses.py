"""ORACLE (test infrastructure, CPU fp32) -- the DYffusion forecaster objective.

Restates /root/reference/src/diffusion/dyffusion.py:496-567 (`p_losses`) together with the two helpers it calls,
:140-163 + :480-494 (`q_sample` / `_interpolate`) and :191-239 (`_predict_last_dynamics` / `predict_x_last`), given two
callables for the networks.  With eval-mode callables this is the objective as the reference evaluates it in validation;
with a training-mode forecaster callable (`nets.unet_simple_forward(..., bn_training=True, dropout=...)`) and parameters that
require grad, torch.autograd over it is the reference's training step (the callables decide the mode, as module.train() /
the interpolator's frozen eval mode do in the reference).
Parity: pinned against tests/golden/plosses_*.npz (outputs of the imported reference's `DYffusion.p_losses`) and
plosses_train_*.npz (its losses AND gradients in training mode) in tests/test_oracle_losses.py.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
"""
from typing import Callable, Dict, Optional

import torch
from torch import Tensor

from .schedule import build_step_tables

NetFn = Callable[[Tensor, Tensor, Optional[Tensor]], Tensor]  # (inputs, time, condition) -> prediction


def criterion_fn(name: str):
    """src/utilities/utils.py:201-212 (`get_loss`, reduction="mean")."""
    name = name.lower().strip().replace("-", "_")
    if name in ("l1", "mae", "mean_absolute_error"):
        return lambda a, b: (a - b).abs().mean()
    if name in ("l2", "mse", "mean_squared_error"):
        return lambda a, b: ((a - b) ** 2).mean()
    if name in ("smoothl1", "smooth"):
        return torch.nn.functional.smooth_l1_loss
    raise ValueError(f"Unknown loss function {name}")


def p_losses(forecaster: NetFn, interpolator: NetFn, xt_last: Tensor, condition: Tensor, t: Tensor,
             static_condition: Optional[Tensor], cfg: dict, noise_fn=None) -> Dict[str, Tensor]:
    """cfg keys (reference kwarg names): timesteps, schedule, additional_interpolation_steps,
    additional_interpolation_steps_factor, interpolate_before_t1, time_encoding, forward_conditioning,
    lambda_reconstruction, lambda_reconstruction2, loss_function."""
    tab = build_step_tables(cfg["timesteps"], cfg.get("schedule", "before_t1_only"),
                            cfg.get("additional_interpolation_steps", 0),
                            cfg.get("additional_interpolation_steps_factor", 0),
                            cfg.get("interpolate_before_t1", False))
    T = tab.num_timesteps
    fcond = cfg.get("forward_conditioning", "data")
    enc = cfg.get("time_encoding", "dynamics")
    lam1, lam2 = cfg.get("lambda_reconstruction", 1.0), cfg.get("lambda_reconstruction2", 0.0)
    crit = criterion_fn(cfg.get("loss_function", "mse"))
    noise_fn = noise_fn or torch.randn_like

    def i_time(tt: Tensor) -> Tensor:  # dyffusion.py:101-138, per batch row
        return torch.tensor([float(tab.interpolation_time(float(v))) for v in tt], dtype=torch.float32)

    def q_sample(x_end, x0, tt, sc):  # interpolator in "interpolation mode": I(x_end = t0 data, x0 = last data, i(t))
        it = i_time(tt)
        assert bool(((0 < it) & (it < tab.horizon)).all()), f"interpolate time must be in (0, {tab.horizon}), got {it}"
        return interpolator(torch.cat([x_end, x0], dim=1), it, sc)

    def predict_x_last(cond_data, x_t, tt, sc):
        assert bool(((0 <= tt) & (tt <= T - 1)).all()), f"Invalid timestep: {tt}"
        if fcond == "data":
            cond = cond_data
        elif fcond == "none":
            cond = None
        elif "data+noise" in fcond:
            tf = (tt / (T - 1)).view(cond_data.shape[0], *[1] * (cond_data.ndim - 1))
            cond = tf * cond_data + (1 - tf) * noise_fn(cond_data)
        else:
            raise ValueError(f"Invalid forward conditioning type: {fcond}")
        if sc is not None:
            cond = sc if cond is None else torch.cat([cond, sc], dim=1)
        time = tt.float() if enc == "discrete" else tt / T if enc == "normalized" else i_time(tt)
        return forecaster(x_t, time, cond)

    def sub(x, m):
        return None if x is None else x[m]

    # 1. forecaster inputs: the initial condition for t = 0, an interpolated state for t > 0
    x_t = condition.clone()
    nz = t > 0
    if bool(nz.any()):
        x_t[nz] = q_sample(condition[nz], xt_last[nz], t[nz], sub(static_condition, nz)).to(x_t.dtype)
    # 2. predict the last state from x_t
    pred = predict_x_last(condition, x_t, t, static_condition)
    loss_forward = crit(pred, xt_last)
    # 3. one more emulated step: interpolate with the PREDICTED last state, predict again
    not_last = t <= T - 2
    loss_forward2 = torch.zeros(())
    if lam2 > 0 and bool(not_last.any()):
        t2 = t[not_last] + 1
        sc2 = sub(static_condition, not_last)
        x_i2 = q_sample(condition[not_last], pred[not_last], t2, sc2)
        pred2 = predict_x_last(condition[not_last], x_i2, t2, sc2)
        loss_forward2 = crit(pred2, xt_last[not_last])
    return {"loss": lam1 * loss_forward + lam2 * loss_forward2, "loss_forward": loss_forward, "loss_forward2": loss_forward2,
            "xt_last_pred": pred}


def interpolation_loss(interpolator: NetFn, dynamics: Tensor, t: Tensor, condition: Optional[Tensor], window: int,
                       loss_function: str) -> Tensor:
    """Stage-1 objective of the interpolator: /root/reference/src/experiment_types/interpolation.py:149-167 (`get_loss`,
    given the drawn interpolation times t in horizon_range = 1..h-1) with :128-141 (`get_inputs_from_dynamics`, window frames
    stacked on channels + the last frame) and src/models/_base_model.py:108-138 (`BaseModel.get_loss`: predict, criterion).
    dynamics (b, window + h, c, H, W).  Pinned by tests/golden/interp_train_*.npz (loss and gradients of the imported
    reference in train mode) in tests/test_oracle_losses.py."""
    b = dynamics.shape[0]
    past = dynamics[:, :window].reshape(b, -1, *dynamics.shape[-2:])  # "b window c lat lon -> b (window c) lat lon"
    inputs = torch.cat([past, dynamics[:, -1]], dim=1)
    targets = dynamics[torch.arange(b), window + t.long() - 1]
    pred = interpolator(inputs, t, condition)
    return criterion_fn(loss_function)(pred, targets)


# ------------------------------------------------------------------------------------------------ mixed-precision training model
# The engine's `train_precision=16` ("bf16-mixed"; csrc/train_gemm.hip, csrc/train_halo16.hip, csrc/train_internal.h): every tensor,
# the master weights, the accumulators and the statistics stay fp32; a training convolution that runs on the 16-bit matrix cores
# rounds its two OPERANDS to bf16 (round-to-nearest-even) while it stages them -- the forward (x, w), the data gradient (dz, w) and
# the weight gradient (dz, x) each on their own.  Which of the three run on the 16-bit cores is a function of the layer's channel
# counts (the dispatch of csrc/train_gemm.hip tgemm_conv_fwd / _dgrad / _wgrad): the 1x1 stem (cin = 5 / 8), the readout's
# transposed conv (3 channels) and every Linear / attention contraction stay fp32.
# This is a MODEL of that arithmetic under torch.autograd (same roundings at the same places, fp32 accumulation by ATen instead of
# the MFMA's order): the reference has no such mode (Lightning's precision=16 is autocast fp16 + GradScaler), so there is no
# golden vector for it -- tests/test_gpu_training.py holds the engine's 16-bit step to this model, and this model's fp32 limit
# (rounding off) IS the reference-pinned oracle above.
def bf16_round(x: Tensor) -> Tensor:
    return x.to(torch.bfloat16).to(torch.float32)


def conv_operand_rules(cin: int, cout: int):
    """(forward, data gradient, weight gradient) run with bf16 operands?  csrc/train_gemm.hip: GN = 64, GK = 16, GK16 = 32."""
    fwd = cin % 16 == 0 and cout % 64 == 0 and cin % 32 == 0
    dgrad = cout % 16 == 0 and cin % 64 == 0 and cout % 32 == 0
    wgrad = cin % 64 == 0 and cout % 4 == 0
    return fwd, dgrad, wgrad


class _Conv2dRoundedOperands(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, w, bias, stride, padding):
        cout, cin = w.shape[0], w.shape[1]
        f16, d16, w16 = conv_operand_rules(cin, cout)
        ctx.save_for_backward(x, w)
        ctx.cfg = (stride, padding, d16, w16, bias is not None)
        xr, wr = (bf16_round(x), bf16_round(w)) if f16 else (x, w)
        return torch.nn.functional.conv2d(xr, wr, bias, stride=stride, padding=padding)

    @staticmethod
    def backward(ctx, dz):
        x, w = ctx.saved_tensors
        stride, padding, d16, w16, has_bias = ctx.cfg
        dz = dz.contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            a, b = (bf16_round(dz), bf16_round(w)) if d16 else (dz, w)
            dx = torch.nn.grad.conv2d_input(x.shape, b, a, stride=stride, padding=padding)
        if ctx.needs_input_grad[1]:
            a, b = (bf16_round(dz), bf16_round(x)) if w16 else (dz, x)
            dw = torch.nn.grad.conv2d_weight(b, w.shape, a, stride=stride, padding=padding)
        if has_bias and ctx.needs_input_grad[2]:
            db = dz.sum(dim=(0, 2, 3))  # the bias gradient is a plain fp32 reduction of dz in the engine too
        return dx, dw, db, None, None


class training_operand_rounding:
    """`with training_operand_rounding(): ...` -- every nn.Conv2d of the restated backbones (oracle.nets.conv2d) runs as the
    engine's bf16-mixed training convolution for the duration of the block."""

    def __enter__(self):
        from . import nets
        self._nets = nets
        self._prev = nets._CONV2D[0]
        nets._CONV2D[0] = lambda x, w, bias=None, stride=1, padding=0: _Conv2dRoundedOperands.apply(x, w, bias, stride, padding)
        return self

    def __exit__(self, *exc):
        self._nets._CONV2D[0] = self._prev
        return False
