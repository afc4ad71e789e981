"""Helpers for the `-m gpu` parity tests: build HIP engines from oracle-style parameter dicts."""
import json

import numpy as np
import torch

import dyffusion_amd as D
from dyffusion_amd import _lib as L
from oracle import init as oinit
from oracle import nets, sampler

DEV = "cuda:0"


def mirror_from_params(P, cfg, n_in, n_cond, n_out):
    net = D.UNet(dim=cfg["dim"], with_time_emb=cfg.get("with_time_emb", True), upsample_dims=cfg.get("upsample_dims"),
                 outer_sample_mode=cfg.get("outer_sample_mode", "bilinear"), dropout=cfg.get("dropout", 0.0), input_dropout=cfg.get("input_dropout", 0.0), num_input_channels=n_in,
                 num_output_channels=n_out, num_conditional_channels=n_cond)
    net.load_state_dict(P, strict=True)
    return net


def nhwc_masks(masks_nchw):
    """oracle keep-masks (uint8, NCHW) -> engine layout (uint8, NHWC) on the GPU"""
    return [m.permute(0, 2, 3, 1).contiguous().to(DEV) for m in masks_nchw]


def seeded_pair(dim, C, Cs, window=1, fcond_channels=0, seeds=(101, 102)):
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(dim, C + fcond_channels + Cs, C), seeds[0])
    PI = oinit.seeded_state(oinit.unet_simple_param_shapes(dim, (window + 1) * C + Cs, C), seeds[1])
    return PF, PI


def build_dyffusion(PF, PI, mcfg, C, Cs, hp, window=1, **engine_kw):
    fc = hp.get("forward_conditioning", "none")
    f_cond = Cs + (0 if fc == "none" else window * C)
    F = mirror_from_params(PF, mcfg, C, f_cond, C)
    I = mirror_from_params(PI, mcfg, (window + 1) * C, Cs, C)
    keys = ["forward_conditioning", "schedule", "additional_interpolation_steps", "additional_interpolation_steps_factor",
            "interpolate_before_t1", "sampling_type", "sampling_schedule", "time_encoding",
            "refine_intermediate_predictions", "prediction_timesteps", "use_cold_sampling_for_last_step", "enable_interpolator_dropout",
            "lambda_reconstruction", "lambda_reconstruction2", "loss_function", "log_every_t"]
    kw = {k: hp[k] for k in keys if k in hp}
    return D.DYffusion(F, D.InterpolatorHandle(I, hp["timesteps"], window), timesteps=hp["timesteps"], **kw, **engine_kw)


def oracle_rollout(PF, PI, mcfg, hp, x0, c, drop=None, noise_fn=None):
    drop = drop or nets.DropoutOff()
    with torch.no_grad():
        return sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward(PF, mcfg, x, t, cond),
                                   lambda x, t, cond: nets.unet_simple_forward(PI, mcfg, x, t, cond, dropout=drop),
                                   x0, c, hp, noise_fn=noise_fn)


_ORACLE_CACHE = {}


def cached(key, fn):
    """Memo for CPU-oracle results shared by the parametrisations of one test (bf16 / fp16 builds compare with the SAME fp32 oracle
    output): the full-size oracle runs are what the GPU suite spends its time on."""
    if key not in _ORACLE_CACHE:
        _ORACLE_CACHE[key] = fn()
    return _ORACLE_CACHE[key]
