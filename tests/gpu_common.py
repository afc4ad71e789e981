"""Helpers for the `-m gpu` parity tests: build HIP engines from oracle-style parameter dicts."""
import json
import os

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
_CACHE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "oracle_cache")
_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def oracle_code_hash():
    """sha-256 over the sources an oracle result is a function of: oracle/*.py (the restatement) and tests/rng_host.py (the host
    rebuild of the engine's dropout masks the `ns80_dropout_*` entry is computed with).  Line endings normalised; the file names
    are part of the digest."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(_ROOT, "oracle", "*.py"))) + [os.path.join(_ROOT, "tests", "rng_host.py")]
    for f in files:
        h.update(os.path.relpath(f, _ROOT).replace(os.sep, "/").encode() + b"\0")
        with open(f, "rb") as fh:
            h.update(fh.read().replace(b"\r\n", b"\n") + b"\0")
    return h.hexdigest()


def config_hash(config):
    import hashlib
    return hashlib.sha256(json.dumps(config, sort_keys=True, default=str).encode()).hexdigest()


def _flatten(depends_on):
    for t in depends_on:
        if isinstance(t, dict):  # a parameter dict: every tensor, in key order
            for k in sorted(t):
                yield t[k]
        else:
            yield t


def _fingerprint(tensors):
    """Identity of what an oracle result was computed FROM (inputs, weights): float64 sums of |x| and of x * index -- cheap, and any
    change of a seed, a shape or an initialiser changes it.  Entries of `tensors` may be parameter dicts (every tensor counts)."""
    acc = []
    for t in _flatten(tensors):
        t = t.detach().double().reshape(-1)
        acc += [float(t.abs().sum()), float((t * torch.arange(1, t.numel() + 1, dtype=torch.float64).remainder(97.0)).sum()), float(t.numel())]
    return acc


def cache_entry_valid(z, fp, config):
    """A disk entry is used only while ALL of these still hold: the fingerprint of the inputs and weights, the hash of the
    hyper-parameters / configuration the oracle ran with, and the hash of the oracle's own sources."""
    if "__code__" not in z.files or "__config__" not in z.files:
        return False
    if str(z["__code__"]) != oracle_code_hash() or str(z["__config__"]) != config_hash(config):
        return False
    return z["__fingerprint__"].shape == (len(fp),) and np.allclose(z["__fingerprint__"], np.array(fp), rtol=1e-12, atol=0.0)


def cached(key, fn, depends_on=None, config=None):
    """Memo for CPU-oracle results.  In the process: shared by the parametrisations of one test (bf16 / fp16 builds compare with the
    SAME fp32 oracle output).  On disk (tests/golden/oracle_cache/<key>.npz, written by tests/golden/make_oracle_cache.py in the build
    container): the full-size oracle runs are what the GPU suite spends its time on -- minutes of fp32 CPU rollouts per test -- so
    their OUTPUTS are kept as small fixtures next to (a) a fingerprint of the inputs and weights they were computed from
    (`depends_on`: tensors and whole parameter dicts), (b) a hash of `config` (hyper-parameters, model configuration, row lists) and
    (c) a hash of the oracle's sources (oracle/*.py, tests/rng_host.py).  An entry for which any of the three differs is ignored and
    the oracle runs.  The oracle itself stays pinned to the reference by the CPU suite; tests/test_oracle_cache.py recomputes rows of
    EVERY cached entry there."""
    if key in _ORACLE_CACHE:
        return _ORACLE_CACHE[key]
    path = os.path.join(_CACHE_DIR, key + ".npz")
    fp = None if depends_on is None else _fingerprint(depends_on)
    if fp is not None and os.path.exists(path) and os.environ.get("DYF_ORACLE_CACHE", "1") != "0":
        with np.load(path, allow_pickle=False) as z:
            if cache_entry_valid(z, fp, config):
                kind = str(z["__kind__"])
                if kind == "tensor":
                    val = torch.from_numpy(z["value"].copy())
                else:
                    val = {k[len("v::"):]: torch.from_numpy(z[k].copy()) for k in z.files if k.startswith("v::")}
                _ORACLE_CACHE[key] = val
                return val
    val = fn()
    _ORACLE_CACHE[key] = val
    if fp is not None and os.environ.get("DYF_WRITE_ORACLE_CACHE") == "1":
        os.makedirs(_CACHE_DIR, exist_ok=True)
        meta = dict(__fingerprint__=np.array(fp), __code__=oracle_code_hash(), __config__=config_hash(config))
        if torch.is_tensor(val):
            np.savez_compressed(path, __kind__="tensor", value=val.detach().cpu().numpy(), **meta)
        elif isinstance(val, dict) and all(torch.is_tensor(v) for v in val.values()):
            np.savez_compressed(path, __kind__="dict", **meta, **{"v::" + k: v.detach().cpu().numpy() for k, v in val.items()})
    return val
