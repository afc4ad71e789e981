"""Shared helpers for the parity tests (fixture loading, error metrics)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def load_npz(name):
    z = np.load(os.path.join(GOLDEN, name), allow_pickle=False)
    return {k: z[k] for k in z.files}


def split_state(z, prefix):
    """Collect 'prefix::key' arrays of a fixture into a state dict of torch tensors."""
    tag = prefix + "::"
    return {k[len(tag):]: torch.from_numpy(np.asarray(v)) for k, v in z.items() if k.startswith(tag)}


def rel_rms(a, b):
    a = torch.as_tensor(a, dtype=torch.float64)
    b = torch.as_tensor(b, dtype=torch.float64)
    return float(((a - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt().clamp_min(1e-30))


def max_abs(a, b):
    return float((torch.as_tensor(a, dtype=torch.float64) - torch.as_tensor(b, dtype=torch.float64)).abs().max())


def jload(name):
    with open(os.path.join(GOLDEN, name)) as f:
        return json.load(f)


def boundary_case(name):
    """Rebuild the inputs of a boundary_*.npz fixture (tests/golden/make_golden.py::gen_boundary): returns
    (system, preds, targets, metadata, time, expected) with `expected` = the imported reference's output."""
    z = load_npz(name + ".npz")
    system, shape, B = str(z["system"]), tuple(int(v) for v in z["shape"]), int(z["B"])
    C, Hh, Ww = shape[-3:]
    g = torch.Generator().manual_seed(int(z["seed"]))
    preds = torch.randn(*shape, generator=g)
    fixed = torch.from_numpy(np.unpackbits(z["fixed_mask"])[: B * C * Hh * Ww].reshape(B, C, Hh, Ww).astype(bool))
    meta = {"fixed_mask": fixed}
    if system == "navier-stokes":
        vert = torch.zeros(B, 2, Hh, Ww)
        vert[:, 1] = torch.from_numpy(z["vertex_y"])[:, None, :].expand(B, Hh, Ww)
        meta.update(in_velocity=torch.from_numpy(z["in_velocity"]), vertices=vert)
    else:
        feat = torch.zeros(B, 5, 4, Hh, Ww)
        feat[:, 0, 2:] = torch.from_numpy(z["base_q"])
        meta["features"] = feat
    t = z["time"]
    time = float(t) if t.ndim == 0 else torch.from_numpy(t.astype(np.float32))
    expected = preds.clone().reshape(-1)
    expected[torch.from_numpy(z["changed_idx"])] = torch.from_numpy(z["changed_val"])
    return system, preds, torch.zeros(B, C, Hh, Ww), meta, time, expected.reshape(shape)


def apply_test_forms():
    """Child scripts of the parity tests: the kernel-form switches of the case arrive in DYF_TEST_FORMS ("KEY=VALUE;..."), are read
    HERE -- by test code -- and handed to the library through dyf_debug_set_form; the library itself reads nothing of the kind from
    the environment (include/dyffusion_hip_testing.h)."""
    from dyffusion_amd import _lib
    for item in filter(None, os.environ.get("DYF_TEST_FORMS", "").split(";")):
        key, _, value = item.partition("=")
        _lib.set_form(key, value)
