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
