"""The committed golden fixtures ARE what the reference produces today: when /root/reference is present (the build container;
never the GPU box) regenerate the fast fixture families with tests/golden/make_golden.py into a scratch directory and compare
every array / JSON value with the committed file, bit for bit.  Guards the recipe itself (the LayerNorm gains of
net_unet_resnet_* were once seeded with Python's per-process salted `hash()` and could not be regenerated)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
GOLDEN = os.path.join(ROOT, "tests", "golden")

pytestmark = pytest.mark.skipif(not os.path.isdir("/root/reference/src"),
                                reason="the reference checkout is only present in the build container")


def _regen(tmp_path, families, hashseed):
    env = dict(os.environ, DYF_GOLDEN_OUT=str(tmp_path), TQDM_DISABLE="1", PYTHONHASHSEED=str(hashseed))
    r = subprocess.run([sys.executable, os.path.join(GOLDEN, "make_golden.py"), *families], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    made = sorted(os.listdir(tmp_path))
    assert made, "the generator wrote nothing"
    return made


def _same_npz(a, b):
    with np.load(a, allow_pickle=False) as x, np.load(b, allow_pickle=False) as y:
        assert sorted(x.files) == sorted(y.files), (os.path.basename(a), sorted(set(x.files) ^ set(y.files)))
        for k in x.files:
            xa, ya = x[k], y[k]
            assert xa.dtype == ya.dtype and xa.shape == ya.shape, (os.path.basename(a), k)
            if xa.dtype.kind in "US":  # embedded JSON (hyper-parameters / config)
                assert json.loads(str(xa)) == json.loads(str(ya)), (os.path.basename(a), k)
            else:
                assert np.array_equal(xa, ya, equal_nan=True), (os.path.basename(a), k, float(np.abs(xa - ya).max()))


@pytest.mark.parametrize("families,hashseed", [(("schedules", "samples"), 0), (("resnet",), 12345)])
def test_fixtures_regenerate_bit_identically(tmp_path, families, hashseed):
    # two different PYTHONHASHSEEDs across the parametrisations: nothing in the recipe may depend on str hashing
    for name in _regen(tmp_path, families, hashseed):
        new, old = os.path.join(tmp_path, name), os.path.join(GOLDEN, name)
        assert os.path.exists(old), f"{name} is generated but not committed"
        if name.endswith(".json"):
            with open(new) as f, open(old) as g:
                assert json.load(f) == json.load(g), name
        else:
            _same_npz(new, old)
