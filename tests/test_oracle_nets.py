"""Oracle backbones vs outputs of the imported reference (fixture G3) + the explicit bilinear formula."""
import json

import pytest
import torch
import torch.nn.functional as F

from oracle import nets
from tests.helpers import load_npz, max_abs, split_state

TOL = 2e-6  # fp32, same ATen primitives as the reference -> agreement to rounding


@pytest.mark.parametrize("name", ["net_unet_simple_a", "net_unet_simple_b", "net_unet_simple_c", "net_unet_simple_d", "net_unet_simple_e"])
def test_unet_simple_matches_reference(name):
    z = load_npz(name + ".npz")
    P = split_state(z, "P")
    cfg = json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    y = nets.unet_simple_forward(P, cfg, x, t, c)
    assert max_abs(y, z["y_eval"]) <= TOL * max(1.0, float(abs(z["y_eval"]).max()))
    y = nets.unet_simple_forward(P, cfg, x, t, c, dropout=nets.DropoutSeeded(int(z["dropout_seed"])))
    assert max_abs(y, z["y_drop"]) <= TOL * max(1.0, float(abs(z["y_drop"]).max()))


def test_dropout_record_and_replay():
    z = load_npz("net_unet_simple_b.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t, c = (torch.from_numpy(z[k]) for k in ("x", "t", "c"))
    src = nets.DropoutSeeded(int(z["dropout_seed"]), record=True)
    y1 = nets.unet_simple_forward(P, cfg, x, t, c, dropout=src)
    assert len(src.masks) == 12  # one per UNetBlock (input dropout has p = 0)
    y2 = nets.unet_simple_forward(P, cfg, x, t, c, dropout=nets.DropoutFromList(src.masks))
    assert torch.equal(y1, y2)


def test_simple_conv_net_matches_reference():
    z = load_npz("net_simple_conv.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    y = nets.simple_conv_net_forward(P, cfg, torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["c"]))
    assert max_abs(y, z["y_eval"]) <= TOL * max(1.0, float(abs(z["y_eval"]).max()))


@pytest.mark.parametrize("shape,out", [((2, 3, 23, 11), (32, 32)), ((1, 2, 64, 64), (23, 11)), ((1, 1, 221, 42), (256, 256)),
                                       ((1, 2, 512, 512), (221, 42)), ((2, 2, 4, 4), (8, 8)), ((1, 3, 7, 5), (14, 10))])
def test_explicit_bilinear_equals_aten(shape, out):
    x = torch.randn(*shape, generator=torch.Generator().manual_seed(3))
    want = F.interpolate(x, size=out, mode="bilinear")
    got = nets.bilinear_resize_explicit(x, *out)
    # source coordinates reach ~500 where one fp32 ulp is 3e-5: the lerp weight itself is only known to ~1e-4
    assert max_abs(got, want) <= (1e-4 if max(shape[2:]) > 256 else 1e-5)
    if out == (shape[2] * 2, shape[3] * 2):  # scale_factor=2 path used by the decoder blocks
        assert max_abs(got, F.interpolate(x, scale_factor=2, mode="bilinear")) <= 1e-5


def test_fullsize_forwards_match_reference_checksums():
    """Fixture G6: NS 221x42, dim 64 @256^2 -- one forecaster and one interpolator forward, seeded parameters."""
    import numpy as np
    from oracle import init as oinit
    from tests.helpers import jload, load_npz, rel_rms

    meta = jload("fullsize_checksums.json")
    fields = load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 3 + 2, 3), meta["seeds"]["forecaster"])
    PI = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 6 + 2, 3), meta["seeds"]["interpolator"])
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    yF = nets.unet_simple_forward(PF, mk, x0, torch.tensor([3.0]), c)
    yI = nets.unet_simple_forward(PI, mk, torch.cat([x0, yF], 1), torch.tensor([5.0]), c)
    assert rel_rms(yF, fields["yF"]) < 1e-5 and rel_rms(yI, fields["yI"]) < 1e-5
    for y, key in ((yF, "forecaster_fwd"), (yI, "interpolator_fwd")):
        assert float(y.mean()) == pytest.approx(meta[key]["mean"], abs=1e-5)
        got = [float(y[tuple(p)]) for p in meta["probes"]]
        assert np.allclose(got, meta[key]["probes"], atol=2e-5)


@pytest.mark.parametrize("name", ["net_unet_resnet_a", "net_unet_resnet_b", "net_unet_resnet_c", "net_unet_resnet_d", "net_unet_resnet_e",
                                  "net_unet_resnet_g"])
def test_resnet_unet_matches_reference(name):
    """src.models.unet.Unet (WS-conv, GroupNorm+SiLU, FiLM, LinearAttention, Attention, channel LayerNorm)."""
    z = load_npz(name + ".npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    y = nets.resnet_unet_forward(P, cfg, x, t, c)
    assert max_abs(y, z["y_eval"]) <= 1e-5 * max(1.0, float(abs(z["y_eval"]).max()))
    y = nets.resnet_unet_forward(P, cfg, x, t, c, dropout=nets.DropoutSeeded(int(z["dropout_seed"])))
    assert max_abs(y, z["y_drop"]) <= 1e-5 * max(1.0, float(abs(z["y_drop"]).max()))
