"""-m gpu: SimpleConvNet (spring-mesh backbone, SURVEY 8a row B7 / BASELINE configs[0]) on the HIP engine: single forwards
against the reference's golden outputs, a DYffusion rollout of a SimpleConvNet pair against the oracle sampler."""
import json

import pytest
import torch

import dyffusion_amd as D
from oracle import init as oinit
from oracle import nets, sampler
from tests.gpu_common import DEV, nhwc_masks
from tests.helpers import load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu
TOL = 1.5e-2


def _mirror(P, cfg, n_in, n_cond, n_out):
    net = D.SimpleConvNet(dim=cfg["dim"], with_time_emb=cfg.get("with_time_emb", True), kernel_sizes=cfg["kernel_sizes"],
                          dropout=cfg.get("dropout", 0.0), num_input_channels=n_in, num_output_channels=n_out,
                          num_conditional_channels=n_cond)
    net.load_state_dict(P, strict=True)
    return net


def test_forward_matches_reference_golden():
    z = load_npz("net_simple_conv.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    net = _mirror(P, cfg, x.shape[1], 0 if c is None else c.shape[1], z["y_eval"].shape[1])
    y = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    err = rel_rms(y, z["y_eval"])
    print("simple_conv eval rel-rms", err)
    assert err <= TOL
    # MC dropout: oracle with a seeded mask stream, the same masks injected into the engine
    src = nets.DropoutSeeded(4242, record=True)
    with torch.no_grad():
        y_or = nets.simple_conv_net_forward(P, cfg, x, t, c, dropout=src)
    y = net._engine.net_forward(0, x.to(DEV), t.to(DEV), None if c is None else c.to(DEV), dropout_mode=2,
                                masks=nhwc_masks(src.masks)).cpu()
    err = rel_rms(y, y_or)
    print("simple_conv dropout rel-rms", err)
    assert err <= TOL
    # engine RNG: different draws per forward, finite
    net.enable_inference_dropout()
    a = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    b = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    assert torch.isfinite(a).all() and not torch.equal(a, b)


def test_spring_mesh_rollout_matches_oracle():
    """BASELINE configs[0] shapes: (NB, 4, 10, 10) dynamics + 1 static channel, h = 4, SimpleConvNet dim 64 k = 9,7,5,3,
    cold sampling + refine, forward_conditioning 'data' (forecaster sees x0 as condition)."""
    C, Cs, h, nb = 4, 1, 4, 5
    cfg = dict(dim=64, kernel_sizes=[9, 7, 5, 3], with_time_emb=True, dropout=0.1, residual=True)
    hp = dict(timesteps=h, forward_conditioning="data", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, num_input_channels=C)
    # gain 0.8: with the default 1.4 these 9x9 / 7x7 stacks amplify their input ~3.5x per forward and a 4-step rollout
    # turns bf16 rounding (4-6e-3 per forward, printed by the test above) into 8e-2; the point here is the sampler plumbing
    PF = oinit.seeded_state(oinit.simple_conv_net_param_shapes(64, C + C + Cs, C, cfg["kernel_sizes"]), seed=7, gain=0.8)
    PI = oinit.seeded_state(oinit.simple_conv_net_param_shapes(64, 2 * C + Cs, C, cfg["kernel_sizes"]), seed=8, gain=0.8)
    F = _mirror(PF, cfg, C, C + Cs, C)
    I = _mirror(PI, cfg, 2 * C, Cs, C)
    m = D.DYffusion(F, D.InterpolatorHandle(I, h), timesteps=h, forward_conditioning="data", interpolate_before_t1=True,
                    refine_intermediate_predictions=True, enable_interpolator_dropout=False, max_batch=nb)
    g = torch.Generator().manual_seed(9)
    x0, c = torch.randn(nb, C, 10, 10, generator=g), torch.rand(nb, Cs, 10, 10, generator=g)
    got = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    with torch.no_grad():
        want = sampler.sample_loop(lambda x, t, cond: nets.simple_conv_net_forward(PF, cfg, x, t, cond),
                                   lambda x, t, cond: nets.simple_conv_net_forward(PI, cfg, x, t, cond), x0, c, hp)
    assert sorted(got) == sorted(want)
    worst = max(rel_rms(got[k].cpu(), want[k]) for k in want)
    print("spring-mesh rollout worst rel-rms", worst)
    assert worst <= 3e-2
    # graph replay == first (capturing) call
    again = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    for k in got:
        assert torch.equal(got[k], again[k])
