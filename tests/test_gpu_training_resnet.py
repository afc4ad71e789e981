"""-m gpu: the training step of the ResNet-UNet forecaster (arch unet.Unet, the OISST backbone) on the engine:
`DYffusion.p_losses` in training mode + `loss.backward()` -- recorded fp32 forward and backward of weight-standardised convs,
GroupNorm + FiLM + SiLU + Dropout, LinearAttention, Attention (dropout on the probabilities), channel LayerNorm, nearest x2
upsampling, the time MLP (csrc/train_resnet.inc), twice through the forecaster and THROUGH the frozen interpolator -- against
torch.autograd over the oracle, which tests/test_oracle_losses.py pins to the imported reference's own losses and gradients
(plosses_train_resnet_*.npz; reference: src/models/unet.py:26-109, 266-315, src/models/modules/attention.py:7-73,
src/diffusion/dyffusion.py:496-567).

The engine draws its dropout masks from its own generator; the oracle replays exactly those masks (tests/rng_host.py).
Tolerance (stated): fp32 end to end -- losses within 1e-4 relative, every parameter's gradient within 1e-3 of the global
gradient norm.
"""
import json

import pytest
import torch

import dyffusion_amd as D
from oracle import losses, nets
from tests import rng_host as R
from tests.gpu_common import DEV
from tests.helpers import load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu


def _mirror(P, cfg, n_in, n_cond, n_out):
    net = D.Unet(dim=cfg["dim"], dim_mults=cfg["dim_mults"], with_time_emb=True, block_dropout=cfg.get("block_dropout", 0.0),
                 block_dropout1=cfg.get("block_dropout1", 0.0), attn_dropout=cfg.get("attn_dropout", 0.0), num_input_channels=n_in,
                 num_output_channels=n_out, num_conditional_channels=n_cond)
    net.load_state_dict(P, strict=True)
    return net


def _build(z, hp, mk):
    PF, PI = split_state(z, "F"), split_state(z, "I")
    fc, ic = hp["forecaster_channels"], hp["interpolator_channels"]
    F_, I_ = _mirror(PF, mk, fc["inputs"], fc["cond"], 1), _mirror(PI, mk, ic["inputs"], ic["cond"], 1)
    keys = ["forward_conditioning", "schedule", "additional_interpolation_steps", "interpolate_before_t1", "time_encoding",
            "lambda_reconstruction", "lambda_reconstruction2", "loss_function", "enable_interpolator_dropout"]
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, hp["timesteps"]), timesteps=hp["timesteps"], max_batch=hp["B"],
                    **{k: hp[k] for k in keys if k in hp})
    return m, PF, PI


def _oracle_step(PF, PI, mk, hp, xt_last, cond, t, drop, noise_fn):
    cfg = dict(mk, resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    PFg = {k: v.clone().requires_grad_(True) for k, v in PF.items()}

    def f_fn(x, tt, c):
        drop.begin_forward()
        return nets.resnet_unet_forward(PFg, cfg, x, tt, c, dropout=drop)

    def i_fn(x, tt, c):
        drop.begin_forward()
        return nets.resnet_unet_forward(PI, cfg, x, tt, c, dropout=drop)

    out = losses.p_losses(f_fn, i_fn, xt_last, cond, t, None, hp, noise_fn=noise_fn)
    out["loss"].backward()
    return out, {k: v.grad for k, v in PFg.items()}


@pytest.mark.parametrize("dropout", [False, True], ids=["no-dropout", "engine-dropout"])
@pytest.mark.parametrize("name", ["plosses_train_resnet_a", "plosses_train_resnet_b"])
def test_resnet_training_step_matches_autograd_of_the_oracle(name, dropout, monkeypatch):
    z = load_npz(name + ".npz")
    hp = json.loads(str(z["hp"]))
    mk = dict(hp["model"])
    if not dropout:
        mk.update(block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    m, PF, PI = _build(z, hp, mk)
    xt_last, cond, t = (torch.from_numpy(z[k]) for k in ("xt_last", "cond", "t"))
    # forward_conditioning="data+noise": the module draws torch.randn_like on the GPU; both sides get the same seeded draws
    gen = torch.Generator().manual_seed(5)
    draws = [torch.randn(cond.shape, generator=gen) for _ in range(4)]
    calls = {"n": 0}
    orig = torch.randn_like

    def fake(tensor, **kw):
        d = draws[calls["n"]][: tensor.shape[0]].to(tensor.device)
        calls["n"] += 1
        return d if d.shape == tensor.shape else orig(tensor, **kw)

    monkeypatch.setattr(torch, "randn_like", fake)
    seed = 20260929
    m.seed(seed)
    m.train()
    out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=None)
    out["loss"].backward()
    n_engine_draws = calls["n"]
    monkeypatch.undo()
    hw_mid = (xt_last.shape[-2] >> (len(mk["dim_mults"]) - 1)) * (xt_last.shape[-1] >> (len(mk["dim_mults"]) - 1))
    drop = R.ResnetEngineDropout(seed, hw_mid) if dropout else nets.DropoutOff()
    if not dropout:
        drop.begin_forward = lambda: None
    it = iter(draws)
    want, grads = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, drop, lambda x: next(it)[: x.shape[0]])
    assert hp["forward_conditioning"] != "data+noise" or n_engine_draws == 2
    for k_got, k_want in (("loss", "loss"), ("train/loss_forward", "loss_forward"), ("train/loss_forward2", "loss_forward2")):
        assert float(out[k_got]) == pytest.approx(float(want[k_want]), rel=1e-4), k_got
    got = {k: p.grad for k, p in m.model.named_parameters()}
    assert sorted(got) == sorted(grads)
    gn = float(torch.cat([g.reshape(-1) for g in grads.values()]).norm())
    errs = {k: float((got[k].cpu() - grads[k]).norm()) / gn for k in grads}
    worst = max(errs, key=errs.get)
    print(f"{name} dropout={dropout}: loss {float(out['loss']):.6f}, grad norm {gn:.4f}, worst per-tensor gradient error / grad norm = "
          f"{errs[worst]:.2e} ({worst})")
    assert errs[worst] <= 1e-3
    m.eval()


def test_resnet_sgd_steps_reduce_the_loss_and_sampling_sees_the_new_weights():
    z = load_npz("plosses_train_resnet_a.npz")
    hp = json.loads(str(z["hp"]))
    mk = dict(hp["model"], block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    m, _, _ = _build(z, hp, mk)
    xt_last, cond, t = (torch.from_numpy(z[k]).to(DEV) for k in ("xt_last", "cond", "t"))
    before = {k: v.clone() for k, v in m.sample(cond).items()}
    m.train()
    opt = torch.optim.SGD(m.model.parameters(), lr=0.02)
    hist = []
    for _ in range(4):
        opt.zero_grad()
        out = m.p_losses(xt_last, cond, t, static_condition=None)
        out["loss"].backward()
        opt.step()
        hist.append(float(out["loss"]))
    print("ResNet-UNet loss over 4 SGD steps:", [round(v, 5) for v in hist])
    assert hist[-1] < hist[0]
    m.eval()
    after = m.sample(cond)
    assert all(bool(torch.isfinite(v).all()) for v in after.values())
    assert any(not torch.equal(after[k], before[k]) for k in after)  # the sampling copy was re-derived from the updated weights


def test_resnet_interpolator_stage1_get_loss_trains():
    """Stage 1 (`BaseModel.get_loss` in train mode, _base_model.py:108-138) for a ResNet-UNet interpolator: loss and gradients against
    autograd over the oracle forward (dropout off)."""
    z = load_npz("plosses_train_resnet_a.npz")
    hp = json.loads(str(z["hp"]))
    mk = dict(hp["model"], block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    PI = split_state(z, "I")
    net = _mirror(PI, mk, 2, 0, 1)
    g = torch.Generator().manual_seed(9)
    x = torch.randn(3, 2, *z["cond"].shape[-2:], generator=g)
    y = torch.randn(3, 1, *z["cond"].shape[-2:], generator=g)
    tt = torch.tensor([1.0, 2.0, 3.0])
    net.train()
    loss = net.get_loss(x.to(DEV), y.to(DEV), time=tt.to(DEV))
    loss.backward()
    cfg = dict(mk, resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    Pg = {k: v.clone().requires_grad_(True) for k, v in PI.items()}
    want = ((nets.resnet_unet_forward(Pg, cfg, x, tt, None) - y) ** 2).mean()
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    gn = float(torch.cat([v.grad.reshape(-1) for v in Pg.values()]).norm())
    worst = max(float((p.grad.cpu() - Pg[k].grad).norm()) for k, p in net.named_parameters()) / gn
    print(f"stage-1 ResNet-UNet get_loss: loss {float(loss):.6f}, worst gradient error / grad norm {worst:.2e}")
    assert worst <= 1e-3


def test_dim64_resnet_training_step_on_the_matrix_core_convs():
    """dim 64, mults (1, 2) on a 16 x 16 grid: every conv with >= 64 channels runs on the fp32 matrix cores (train_gemm.hip:
    forward, dgrad, wgrad incl. the 1x1 and 4x4 / stride-2 forms); both loss terms, no dropout; against autograd over the oracle."""
    from tests.test_gpu_unet_resnet import seeded_unet
    mk = dict(dim=64, dim_mults=[1, 2], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    PF, PI = seeded_unet(64, (1, 2), 2, 1, seed=91), seeded_unet(64, (1, 2), 2, 1, seed=92)
    hp = dict(timesteps=4, forward_conditioning="data", schedule="before_t1_only", additional_interpolation_steps=0,
              interpolate_before_t1=True, time_encoding="dynamics", lambda_reconstruction=1.0, lambda_reconstruction2=0.5,
              loss_function="l1", enable_interpolator_dropout=True, B=3)
    F_, I_ = _mirror(PF, mk, 1, 1, 1), _mirror(PI, mk, 2, 0, 1)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 4), timesteps=4, max_batch=3,
                    **{k: hp[k] for k in ("forward_conditioning", "schedule", "interpolate_before_t1", "time_encoding",
                                          "lambda_reconstruction", "lambda_reconstruction2", "loss_function")})
    g = torch.Generator().manual_seed(17)
    xt_last, cond, t = torch.randn(3, 1, 16, 16, generator=g), torch.randn(3, 1, 16, 16, generator=g), torch.tensor([0, 2, 3])
    m.train()
    out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=None)
    out["loss"].backward()
    drop = nets.DropoutOff()
    drop.begin_forward = lambda: None
    want, grads = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, drop, None)
    assert float(out["loss"]) == pytest.approx(float(want["loss"]), rel=1e-4)
    got = {k: p.grad for k, p in m.model.named_parameters()}
    gn = float(torch.cat([v.reshape(-1) for v in grads.values()]).norm())
    errs = {k: float((got[k].cpu() - grads[k]).norm()) / gn for k in grads}
    worst = max(errs, key=errs.get)
    print(f"dim-64 ResNet-UNet: loss {float(out['loss']):.6f}, grad norm {gn:.4f}, worst gradient error / grad norm {errs[worst]:.2e} ({worst})")
    assert errs[worst] <= 1e-3
    m.eval()


def test_resnet_training_step_with_input_dropout_matches_autograd_of_the_oracle():
    """input_dropout > 0 (unet.py:162-163, 276-277: dropout_input_for_residual and dropout_input on init_conv's output, the first two
    sites of a forward) in TRAIN mode, on the network of the reference golden net_unet_resnet_c: stage-1 `get_loss` + backward on the
    engine against torch.autograd over the oracle with the engine's masks."""
    from tests.test_gpu_unet_resnet import mirror
    z = load_npz("net_unet_resnet_c.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    assert cfg["input_dropout"] > 0
    x, t, c = torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["c"])
    net = mirror(P, cfg, x.shape[1], c.shape[1], 1)
    net.train()
    y = torch.randn(x.shape[0], 1, *x.shape[-2:], generator=torch.Generator().manual_seed(12))
    seed = 777001
    net._own_engine(x.shape[0], x.shape[-2:]).seed(seed)
    loss = net.get_loss(x.to(DEV), y.to(DEV), condition=c.to(DEV), time=t.to(DEV))
    loss.backward()
    nlev = len(cfg["dim_mults"])
    hw_mid = (x.shape[-2] >> (nlev - 1)) * (x.shape[-1] >> (nlev - 1))
    drop = R.ResnetEngineDropout(seed, hw_mid)
    drop.begin_forward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    want = ((nets.resnet_unet_forward(Pg, cfg, x, t, c, dropout=drop) - y) ** 2).mean()
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    gn = float(torch.cat([v.grad.reshape(-1) for v in Pg.values()]).norm())
    worst = max(float((p.grad.cpu() - Pg[k].grad).norm()) for k, p in net.named_parameters()) / gn
    print(f"ResNet-UNet input_dropout training step: loss {float(loss):.6f}, worst gradient error / grad norm {worst:.2e}")
    assert worst <= 1e-3


def test_resnet_sampling_forward_with_input_dropout_draws_the_same_streams():
    """The SAMPLING path (16-bit activations, `drop16_kernel`) with the engine's generator: sites 0 / 1 of a forward are the residual
    copy's and the input's dropout, as in the training step -- forward with MC dropout on vs the oracle on host-rebuilt masks
    (attention dropout off: its probability mask is covered by the injected-mask golden test)."""
    from tests.test_gpu_unet_resnet import mirror
    z = load_npz("net_unet_resnet_c.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    cfg = dict(cfg, attn_dropout=0.0)
    x, t, c = torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["c"])
    net = mirror(P, cfg, x.shape[1], c.shape[1], 1)
    seed = 31415
    with net.inference_dropout_scope(True):
        net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV))
        net._engine.seed(seed)
        y = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    nlev = len(cfg["dim_mults"])
    drop = R.ResnetEngineDropout(seed, (x.shape[-2] >> (nlev - 1)) * (x.shape[-1] >> (nlev - 1)))
    drop.begin_forward()
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, c, dropout=drop)
        off = nets.resnet_unet_forward(P, cfg, x, t, c)
    err = rel_rms(y, want)
    print(f"ResNet-UNet sampling forward, input_dropout on, engine streams vs host masks: rel-rms {err:.3e} (dropout effect {rel_rms(off, want):.2f})")
    assert err <= 4e-3 and rel_rms(off, want) > 0.05


@pytest.mark.parametrize("name", ["net_unet_resnet_d", "net_unet_resnet_e", "net_unet_resnet_g"],
                         ids=["keep_spatial_dims", "single_conv_layer", "learned_sinusoidal_cond"])
def test_resnet_training_step_with_unet_options_matches_autograd_of_the_oracle(name):
    """The unet.Unet options no shipped config sets (unet.py:127-135), in TRAIN mode on the networks of the reference goldens:
    keep_spatial_dims (plain 3x3 convs between the levels), double_conv_layer=False (block2 = Identity), learned_sinusoidal_cond
    (the frequencies `time_emb_mlp.0.weights` are a parameter and get a gradient).  Stage-1 `get_loss` + backward vs torch.autograd
    over the oracle with the engine's masks."""
    from tests.test_gpu_unet_resnet import mirror
    z = load_npz(name + ".npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    n_out = z["y_eval"].shape[1]
    net = mirror(P, cfg, x.shape[1], 0 if c is None else c.shape[1], n_out)
    net.train()
    y = torch.randn(x.shape[0], n_out, *x.shape[-2:], generator=torch.Generator().manual_seed(21))
    seed = 90210
    net._own_engine(x.shape[0], x.shape[-2:]).seed(seed)
    loss = net.get_loss(x.to(DEV), y.to(DEV), condition=None if c is None else c.to(DEV), time=t.to(DEV))
    loss.backward()
    nlev = len(cfg["dim_mults"])
    shrink = 0 if cfg.get("keep_spatial_dims") else nlev - 1
    drop = R.ResnetEngineDropout(seed, (x.shape[-2] >> shrink) * (x.shape[-1] >> shrink))
    drop.begin_forward()
    Pg = {k: v.clone().requires_grad_(True) for k, v in P.items()}
    want = ((nets.resnet_unet_forward(Pg, cfg, x, t, c, dropout=drop) - y) ** 2).mean()
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    gn = float(torch.cat([v.grad.reshape(-1) for v in Pg.values()]).norm())
    got = dict(net.named_parameters())
    assert sorted(got) == sorted(Pg)
    worst = max(float((p.grad.cpu() - Pg[k].grad).norm()) for k, p in got.items()) / gn
    print(f"{name}: loss {float(loss):.6f}, worst gradient error / grad norm {worst:.2e}")
    assert worst <= 1e-3
    if "time_emb_mlp.0.weights" in Pg:
        gw, ww = got["time_emb_mlp.0.weights"].grad.cpu(), Pg["time_emb_mlp.0.weights"].grad
        assert float((gw - ww).norm()) <= 1e-3 * float(ww.norm()) + 1e-9 and float(ww.norm()) > 0
