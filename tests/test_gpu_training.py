"""-m gpu: the training step of the forecaster objective on the engine: `DYffusion.p_losses` in training
mode + `loss.backward()` -- forward with batch-statistics BatchNorm and Dropout, backward through the forecaster (twice) and
THROUGH the frozen interpolator (second loss term) -- against torch.autograd over the oracle, which
tests/test_oracle_losses.py pins to the imported reference's own losses and gradients (plosses_train_*.npz).

The engine draws its dropout masks from its own generator; the oracle replays exactly those masks (tests/rng_host.py):
forward counter 0 = first interpolator call, 1 = first forecaster pass, 2 / 3 = the second pair.
Tolerance (stated): the training path is fp32 end to end -- losses within 1e-4 relative, every parameter's gradient within
1e-3 of the global gradient norm (measured ~1e-5; the reductions run in a different order than ATen's).
"""
import json

import pytest
import torch
import torch.nn.functional as F

from oracle import losses, nets
from tests import rng_host as R
from tests.gpu_common import DEV, build_dyffusion
from tests.helpers import load_npz, split_state

pytestmark = pytest.mark.gpu


def _oracle_step(PF, PI, mk, hp, xt_last, cond, t, sc, seed):
    uh, uw = mk["upsample_dims"]
    drop = R.EngineDropout(seed, mk["dim"], uh, uw)
    PFg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
           for k, v in PF.items()}
    bn_batches = []
    orig = F.batch_norm

    def spy(x, rm, rv, weight=None, bias=None, training=False, momentum=0.1, eps=1e-5):
        if training:
            bn_batches.append((x.detach().mean((0, 2, 3)), x.detach().var((0, 2, 3), unbiased=True)))
        return orig(x, rm, rv, weight, bias, training, momentum, eps)

    def f_fn(x, tt, c):
        drop.begin_forward()
        return nets.unet_simple_forward(PFg, mk, x, tt, c, dropout=drop, bn_training=True)

    def i_fn(x, tt, c):
        drop.begin_forward()
        return nets.unet_simple_forward(PI, mk, x, tt, c, dropout=drop)

    F.batch_norm = spy
    try:
        out = losses.p_losses(f_fn, i_fn, xt_last, cond, t, sc, hp)
    finally:
        F.batch_norm = orig
    out["loss"].backward()
    return out, {k: v.grad for k, v in PFg.items() if torch.is_tensor(v) and v.requires_grad}, bn_batches


@pytest.mark.parametrize("name,mode", [("plosses_train_a", "bilinear"), ("plosses_train_b", "bilinear"),
                                       ("plosses_train_a", "nearest")])
def test_training_step_matches_autograd_of_the_oracle(name, mode):
    """`mode`: the outer_sample_mode of both networks (the nearest case reuses the fixture's weights and inputs; the oracle's
    nearest forward is pinned by net_unet_simple_d.npz, its gradients by torch.autograd)."""
    z = load_npz(name + ".npz")
    hp = json.loads(str(z["hp"]))
    mk = dict(hp["model"], outer_sample_mode=mode)
    PF, PI = split_state(z, "F"), split_state(z, "I")
    xt_last, cond, sc, t = (torch.from_numpy(z[k]) for k in ("xt_last", "cond", "sc", "t"))
    m = build_dyffusion(PF, PI, mk, 4, 1, hp, max_batch=hp["B"])
    seed = 20240928
    m.seed(seed)
    m.train()
    out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=sc.to(DEV))
    assert set(out) == {"loss", "train/loss_forward", "train/loss_forward2"}
    out["loss"].backward()
    want, grads, bn_batches = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, sc, seed)
    for k_got, k_want in (("loss", "loss"), ("train/loss_forward", "loss_forward"), ("train/loss_forward2", "loss_forward2")):
        assert float(out[k_got]) == pytest.approx(float(want[k_want]), rel=1e-4), k_got
    got = {k: p.grad for k, p in m.model.named_parameters()}
    assert sorted(got) == sorted(grads)
    gn = float(torch.cat([g.reshape(-1) for g in grads.values()]).norm())
    worst = max(float((got[k].cpu() - grads[k]).norm()) for k in grads) / gn
    print(f"{name}: loss {float(out['loss']):.6f}, grad norm {gn:.4f}, worst per-tensor gradient error / grad norm = {worst:.2e}")
    assert worst <= 1e-3
    # the reference's own gradients (different dropout masks) have the same scale: a sanity anchor, not a parity check
    G = split_state(z, "G")
    assert mode != "bilinear" or 0.5 <= gn / float(torch.cat([g.reshape(-1) for g in G.values()]).norm()) <= 2.0
    # BatchNorm buffers as module.train() leaves them: two forecaster passes -> two momentum-0.1 updates
    sd = m.model.state_dict()
    n_pass = 2 if hp["lambda_reconstruction2"] > 0 else 1
    assert int(sd["input_ops.0.ops.1.num_batches_tracked"]) == n_pass
    rm, rv = PF["input_ops.0.ops.1.running_mean"].clone(), PF["input_ops.0.ops.1.running_var"].clone()
    per_pass = len(bn_batches) // n_pass
    for p in range(n_pass):
        bm, bv = bn_batches[p * per_pass]
        rm, rv = 0.9 * rm + 0.1 * bm, 0.9 * rv + 0.1 * bv
    assert torch.allclose(sd["input_ops.0.ops.1.running_mean"].cpu(), rm, rtol=1e-4, atol=1e-6)
    assert torch.allclose(sd["input_ops.0.ops.1.running_var"].cpu(), rv, rtol=1e-4, atol=1e-6)
    m.eval()


def test_sgd_steps_reduce_the_loss_and_resync_the_engine():
    """torch.optim over the mirror's parameters: the engine re-uploads weights that an optimizer modified in place."""
    z = load_npz("plosses_train_a.npz")
    hp = json.loads(str(z["hp"]))
    hp["model"] = dict(hp["model"], dropout=0.0)  # deterministic objective
    PF, PI = split_state(z, "F"), split_state(z, "I")
    xt_last, cond, sc, t = (torch.from_numpy(z[k]).to(DEV) for k in ("xt_last", "cond", "sc", "t"))
    m = build_dyffusion(PF, PI, hp["model"], 4, 1, hp, max_batch=hp["B"])
    m.train()
    opt = torch.optim.SGD(m.model.parameters(), lr=0.05)
    hist = []
    for _ in range(4):
        opt.zero_grad()
        out = m.p_losses(xt_last, cond, t, static_condition=sc)
        out["loss"].backward()
        opt.step()
        hist.append(float(out["loss"]))
    print("loss over 4 SGD steps:", [round(v, 5) for v in hist])
    assert hist[-1] < hist[0]


@pytest.mark.parametrize("name", ["interp_train_a", "interp_train_b"])
def test_interpolator_training_step_matches_autograd_of_the_oracle(name, monkeypatch):
    """Stage 1 (`InterpolationExperiment.get_loss` in train mode + backward, interpolation.py:149-167) on the engine against
    torch.autograd over `oracle.losses.interpolation_loss`, which tests/test_oracle_losses.py pins to the imported reference's
    loss and gradients (interp_train_*.npz).  The oracle replays the engine's dropout masks (forward counter 0)."""
    import dyffusion_amd as D
    from tests.gpu_common import mirror_from_params
    z = load_npz(name + ".npz")
    hp = json.loads(str(z["hp"]))
    mk, P = hp["model"], split_state(z, "P")
    dyn, cond, t = torch.from_numpy(z["dynamics"]), torch.from_numpy(z["cond"]), torch.from_numpy(z["t"])
    B, C = dyn.shape[0], dyn.shape[2]
    net = mirror_from_params(P, mk, (hp["window"] + 1) * C, cond.shape[1], C)
    net.hparams.loss_function = hp["loss_function"]
    exp = D.InterpolationExperiment(net, horizon=hp["horizon"], window=hp["window"])
    exp.train()
    seed = 777
    net._own_engine(B, dyn.shape[-2:]).seed(seed)
    idx = torch.tensor(hp["randint"])
    monkeypatch.setattr(torch, "randint", lambda *a, **k: idx.to(k.get("device", "cpu")))
    loss = exp.get_loss(dict(dynamics=dyn.to(DEV), condition=cond.to(DEV)))
    monkeypatch.undo()
    loss.backward()
    # oracle with the engine's masks
    uh, uw = mk["upsample_dims"]
    drop = R.EngineDropout(seed, mk["dim"], uh, uw)
    drop.begin_forward()
    Pg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
          for k, v in P.items()}
    want = losses.interpolation_loss(lambda x, tt, c: nets.unet_simple_forward(Pg, mk, x, tt, c, dropout=drop, bn_training=True),
                                     dyn, t, cond, hp["window"], hp["loss_function"])
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    grads = {k: v.grad for k, v in Pg.items() if torch.is_tensor(v) and v.requires_grad}
    got = {k: p.grad for k, p in net.named_parameters()}
    assert sorted(got) == sorted(grads)
    gn = float(torch.cat([g.reshape(-1) for g in grads.values()]).norm())
    worst = max(float((got[k].cpu() - grads[k]).norm()) for k in grads) / gn
    print(f"{name}: loss {float(loss):.6f}, grad norm {gn:.4f}, worst per-tensor gradient error / grad norm = {worst:.2e}")
    assert worst <= 1e-3
    G = split_state(z, "G")  # the reference's own gradients (other masks): same scale
    assert 0.5 <= gn / float(torch.cat([g.reshape(-1) for g in G.values()]).norm()) <= 2.0
    assert int(net.state_dict()["input_ops.0.ops.1.num_batches_tracked"]) == int(P["input_ops.0.ops.1.num_batches_tracked"]) + 1
    # an optimizer step on the module's parameters reaches the engine: the loss of the same batch changes
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    opt.step()
    net._own_engine(B, dyn.shape[-2:]).seed(seed)
    monkeypatch.setattr(torch, "randint", lambda *a, **k: idx.to(k.get("device", "cpu")))
    loss2 = exp.get_loss(dict(dynamics=dyn.to(DEV), condition=cond.to(DEV)))
    monkeypatch.undo()
    assert float(loss2) < float(loss)
    # eval mode: plain forward + criterion (no tape)
    exp.eval()
    le = net.get_loss(exp.get_inputs_from_dynamics(dyn.to(DEV)), dyn[torch.arange(B), hp["window"] + t - 1].to(DEV), time=t.to(DEV),
                      condition=cond.to(DEV))
    assert le.requires_grad is False and float(le) > 0


def test_training_step_at_dim64_on_the_matrix_cores_matches_autograd_of_the_oracle():
    """The fp32 MFMA implicit-GEMM convolutions (csrc/train_gemm.hip: forward, dgrad, wgrad, split-K on the small planes) only
    take layers with >= 64 channels: a dim-64 pair on 23 x 11 fields / 128 x 128 backbone grid, both loss terms, against
    torch.autograd over the oracle with the engine's own masks -- same tolerances as the dim-4 fixtures.  Measured 4.8e-4 of the
    gradient norm with the split-K convolutions, 4.5e-6 with DYF_TRAIN_SPLITK=0 or the VALU kernels: a different (fixed)
    summation order on the small planes moves one pre-activation across zero, i.e. one (Leaky)ReLU derivative differs from the
    oracle's -- the kernels themselves agree with the plain ones to 2e-6 (test_matrix_core_training_convs_match_the_plain_kernels).
    (At a 64 x 64 grid the batch statistics of the 2 x 2 plane come from 12 values and amplify fp32 rounding further: 1.2e-3.)"""
    from tests.gpu_common import seeded_pair
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[128, 128], with_time_emb=True, input_dropout=0.0, dropout=0.15)
    hp = dict(timesteps=4, schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
              interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none", lambda_reconstruction=1.0,
              lambda_reconstruction2=0.5, loss_function="l1", enable_interpolator_dropout=True, model=mk)
    C, Cs, B = 3, 2, 3
    PF, PI = seeded_pair(64, C, Cs)
    g = torch.Generator().manual_seed(3)
    xt_last, cond = torch.randn(B, C, 23, 11, generator=g), torch.randn(B, C, 23, 11, generator=g)
    sc, t = torch.rand(B, Cs, 23, 11, generator=g), torch.tensor([0, 2, 3])
    m = build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=B)
    seed = 4242
    m.seed(seed)
    m.train()
    out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=sc.to(DEV))
    out["loss"].backward()
    want, grads, _ = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, sc, seed)
    for k_got, k_want in (("loss", "loss"), ("train/loss_forward", "loss_forward"), ("train/loss_forward2", "loss_forward2")):
        assert float(out[k_got]) == pytest.approx(float(want[k_want]), rel=1e-4), k_got
    got = {k: p.grad for k, p in m.model.named_parameters()}
    gn = float(torch.cat([g_.reshape(-1) for g_ in grads.values()]).norm())
    errs = {k: float((got[k].cpu() - grads[k]).norm()) / gn for k in grads}
    worst = max(errs, key=errs.get)
    print(f"dim 64: loss {float(out['loss']):.6f}, grad norm {gn:.4f}, worst per-tensor gradient error / grad norm = {errs[worst]:.2e} ({worst})")
    assert errs[worst] <= 1e-3
    m.eval()


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_training_step_with_16bit_conv_operands_matches_the_operand_rounding_model(dtype):
    """`train_precision=16` ("bf16-mixed"): the training convolutions round their operands to bf16 while staging them
    (csrc/train_halo16.hip for the 3 x 3 / stride-1 layers, csrc/train_gemm.hip t_gemm_mfma16 for the rest; fp32 tensors, master
    weights, accumulation and statistics; bf16 on the fp16 engine too: csrc/train_internal.h).  The check is against an ORACLE-SIDE
    MODEL of that arithmetic -- `oracle.losses.training_operand_rounding`: torch.autograd over the reference-pinned restatement with
    every conv's forward / data-gradient / weight-gradient operands rounded to bf16 where the engine's dispatch rounds them -- with
    the engine's own dropout masks, on the dim-64 pair of the fp32 test above (128 x 128 backbone grid, both loss terms, L1).
    Until round 5 this mode was only compared with the engine's own fp32 step (per-parameter ||dg|| / ||g|| <= 0.35).
    What can be asserted: this objective (L1 behind (Leaky)ReLUs, MC dropout, batch statistics over planes down to 2 x 2) is CHAOTIC
    under bf16 operand rounding -- measured on the model itself (CPU): a 1e-6 relative jitter of the operands BEFORE rounding (what a
    different fp32 summation order amounts to) moves the model's whole gradient by 0.129 of its norm, as far as the rounding moves it
    from the fp32 oracle (0.135); the fp32 objective under the same jitter moves 2e-3.  So engine and model are two members of one
    cloud around the fp32 gradient, and the test is that they are members of the SAME cloud: the engine's distance from the fp32
    oracle's gradient is that of the model -- whole gradient within [0.7, 1.4] x (cloud members measured: 1.02 x), every parameter
    within 2.2 x (measured spread 0.54 .. 1.59 x) -- and the losses agree to 1e-3.  A rounding applied to a wrong or an extra operand,
    a dropped scale or a wrong tap moves the gradient out of that band (the taps themselves are held to 2e-5 on bf16-representable
    data by test_16bit_halo_training_convs_match_the_plain_kernels)."""
    from tests.gpu_common import seeded_pair
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[128, 128], with_time_emb=True, input_dropout=0.0, dropout=0.15)
    hp = dict(timesteps=4, schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
              interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none", lambda_reconstruction=1.0,
              lambda_reconstruction2=0.5, loss_function="l1", enable_interpolator_dropout=True, model=mk)
    C, Cs, B = 3, 2, 4
    PF, PI = seeded_pair(64, C, Cs)
    g = torch.Generator().manual_seed(3)
    xt_last, cond = torch.randn(B, C, 23, 11, generator=g), torch.randn(B, C, 23, 11, generator=g)
    sc, t = torch.rand(B, Cs, 23, 11, generator=g), torch.tensor([0, 2, 3, 1])
    seed = 4242
    m = build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=B, dtype=dtype, train_precision="bf16-mixed")
    m.seed(seed)
    m.train()
    m._ensure_engine((23, 11), B)
    eng = m._engine
    eng.form_log(True)
    out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=sc.to(DEV))
    out["loss"].backward()
    forms = eng.form_log_read()
    eng.form_log(False)
    assert any(k.startswith("t_halo3x3_16") for k in forms), sorted(forms)  # the 16-bit kernels did run
    got = {k: p.grad.detach().cpu().clone() for k, p in m.model.named_parameters()}
    with losses.training_operand_rounding():
        want, model, _ = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, sc, seed)
    want32, g32, _ = _oracle_step(PF, PI, mk, hp, xt_last, cond, t, sc, seed)
    cat = lambda d: torch.cat([d[k].reshape(-1) for k in sorted(model)])
    gn = float(cat(model).norm())
    errs = {k: float((got[k] - model[k]).norm()) / gn for k in model}
    worst = max(errs, key=errs.get)
    e_model, e_round = float((cat(got) - cat(model)).norm()) / gn, float((cat(model) - cat(g32)).norm()) / gn
    print(f"{dtype} engine, bf16-mixed step: loss {float(out['loss']):.6f} (model {float(want['loss']):.6f}, fp32 oracle {float(want32['loss']):.6f}); "
          f"engine vs model: whole gradient {e_model:.2e}, worst tensor {errs[worst]:.2e} of the gradient norm ({worst}); "
          f"model vs fp32 oracle (the rounding itself): {e_round:.2e}")
    for k_got, k_want in (("loss", "loss"), ("train/loss_forward", "loss_forward"), ("train/loss_forward2", "loss_forward2")):
        assert float(out[k_got]) == pytest.approx(float(want[k_want]), rel=1e-3), k_got
    d_eng = {k: float((got[k] - g32[k]).norm()) for k in model}
    d_mod = {k: float((model[k] - g32[k]).norm()) for k in model}
    big = [k for k in model if float(g32[k].norm()) > 1e-5 * gn]
    ratios = {k: d_eng[k] / max(d_mod[k], 1e-30) for k in big}
    hi_k, lo_k = max(ratios, key=ratios.get), min(ratios, key=ratios.get)
    e_eng = float((cat(got) - cat(g32)).norm()) / gn
    print(f"distance from the fp32 oracle: engine {e_eng:.3e}, model {e_round:.3e} (ratio {e_eng / e_round:.2f}); per parameter "
          f"engine / model: {ratios[lo_k]:.2f} ({lo_k}) .. {ratios[hi_k]:.2f} ({hi_k}) over {len(big)} tensors")
    assert e_round > 1e-3                                   # the rounding is visible at all
    assert 0.7 <= e_eng / e_round <= 1.4                    # same cloud: the engine rounds what the model rounds
    assert ratios[hi_k] <= 2.2
    m.eval()
    eng.close()


def test_train_precision_knob_selects_the_operand_format(form_switch):
    """`DYffusion(train_precision=...)` / `HipEngine.train_set_precision` -> C-ABI `dyf_train_set_precision` (the reference's Lightning
    `trainer.precision`): 16 / "16-mixed" runs the 16-bit-operand kernels without any environment variable, 32 keeps fp32 operands even
    when DYF_TRAIN_OPERANDS asks for 16 bits, None leaves the choice to the variable; bad values are refused."""
    from tests.gpu_common import seeded_pair
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[128, 128], with_time_emb=True, input_dropout=0.0, dropout=0.0)
    hp = dict(timesteps=4, schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
              interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none", lambda_reconstruction=1.0,
              lambda_reconstruction2=0.5, loss_function="l1", enable_interpolator_dropout=True, model=mk)
    C, Cs, B = 3, 2, 4
    PF, PI = seeded_pair(64, C, Cs)
    g = torch.Generator().manual_seed(3)
    xt_last, cond = torch.randn(B, C, 23, 11, generator=g), torch.randn(B, C, 23, 11, generator=g)
    sc, t = torch.rand(B, Cs, 23, 11, generator=g), torch.tensor([0, 2, 3, 1])

    def step(precision, env):
        if env:
            form_switch.setenv("DYF_TRAIN_OPERANDS", env)
        else:
            form_switch.delenv("DYF_TRAIN_OPERANDS", raising=False)
        m = build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=B, train_precision=precision)
        m.seed(7)
        m.train()
        m._ensure_engine((23, 11), B)
        eng = m._engine
        eng.form_log(True)
        out = m.p_losses(xt_last.to(DEV), cond.to(DEV), t.to(DEV), static_condition=sc.to(DEV))
        out["loss"].backward()
        forms = eng.form_log_read()
        eng.form_log(False)
        bits = eng.train_precision
        loss = float(out["loss"].detach())
        m.eval()
        eng.close()
        return bits, any(k.startswith("t_halo3x3_16") for k in forms), loss

    b_knob, used_knob, l_knob = step("16-mixed", None)
    b_env, used_env, l_env = step(None, "bf16")
    b_32, used_32, l_32 = step(32, "bf16")
    b_def, used_def, l_def = step(None, None)
    assert (b_knob, b_env, b_32, b_def) == (16, 0, 32, 0)
    assert used_knob and used_env and not used_32 and not used_def
    assert l_knob == pytest.approx(l_env, rel=1e-6) and l_32 == pytest.approx(l_def, rel=1e-6)
    assert l_knob != l_def
    with pytest.raises(ValueError):
        build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=B, train_precision="8")._ensure_engine((23, 11), B)


def test_sampling_after_training_uses_the_updated_weights():
    """Train -> sample -> train -> sample on ONE engine (captured rollout graph, weights re-uploaded after optimizer.step()):
    every sample must equal, bit for bit, what a fresh engine built from the current state_dict samples with the same seed --
    i.e. weight reload invalidates everything derived from the old weights (packed fragments, FiLM tables, graph)."""
    from tests.gpu_common import seeded_pair
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True, input_dropout=0.0, dropout=0.1)
    hp = dict(timesteps=4, schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
              interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none", lambda_reconstruction=1.0,
              lambda_reconstruction2=0.5, loss_function="l1", enable_interpolator_dropout=True, sampling_type="cold",
              refine_intermediate_predictions=True, model=mk)
    C, Cs, B = 3, 2, 3
    PF, PI = seeded_pair(64, C, Cs)
    g = torch.Generator().manual_seed(5)
    xt_last, cond = torch.randn(B, C, 23, 11, generator=g).to(DEV), torch.randn(B, C, 23, 11, generator=g).to(DEV)
    sc = torch.rand(B, Cs, 23, 11, generator=g).to(DEV)
    m = build_dyffusion(PF, PI, mk, C, Cs, hp, max_batch=B, use_graph=True)
    opt = torch.optim.SGD(m.model.parameters(), lr=0.05)
    prev = None
    for round_ in range(2):
        m.train()
        for _ in range(2):
            opt.zero_grad()
            out = m.p_losses(xt_last, cond, torch.tensor([0, 1, 3], device=DEV), static_condition=sc)
            out["loss"].backward()
            opt.step()
        m.eval()
        m.seed(99)
        got = {k: v.clone() for k, v in m.sample(cond, static_condition=sc).items()}
        sdF = {k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()}
        fresh = build_dyffusion(sdF, PI, mk, C, Cs, hp, max_batch=B, use_graph=True)
        fresh.seed(99)
        want = fresh.sample(cond, static_condition=sc)
        for k in want:
            assert torch.equal(got[k], want[k]), (round_, k)
        if prev is not None:
            assert not torch.equal(prev["t4_preds"], got["t4_preds"])  # the second round of training changed the forecast
        prev = got


CONV_CASES = [
    # (n, h, w, cin, cout, k, s, p)
    (2, 32, 32, 64, 128, 4, 2, 1),    # encoder conv, 512 output pixels: split-K
    (3, 8, 8, 512, 512, 2, 2, 0),     # 2x2 / s2 on a tiny plane: 48 output pixels, K = 2048
    (2, 16, 16, 256, 128, 3, 1, 1),   # decoder 3x3
    (1, 40, 24, 64, 64, 3, 1, 1),     # 960 pixels: not a multiple of the 128-row tile
    (2, 8, 8, 1024, 512, 1, 1, 0),    # 1x1 on the concat
    (2, 64, 64, 128, 64, 3, 1, 1),    # many tiles: no split
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("kind", [0, 1, 2], ids=["forward", "dgrad", "wgrad"])
def test_matrix_core_training_convs_match_the_plain_kernels(case, kind):
    """csrc/train_gemm.hip (fp32 MFMA implicit GEMM: forward, data gradient, weight gradient; split-K with the ordered merge) against
    the one-thread-per-output VALU kernels of csrc/train.hip on hash-random data: fp32 operands on both sides, so only the
    summation order differs."""
    import dyffusion_amd as D
    from dyffusion_amd.engine import net_config
    cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
    eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False)
    split, unsplit, took = eng.train_conv_check(kind, *case, seed=7 + kind)
    assert took, "the matrix-core form declined a shape it is meant for"
    print(f"kind {kind} {case}: rel max err split {split:.2e}, unsplit {unsplit:.2e}")
    assert split <= 2e-5 and unsplit <= 2e-5


HALO16_CASES = [
    # (n, h, w, cin, cout, k, s, p): 3 x 3 / stride 1 / pad 1 layers with >= 192 tiles of 8 x 16 pixels
    (8, 64, 64, 128, 64, 3, 1, 1),    # two K chunks, 64 output channels
    (8, 64, 64, 64, 128, 3, 1, 1),    # 128-channel column block (forward) / two K chunks (dgrad)
    (8, 60, 60, 64, 64, 3, 1, 1),     # OISST plane: ragged tiles in both directions
    (8, 40, 72, 192, 128, 3, 1, 1),   # three K chunks, h a multiple of the tile, w not
    (13, 17, 250, 64, 64, 3, 1, 1),   # rows of 250 pixels: last column tile 10 wide, h = 17: a one-row tile at the bottom
]


@pytest.mark.parametrize("case", HALO16_CASES, ids=lambda c: "x".join(map(str, c)))
@pytest.mark.parametrize("kind", [0, 1, 2], ids=["forward", "dgrad", "wgrad"])
def test_16bit_halo_training_convs_match_the_plain_kernels(case, kind, form_switch):
    """csrc/train_halo16.hip (round 5: 3 x 3 / stride 1 layers of the training step with 16-bit operands -- tile + halo staged once,
    all nine taps from LDS; the weight gradient with both operands transposed on the way into LDS and the column shift made in
    registers) against the one-thread-per-output fp32 kernels.  The check rounds its hash-random inputs to the 16-bit format first,
    so both sides sum the same exact products and only the summation order differs: 2e-5 of the largest reference value, as
    for the fp32 matrix-core forms.  The tap-by-tap 16-bit implicit GEMM these replace (DYF_TRAIN_HALO16=0) must pass the same check."""
    import dyffusion_amd as D
    from dyffusion_amd.engine import net_config
    form_switch.setenv("DYF_TRAIN_OPERANDS", "bf16")
    cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
    eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False)
    want_form = ("t_halo3x3_16:forward", "t_halo3x3_16:dgrad", "t_wgrad3x3_16")[kind]
    for halo in ("1", "0"):
        form_switch.setenv("DYF_TRAIN_HALO16", halo)
        eng.form_log(True)
        split, unsplit, took = eng.train_conv_check(kind, *case, seed=17 + kind)
        forms = eng.form_log_read()
        eng.form_log(False)
        assert took
        assert (want_form in forms) == (halo == "1"), forms
        print(f"kind {kind} {case} halo16={halo}: rel max err {split:.2e} / {unsplit:.2e}")
        assert split <= 2e-5 and unsplit <= 2e-5


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
@pytest.mark.parametrize("kind", [1, 2], ids=["dgrad", "wgrad"])
def test_16bit_gradient_operands_keep_their_accuracy_at_bench_scale_magnitudes(kind, dtype, form_switch):
    """a mean-reduced loss at real batch sizes hands the backward output gradients of 1e-6 .. 1e-7 (B = 64 on 60 x 60:
    4e-6; NS B = 32 on 256^2: 1e-7) -- fp16's subnormal range (smallest normal 6.1e-5, smallest subnormal 6e-8), where an fp16
    gradient operand without a loss scale keeps a few bits or none.  The 16-bit training mode therefore rounds its operands to bf16 in
    BOTH builds of the library (train_internal.h).  Held here on the engine of either storage type: the data- and weight-gradient
    kernels with the output gradient scaled by 2^-24 (6e-8) must match the fp32 one-thread-per-output kernels exactly as well as at
    O(1) magnitudes (2e-5 of the largest reference value: a power of two commutes with the bf16 rounding); with an fp16 operand the
    result would be zero or off by tens of percent."""
    import dyffusion_amd as D
    from dyffusion_amd.engine import net_config
    form_switch.setenv("DYF_TRAIN_OPERANDS", "bf16")
    cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
    eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False, dtype=dtype)
    errs = {}
    for exp2 in (None, "24"):
        if exp2 is None:
            form_switch.delenv("DYF_TRAIN_CHECK_DZ_EXP2")
        else:
            form_switch.setenv("DYF_TRAIN_CHECK_DZ_EXP2", exp2)
        for case in ((8, 60, 60, 64, 64, 3, 1, 1), (8, 64, 64, 128, 128, 4, 2, 1)):
            split, unsplit, took = eng.train_conv_check(kind, *case, seed=29 + kind)
            assert took
            errs[(exp2, case)] = max(split, unsplit)
            print(f"{dtype} engine, kind {kind} {case}, dz x 2^-{exp2 or 0}: rel max err {split:.2e} / {unsplit:.2e}")
            assert split <= 2e-5 and unsplit <= 2e-5
    eng.close()


@pytest.mark.parametrize("case", [(8, 64, 64, 64, 128, 4, 2, 1), (4, 60, 92, 128, 64, 4, 2, 1), (2, 128, 128, 128, 256, 4, 2, 1)],
                         ids=lambda c: "x".join(map(str, c)))
def test_16bit_stride2_data_gradient_by_parity_classes_matches_the_plain_kernel(case, form_switch):
    """Data gradient of the 4 x 4 / stride 2 / pad 1 encoder convs with 16-bit operands: one launch slice per parity class of the input
    pixels, each running its 2 x 2 reachable taps as a dense K = 4 cout (csrc/train_gemm.hip, pmode) instead of all 16 taps with three
    quarters of the gathers predicated off.  Against the one-thread-per-output fp32 kernel on inputs rounded to 16 bit (2e-5), and the
    16-tap form (DYF_TRAIN_DGRAD_PARITY=0) held to the same bound."""
    import dyffusion_amd as D
    from dyffusion_amd.engine import net_config
    form_switch.setenv("DYF_TRAIN_OPERANDS", "bf16")
    cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
    eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False)
    for parity in ("1", "0"):
        form_switch.setenv("DYF_TRAIN_DGRAD_PARITY", parity)
        eng.form_log(True)
        split, unsplit, took = eng.train_conv_check(1, *case, seed=23)
        forms = eng.form_log_read()
        eng.form_log(False)
        assert took and ("t_gemm_mfma16:dgrad_parity" in forms) == (parity == "1"), forms
        print(f"dgrad {case} parity={parity}: rel max err {split:.2e} / {unsplit:.2e}")
        assert split <= 2e-5 and unsplit <= 2e-5


@pytest.mark.parametrize("case", [(4, 64, 64, 5, 64, 1, 1, 0), (2, 128, 128, 3, 64, 4, 2, 1), (3, 60, 60, 2, 64, 7, 1, 3), (2, 96, 64, 1, 128, 7, 1, 3)],
                         ids=lambda c: "x".join(map(str, c)))
def test_small_channel_weight_gradient_on_the_matrix_cores_matches_the_plain_kernel(case, form_switch):
    """Weight gradient of the convs with a handful of input channels (1 x 1 stem on 5, the readout's 4 x 4 / stride 2 on 3, the ResNet-UNet's
    7 x 7 init conv on 1-2): `t_conv_wgrad_smallc_mfma` (fp32 MFMA, the (tap, channel) axis gathered by constant per-lane offsets, one column of
    ones for the bias gradient's sums) and the per-lane-sums form it replaces, both against the one-thread-per-output kernel."""
    import dyffusion_amd as D
    from dyffusion_amd.engine import net_config
    cfg = net_config(in_channels=3, cond_channels=2, out_channels=3, dim=64, with_time_emb=True, upsample_dims=(64, 64), dropout=0.0)
    eng = D.HipEngine(cfg, cfg, 23, 11, max_batch=1, use_graph=False)
    for mfma in ("1", "0"):
        form_switch.setenv("DYF_TRAIN_SMALLC_MFMA", mfma)
        err, _, took = eng.train_conv_check(3, *case, seed=31)
        print(f"wgrad {case} mfma={mfma}: rel max err {err:.2e}")
        assert took and err <= 2e-5
        # the forward of the same layer (t_conv_fwd_smallc_mfma where the output rows are multiples of 32 pixels, else the
        # four-channels-per-thread kernel), kind 4 = whatever conv_fwd picks
        errf, _, tookf = eng.train_conv_check(4, *case, seed=32)
        print(f"forward {case} mfma={mfma}: rel max err {errf:.2e}")
        assert tookf and errf <= 2e-5


def test_gpu_resident_parameters_train_like_cpu_resident_ones():
    """A forecaster moved to the GPU: gradients are exported device-to-device (dyf_train_export_dev) and the refreshed weights read
    in place (dyf_train_load_weights_dev).  Two SGD steps must produce the parameters of the CPU-resident run (host round
    trips through dyf_train_export / dyf_train_load_weights), up to the atomics of the weight-gradient kernels.  (SGD, not
    Adam: the conv biases in front of a training-mode BatchNorm have an exactly-zero true gradient, i.e. pure rounding noise,
    which Adam's normalisation turns into steps of size lr in a random direction.)"""
    z = load_npz("plosses_train_a.npz")
    hp = json.loads(str(z["hp"]))
    hp["model"] = dict(hp["model"], dropout=0.0)
    PF, PI = split_state(z, "F"), split_state(z, "I")
    xt_last, cond, sc, t = (torch.from_numpy(z[k]).to(DEV) for k in ("xt_last", "cond", "sc", "t"))
    finals = []
    for on_gpu in (False, True):
        m = build_dyffusion(PF, PI, hp["model"], 4, 1, hp, max_batch=hp["B"])
        if on_gpu:
            m.model.cuda()
        m.train()
        opt = torch.optim.SGD(m.model.parameters(), lr=0.05)
        for _ in range(2):
            opt.zero_grad()
            out = m.p_losses(xt_last, cond, t, static_condition=sc)
            out["loss"].backward()
            assert all(p.grad is not None and p.grad.device == p.device for p in m.model.parameters())
            opt.step()
        finals.append({k: v.detach().cpu().clone() for k, v in m.model.state_dict().items()})
    for k, a in finals[0].items():
        b = finals[1][k]
        if a.is_floating_point():
            # (tolerance: the fp32 atomics of the weight-gradient kernels make the first step's gradients differ in the last bits from run
            # to run; the second step then sees weights that differ by ~1e-8, and a (Leaky)ReLU or L1 residual that this moves across
            # zero changes a gradient by a whole term -- parameters after two steps differ by up to ~1e-5 absolute / 5e-5 relative
            # between two runs of the SAME path, seen in about half the runs.  A wrong export or refresh path is off by O(1).)
            assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), k
        else:
            assert torch.equal(a, b), k


def test_engine_lifecycle_does_not_leak_device_memory():
    """Create -> sample (graph) -> training step -> sample -> destroy, repeatedly: everything the engine allocated (workspace, packed
    weights, captured graphs, training copy, tapes, allocator pool) is returned when it is destroyed."""
    import gc
    from tests.gpu_common import seeded_pair
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True, input_dropout=0.0, dropout=0.1)
    hp = dict(timesteps=3, schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
              interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none", lambda_reconstruction=1.0,
              lambda_reconstruction2=0.5, loss_function="l1", enable_interpolator_dropout=True, sampling_type="cold",
              refine_intermediate_predictions=True, model=mk)
    PF, PI = seeded_pair(64, 3, 2)
    g = torch.Generator().manual_seed(5)
    x, c = torch.randn(2, 3, 23, 11, generator=g).to(DEV), torch.rand(2, 2, 23, 11, generator=g).to(DEV)

    def used():
        free, total = torch.cuda.mem_get_info()
        return (total - free) / 2 ** 20

    marks = []
    for _ in range(5):
        m = build_dyffusion(PF, PI, mk, 3, 2, hp, max_batch=2, use_graph=True)
        m.sample(x, static_condition=c)
        m.train()
        out = m.p_losses(x, x, torch.tensor([0, 2], device=DEV), static_condition=c)
        out["loss"].backward()
        m.eval()
        m.sample(x, static_condition=c)
        del m, out
        gc.collect()
        torch.cuda.synchronize()
        marks.append(used())
    print("MiB in use after each destroy:", [round(v) for v in marks])
    assert marks[-1] - marks[1] <= 32.0


@pytest.mark.parametrize("mask_kind", ["rows", "elements"])
def test_get_loss_with_predictions_mask_matches_autograd_of_the_oracle(mask_kind):
    """`BaseModel.get_loss(..., predictions_mask=m)` (_base_model.py:132-135): `criterion(predictions[m], targets)` -- the mean and
    its gradient run over the selected elements only.  Engine training step vs torch.autograd over the oracle's forward with the
    engine's dropout masks; a mask over the batch rows (1-D) and a full-shape element mask."""
    from tests.gpu_common import mirror_from_params
    z = load_npz("interp_train_a.npz")
    hp = json.loads(str(z["hp"]))
    mk, P = hp["model"], split_state(z, "P")
    dyn, cond, t = torch.from_numpy(z["dynamics"]), torch.from_numpy(z["cond"]), torch.from_numpy(z["t"])
    B, C = dyn.shape[0], dyn.shape[2]
    w = hp["window"]
    net = mirror_from_params(P, mk, (w + 1) * C, cond.shape[1], C)
    net.hparams.loss_function = hp["loss_function"]
    net.train()
    x = torch.cat([dyn[:, :w].reshape(B, w * C, *dyn.shape[-2:]), dyn[:, -1]], dim=1)  # window frames + the last frame
    full_targets = dyn[torch.arange(B), w + t - 1]
    g = torch.Generator().manual_seed(3)
    if mask_kind == "rows":
        m = torch.zeros(B, dtype=torch.bool)
        m[::2] = True
    else:
        m = torch.rand(full_targets.shape, generator=g) < 0.4
    targets = full_targets[m]
    seed = 515
    net._own_engine(B, dyn.shape[-2:]).seed(seed)
    loss = net.get_loss(x.to(DEV), targets.to(DEV), condition=cond.to(DEV), time=t.to(DEV), predictions_mask=m.to(DEV))
    loss.backward()
    uh, uw = mk["upsample_dims"]
    drop = R.EngineDropout(seed, mk["dim"], uh, uw)
    drop.begin_forward()
    Pg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
          for k, v in P.items()}
    pred = nets.unet_simple_forward(Pg, mk, x, t.float(), cond, dropout=drop, bn_training=True)
    want = losses.criterion_fn(hp["loss_function"])(pred[m], targets)
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    grads = {k: v.grad for k, v in Pg.items() if torch.is_tensor(v) and v.requires_grad}
    got = {k: p.grad for k, p in net.named_parameters()}
    gn = float(torch.cat([gg.reshape(-1) for gg in grads.values()]).norm())
    worst = max(float((got[k].cpu() - grads[k]).norm()) for k in grads) / gn
    print(f"predictions_mask ({mask_kind}): loss {float(loss):.6f}, worst per-tensor gradient error / grad norm = {worst:.2e}")
    assert worst <= 1e-3


def test_training_step_with_input_dropout_matches_autograd_of_the_oracle():
    """input_dropout > 0 in TRAIN mode (reference golden net_unet_simple_e's network: dim 8, input_dropout 0.1, dropout 0.15): the
    stage-1 `get_loss` step -- dropout_input and its adjoint (the same keep map on the gradient) -- against torch.autograd over the
    oracle with the engine's masks."""
    from tests.gpu_common import mirror_from_params
    z = load_npz("net_unet_simple_e.npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    assert cfg["input_dropout"] > 0
    x, t, c = torch.from_numpy(z["x"]), torch.from_numpy(z["t"]), torch.from_numpy(z["c"])
    n_out = z["y_eval"].shape[1]
    net = mirror_from_params(P, cfg, x.shape[1], c.shape[1], n_out)
    net.train()
    targets = torch.randn(x.shape[0], n_out, *x.shape[-2:], generator=torch.Generator().manual_seed(8))
    seed = 2024
    net._own_engine(x.shape[0], x.shape[-2:]).seed(seed)
    loss = net.get_loss(x.to(DEV), targets.to(DEV), condition=c.to(DEV), time=t.to(DEV))
    loss.backward()
    uh, uw = cfg["upsample_dims"]
    drop = R.EngineDropout(seed, cfg["dim"], uh, uw, input_dropout=True)
    drop.begin_forward()
    Pg = {k: (v.clone().requires_grad_(True) if v.is_floating_point() and not k.endswith(("running_mean", "running_var")) else v)
          for k, v in P.items()}
    pred = nets.unet_simple_forward(Pg, cfg, x, t, c, dropout=drop, bn_training=True)
    want = F.mse_loss(pred, targets)
    want.backward()
    assert float(loss) == pytest.approx(float(want), rel=1e-4)
    grads = {k: v.grad for k, v in Pg.items() if torch.is_tensor(v) and v.requires_grad}
    got = {k: p.grad for k, p in net.named_parameters()}
    gn = float(torch.cat([gg.reshape(-1) for gg in grads.values()]).norm())
    worst = max(float((got[k].cpu() - grads[k]).norm()) for k in grads) / gn
    print(f"input_dropout training step: loss {float(loss):.6f}, worst per-tensor gradient error / grad norm = {worst:.2e}")
    assert worst <= 1e-3
