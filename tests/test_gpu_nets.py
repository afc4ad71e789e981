"""-m gpu: dyf_net_forward (HIP) against the oracle / the reference's golden outputs.

Tolerance (stated): the engine keeps activations in bf16 (8 mantissa bits) and accumulates in fp32; against the
fp32 oracle we require rel-RMS <= 1e-2 per unet_simple forward (measured 4.3e-3 - 6.3e-3, printed; the reference itself
under torch.autocast(bf16) is at 1.2e-2 per forward on the full-size fixture, profiles/r02_bf16_drift.json).
"""
import json

import pytest
import torch

from oracle import init as oinit
from oracle import nets
from tests.gpu_common import DEV, mirror_from_params, nhwc_masks, seeded_pair
from tests.helpers import jload, load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu
TOL = 1e-2


@pytest.mark.parametrize("name", ["net_unet_simple_a", "net_unet_simple_b", "net_unet_simple_c", "net_unet_simple_d", "net_unet_simple_e"])
def test_small_nets_match_reference_goldens(name):
    """dim 4/8 networks straight from the reference's own outputs (direct-conv path: channels < 64)."""
    z = load_npz(name + ".npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    net = mirror_from_params(P, cfg, x.shape[1], 0 if c is None else c.shape[1], z["y_eval"].shape[1])
    y = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    err = rel_rms(y, z["y_eval"])
    print(name, "eval rel-rms", err)
    assert err <= TOL
    # dropout with the reference's own mask stream (seeded), injected as keep-masks
    src = nets.DropoutSeeded(int(z["dropout_seed"]), record=True)
    y_or = nets.unet_simple_forward(P, cfg, x, t, c, dropout=src)
    assert rel_rms(y_or, z["y_drop"]) < 1e-5
    eng = net._engine
    y = eng.net_forward(0, x.to(DEV), t.to(DEV), None if c is None else c.to(DEV), dropout_mode=2,
                        masks=nhwc_masks(src.masks)).cpu()
    err = rel_rms(y, z["y_drop"])
    print(name, "dropout rel-rms", err)
    assert err <= TOL


@pytest.mark.parametrize("hw,up,nb,n_in,n_cond,mode", [((23, 11), (64, 64), 3, 6, 2, "bilinear"),
                                                        ((40, 17), (128, 64), 2, 3, 2, "bilinear"),
                                                        ((32, 32), None, 2, 3, 0, "bilinear"),
                                                        ((23, 11), (64, 64), 3, 6, 2, "nearest"),
                                                        ((221, 42), (256, 256), 1, 3, 2, "nearest")])
def test_dim64_mfma_path_matches_oracle(hw, up, nb, n_in, n_cond, mode):
    cfg = dict(dim=64, upsample_dims=None if up is None else list(up), outer_sample_mode=mode, with_time_emb=True,
               dropout=0.15, input_dropout=0.0)
    if up is None:
        hw = (64, 64)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, n_in + n_cond, 3), seed=33)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(nb, n_in, *hw, generator=g)
    c = torch.rand(nb, n_cond, *hw, generator=g) if n_cond else None
    t = torch.tensor([1.0, 6.5, 0.25][:nb])
    net = mirror_from_params(P, cfg, n_in, n_cond, 3)
    taps = {}
    with torch.no_grad():
        want = nets.unet_simple_forward(P, cfg, x, t, c, taps=taps)
    got = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    err = rel_rms(got, want)
    print("dim64", hw, up, "rel-rms", err)
    assert err <= TOL
    # MC dropout, injected masks
    src = nets.DropoutSeeded(5, record=True)
    with torch.no_grad():
        want = nets.unet_simple_forward(P, cfg, x, t, c, dropout=src)
    got = net._engine.net_forward(0, x.to(DEV), t.to(DEV), None if c is None else c.to(DEV), dropout_mode=2,
                                  masks=nhwc_masks(src.masks)).cpu()
    err = rel_rms(got, want)
    print("dim64 dropout", hw, up, "rel-rms", err)
    assert err <= TOL


def test_mfma_and_direct_engines_agree():
    import dyffusion_amd as D
    cfg = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.0)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=34)
    g = torch.Generator().manual_seed(8)
    x, c, t = torch.randn(2, 3, 23, 11, generator=g), torch.rand(2, 2, 23, 11, generator=g), torch.tensor([2.0, 3.0])
    outs = []
    for mfma in (True, False):
        net = mirror_from_params(P, cfg, 3, 2, 3)
        nc = net.engine_net_config()
        eng = D.HipEngine(nc, nc, 23, 11, max_batch=2, use_graph=False, enable_mfma=mfma)
        eng.load_weights(0, net.state_dict())
        outs.append(eng.net_forward(0, x.to(DEV), t.to(DEV), c.to(DEV)).cpu())
    assert rel_rms(outs[0], outs[1]) <= 6e-3


def test_rows_are_independent():
    cfg = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.0)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=35)
    g = torch.Generator().manual_seed(9)
    x, c = torch.randn(3, 3, 23, 11, generator=g), torch.rand(3, 2, 23, 11, generator=g)
    t = torch.tensor([1.0, 2.0, 3.0])
    net = mirror_from_params(P, cfg, 3, 2, 3)
    full = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    for r in range(3):
        one = net(x[r:r + 1].to(DEV), time=t[r:r + 1].to(DEV), condition=c[r:r + 1].to(DEV)).cpu()
        assert torch.equal(one[0], full[r]), f"row {r} depends on its batch neighbours"


@pytest.mark.parametrize("force_igemm2", [False, True], ids=["default-conv-forms", "second-igemm-form"])
def test_fullsize_ns_forwards_match_reference_fields(force_igemm2, form_switch):
    """BASELINE config 2 shapes (221x42 -> 256^2, dim 64): one forecaster + one interpolator forward vs the
    reference's own outputs (fixture G6).  force_igemm2: the conv form production picks at bench batch sizes."""
    if force_igemm2:
        form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")
    meta, fields = jload("fullsize_checksums.json"), load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), meta["seeds"]["forecaster"])
    PI = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 8, 3), meta["seeds"]["interpolator"])
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    F_ = mirror_from_params(PF, mk, 3, 2, 3)
    I_ = mirror_from_params(PI, mk, 6, 2, 3)
    yF = F_(x0.to(DEV), time=torch.tensor([3.0], device=DEV), condition=c.to(DEV)).cpu()
    eF = rel_rms(yF, fields["yF"])
    # feed the interpolator the REFERENCE's forecaster output so the two checks are independent
    yI = I_(torch.cat([x0, torch.from_numpy(fields["yF"])], 1).to(DEV), time=torch.tensor([5.0], device=DEV),
            condition=c.to(DEV)).cpu()
    eI = rel_rms(yI, fields["yI"])
    print("fullsize forward rel-rms F", eF, "I", eI)
    assert eF <= TOL and eI <= TOL


def test_sparse_last_decoder_block_writes_every_pixel_the_readout_reads(form_switch):
    """The last decoder block computes only the output columns the readout's final bilinear resample touches (104 of 256
    for the NS grid).  With the block's output buffer poisoned (NaN) before every launch, a needed-but-skipped pixel would
    surface as NaN in the network output; the dense form (DYF_SPARSE_DEC5=0 at weight upload) must agree bit for bit on the
    pixels that matter, i.e. on the final output."""
    meta, fields = jload("fullsize_checksums.json"), load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), meta["seeds"]["forecaster"])
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"] + 1)
    x0 = torch.randn(3, 3, 221, 42, generator=g)
    c = torch.rand(3, 2, 221, 42, generator=g)
    t = torch.tensor([3.0, 0.0, 11.0], device=DEV)
    form_switch.setenv("DYF_POISON_DEC5", "1")
    sparse = mirror_from_params(PF, mk, 3, 2, 3)
    y_sparse = sparse(x0.to(DEV), time=t, condition=c.to(DEV)).cpu()
    assert torch.isfinite(y_sparse).all(), "a pixel the readout reads was not written by the sparse-column conv"
    assert rel_rms(y_sparse[:1], nets.unet_simple_forward(PF, mk, x0[:1], t[:1].cpu(), c[:1])) <= TOL
    form_switch.setenv("DYF_SPARSE_DEC5", "0")
    dense = mirror_from_params(PF, mk, 3, 2, 3)
    y_dense = dense(x0.to(DEV), time=t, condition=c.to(DEV)).cpu()
    assert torch.equal(y_sparse, y_dense)


def test_interpolation_experiment_evaluation_step_matches_oracle():
    """Stage-1 interpolator evaluation (interpolation.py:69-127): t = 1..h-1 from (first frame, last frame), N members."""
    import dyffusion_amd as D

    h, C, n_cond, nb, N = 4, 2, 1, 2, 3
    cfg = dict(dim=8, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.1, input_dropout=0.0)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(8, 2 * C + n_cond, C), seed=5)
    net = mirror_from_params(P, cfg, 2 * C, n_cond, C)
    exp = D.InterpolationExperiment(net, horizon=h, num_predictions=N, enable_inference_dropout=False)
    g = torch.Generator().manual_seed(8)
    dyn = torch.randn(nb, h + 1, C, 19, 13, generator=g)
    cond = torch.rand(nb, n_cond, 19, 13, generator=g)
    out = exp.evaluation_step({"dynamics": dyn.to(DEV), "condition": cond.to(DEV)})
    x = torch.cat([dyn[:, 0], dyn[:, -1]], 1)
    mses = []
    for t in range(1, h):
        with torch.no_grad():
            want = nets.unet_simple_forward(P, cfg, x, torch.full((nb,), float(t)), cond)
        got = out[f"t{t}_preds"].cpu()
        assert got.shape == (N, nb, C, 19, 13)
        for n in range(N):  # dropout off: every member equals the deterministic forward
            assert rel_rms(got[n], want) <= TOL
        assert torch.equal(out[f"t{t}_targets"].cpu(), dyn[:, t])
        mses.append(float(((want - dyn[:, t]) ** 2).mean()))
        assert out[f"val/t{t}/ipol/mse"] == pytest.approx(mses[-1], rel=5e-2)
    assert out[f"val/{h}h_avg/ipol/mse"] == pytest.approx(sum(mses) / len(mses), rel=5e-2)
    # the same object drives DYffusion as its interpolator
    assert exp.true_horizon == h and exp.window == 1 and exp.horizon_range == [1, 2, 3]


def test_module_on_the_gpu_and_in_place_weight_edits_reach_the_engine():
    """`net.cuda()` (state_dict tensors on the device) uploads like CPU parameters; an in-place edit of a parameter afterwards
    (EMA swap, manual surgery) is picked up by the next forward through the version counters, without `load_state_dict`."""
    cfg = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.0)
    P = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), seed=33)
    g = torch.Generator().manual_seed(4)
    x, c, t = torch.randn(2, 3, 23, 11, generator=g), torch.rand(2, 2, 23, 11, generator=g), torch.tensor([1.0, 2.0])
    net = mirror_from_params(P, cfg, 3, 2, 3).cuda()
    y = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    assert rel_rms(y, nets.unet_simple_forward(P, cfg, x, t, c)) <= TOL
    with torch.no_grad():
        net.readout[0].bias.add_(0.5)          # in place: no load_state_dict
        net.output_ops[5].ops[2].weight.mul_(1.25)
    P2 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    y2 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    assert rel_rms(y2, nets.unet_simple_forward(P2, cfg, x, t, c)) <= TOL
    assert rel_rms(y2, y) > 0.05
    # a write through the `.data` alias is invisible to the version counters: the module must be told
    net.readout[0].bias.data.sub_(0.5)
    net.mark_weights_modified()
    P3 = {k: v.detach().cpu().clone() for k, v in net.state_dict().items()}
    y3 = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
    assert rel_rms(y3, nets.unet_simple_forward(P3, cfg, x, t, c)) <= TOL and rel_rms(y3, y2) > 0.01


_ENC0_SCRIPT = """
import sys, torch
sys.path.insert(0, {root!r})
from tests.gpu_common import mirror_from_params, seeded_pair
from tests.helpers import apply_test_forms
apply_test_forms()
PF, PI = seeded_pair(64, 3, 2, seeds=(31, 32))
mk = dict(dim=64, with_time_emb=True, upsample_dims=[256, 256], dropout=0.25)
g = torch.Generator().manual_seed(5)
x, c, t = torch.randn(8, 6, 221, 42, generator=g), torch.rand(8, 2, 221, 42, generator=g), torch.arange(8.0)
net = mirror_from_params(PI, mk, 6, 2, 3)
net.engine_dtype = {dtype!r}
net(x.cuda(), time=t.cuda(), condition=c.cuda())  # creates the engine
eng = net._engine
outs = {{}}
for mode in (0, 1):
    eng.seed(77)
    y = eng.net_forward(net._engine_slot, x.cuda(), t.cuda(), c.cuda(), dropout_mode=mode)
    outs[f"enc0_{{mode}}"] = eng.read_block_output(net._engine_slot, 0, 8).cpu()
    outs[f"y_{{mode}}"] = y.cpu()
torch.save(outs, {out!r})
"""


@pytest.mark.parametrize("dtype", ["bf16", "fp16"])
def test_persistent_enc0_kernel_matches_the_implicit_gemm_form(tmp_path, dtype, form_switch):
    """enc0 on the fused stem at NB = 8 (4 096 row segments: the persistent kernel of conv_enc0_stem.hip takes the layer) against
    the same forward with DYF_ENC0_STEM=0 (conv_igemm2_kernel): same K order, same epilogue and dropout stream -> the block output
    agrees to a 16-bit rounding tie, with and without engine dropout.  The form is chosen once per process: two subprocesses."""
    import os
    import subprocess
    import sys as _sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    outs = []
    for on in ("1", "0"):
        out = str(tmp_path / f"enc0_{on}.pt")
        subprocess.run([_sys.executable, "-c", _ENC0_SCRIPT.format(root=root, out=out, dtype=dtype)], check=True, env=form_switch.env(DYF_ENC0_STEM=on),
                       timeout=900)
        outs.append(torch.load(out))
    for k in outs[0]:
        a, b = outs[0][k], outs[1][k]
        assert torch.isfinite(a).all() and float(a.std()) > 0, k
        err, frac = rel_rms(a, b), float((a != b).float().mean())
        print(f"{k}: persistent vs implicit-GEMM enc0 rel-rms {err:.2e}, {frac:.2e} of the elements differ")
        assert err <= (2e-3 if k.startswith("enc0") else 1e-2), k
    assert float((outs[0]["enc0_1"] == 0).float().mean()) > 0.2  # dropout was on (p = 0.25 after ReLU)


@pytest.mark.parametrize("mode", ["bilinear", "nearest"])
def test_row_block_stem_matches_the_per_pixel_form(mode, form_switch):
    """The fused stem's resample kernel (221 x 42 -> 256 x 256, zero-bordered 16-channel tensor): the row-block form
    (stem16_rows_kernel: the block's source rows in LDS, 16-byte tap reads, two lanes per pixel) evaluates the same expressions in
    the same order as the per-pixel form (DYF_STEM16_ROWS=0); FMA contraction differs, so the first block's output agrees to a
    16-bit rounding tie on a few elements (nearest: no arithmetic, bit for bit)."""
    PF, PI = seeded_pair(64, 3, 2, seeds=(41, 42))
    mk = dict(dim=64, with_time_emb=True, upsample_dims=[256, 256], outer_sample_mode=mode)
    g = torch.Generator().manual_seed(9)
    x, c, t = torch.randn(3, 6, 221, 42, generator=g), torch.rand(3, 2, 221, 42, generator=g), torch.tensor([1.0, 4.0, 7.0])
    net = mirror_from_params(PI, mk, 6, 2, 3)
    outs = []
    for rows in ("1", "0"):
        form_switch.setenv("DYF_STEM16_ROWS", rows)
        y = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
        outs.append((net._engine.read_block_output(net._engine_slot, 0, 3).cpu(), y))
    (e_new, y_new), (e_old, y_old) = outs
    assert torch.isfinite(y_new).all() and float(e_new.std()) > 0
    frac = float((e_new != e_old).float().mean())
    print(f"stem forms ({mode}): {frac:.2e} of the first block's outputs differ, rel-rms {rel_rms(e_new, e_old):.2e}; network output rel-rms {rel_rms(y_new, y_old):.2e}")
    if mode == "nearest":
        assert torch.equal(e_new, e_old) and torch.equal(y_new, y_old)
    else:
        assert frac <= 1e-3 and rel_rms(e_new, e_old) <= 1e-3 and rel_rms(y_new, y_old) <= 1e-2  # a flipped 16-bit value is a 4e-3 perturbation


def test_quad_form_of_the_x2_upsample_matches_the_per_pixel_form(form_switch):
    """The materialised x2 bilinear upsample of the small decoder planes (dec0-dec2): up2x_quad_kernel (a thread makes the 2 x 2
    outputs of one input pixel from its 3 x 3 neighbourhood) against up2x_kernel<8> (DYF_UP2X_QUAD=0): same stencils and
    expressions; FMA contraction may differ by a 16-bit rounding tie on a few elements."""
    PF, PI = seeded_pair(64, 3, 2, seeds=(43, 44))
    mk = dict(dim=64, with_time_emb=True, upsample_dims=[256, 256])
    g = torch.Generator().manual_seed(10)
    x, c, t = torch.randn(2, 6, 221, 42, generator=g), torch.rand(2, 2, 221, 42, generator=g), torch.tensor([2.0, 6.0])
    net = mirror_from_params(PI, mk, 6, 2, 3)
    outs = []
    for quad in ("1", "0"):
        form_switch.setenv("DYF_UP2X_QUAD", quad)
        y = net(x.to(DEV), time=t.to(DEV), condition=c.to(DEV)).cpu()
        outs.append([net._engine.read_block_output(net._engine_slot, li, 2).cpu() for li in (6, 7, 8)] + [y])
    for name, a, b in zip(("dec0", "dec1", "dec2", "y"), outs[0], outs[1]):
        assert torch.isfinite(a).all() and float(a.std()) > 0
        frac, err = float((a != b).float().mean()), rel_rms(a, b)
        print(f"{name}: quad vs per-pixel upsample: {frac:.2e} of the elements differ, rel-rms {err:.2e}")
        assert err <= (1e-3 if name == "dec0" else 1e-2), name
