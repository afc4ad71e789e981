"""-m gpu: the ResNet-UNet backbone (src/models/unet.py: WS-conv, GroupNorm+SiLU+FiLM, LinearAttention, Attention,
channel LayerNorm) on the HIP engine, against the reference's golden outputs and the oracle.

Both 16-bit builds are tested.  fp16 is this backbone's DEFAULT (dyffusion_amd.engine.default_dtype_for):
rel-RMS <= 4e-3 per forward, <= 1e-2 per field over a rollout (SURVEY 8c's bound for a 16-bit engine).  bf16 (explicit
`engine_dtype = "bf16"` / `DYffusion(dtype="bf16")`): <= 2e-2 per forward (the net is ~3x deeper than unet_simple), rollouts as
stated in the tests."""
import json
import os

import pytest
import torch

import dyffusion_amd as D
from oracle import init as oinit
from oracle import nets, sampler
from tests.gpu_common import DEV
from tests.helpers import load_npz, rel_rms, split_state

pytestmark = pytest.mark.gpu
TOL = {"bf16": 2e-2, "fp16": 4e-3}
DTYPES = ["fp16", "bf16"]


def test_fp16_is_the_default_dtype_of_the_resnet_unet():
    net = D.Unet(dim=8, dim_mults=(1, 2), with_time_emb=True, num_input_channels=2, num_output_channels=1)
    assert D.default_dtype_for(net) == "fp16"
    y = net(torch.randn(1, 2, 8, 8).to(DEV), time=torch.ones(1).to(DEV))
    assert net._engine.dtype == "fp16" and bool(torch.isfinite(y).all())
    m = D.DYffusion(net, D.InterpolatorHandle(D.Unet(dim=8, dim_mults=(1, 2), with_time_emb=True, num_input_channels=2,
                                                     num_output_channels=1), 4), timesteps=4, forward_conditioning="data",
                    interpolate_before_t1=True)
    assert m._engine_opts["dtype"] == "fp16"
    ns = D.UNet(dim=8, with_time_emb=True, upsample_dims=[64, 64], num_input_channels=2, num_output_channels=1)
    assert D.default_dtype_for(ns) == "bf16"


def mirror(P, cfg, n_in, n_cond, n_out, dtype=None):
    net = D.Unet(dim=cfg["dim"], dim_mults=cfg["dim_mults"], with_time_emb=cfg.get("with_time_emb", True),
                 block_dropout=cfg.get("block_dropout", 0.0), block_dropout1=cfg.get("block_dropout1", 0.0),
                 attn_dropout=cfg.get("attn_dropout", 0.0), input_dropout=cfg.get("input_dropout", 0.0), num_input_channels=n_in,
                 keep_spatial_dims=cfg.get("keep_spatial_dims", False), double_conv_layer=cfg.get("double_conv_layer", True),
                 learned_sinusoidal_cond=cfg.get("learned_sinusoidal_cond", False), learned_sinusoidal_dim=cfg.get("learned_sinusoidal_dim", 16),
                 num_output_channels=n_out,
                 num_conditional_channels=n_cond)
    if dtype is not None:
        net.engine_dtype = dtype
    net.load_state_dict(P, strict=True)
    return net


def engine_masks(masks, nlev=None, n_input=0):
    """oracle keep-masks -> engine layout: activations NCHW -> NHWC; the attention-probability mask (b, heads = 4, n, n) of
    mid_attn is passed unchanged (every activation has >= 8 channels, so the head count identifies it).  Site order = execution
    order over the layers with p > 0."""
    return [(m if (m.shape[1] == 4 and m.shape[2] == m.shape[3]) else m.permute(0, 2, 3, 1)).contiguous().to(DEV) for m in masks]


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("name", ["net_unet_resnet_a", "net_unet_resnet_b", "net_unet_resnet_c", "net_unet_resnet_d", "net_unet_resnet_e",
                                  "net_unet_resnet_g"])
def test_small_resnet_unets_match_reference_goldens(name, dtype):
    z = load_npz(name + ".npz")
    P, cfg = split_state(z, "P"), json.loads(str(z["cfg"]))
    x, t = torch.from_numpy(z["x"]), torch.from_numpy(z["t"])
    c = torch.from_numpy(z["c"]) if "c" in z else None
    net = mirror(P, cfg, x.shape[1], 0 if c is None else c.shape[1], z["y_eval"].shape[1], dtype)
    y = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    assert net._engine.dtype == dtype
    err = rel_rms(y, z["y_eval"])
    print(name, dtype, "eval rel-rms", err)
    assert err <= TOL[dtype]
    src = nets.DropoutSeeded(int(z["dropout_seed"]), record=True)
    y_or = nets.resnet_unet_forward(P, cfg, x, t, c, dropout=src)
    assert rel_rms(y_or, z["y_drop"]) < 1e-5
    y = net._engine.net_forward(0, x.to(DEV), t.to(DEV), None if c is None else c.to(DEV), dropout_mode=2,
                                masks=engine_masks(src.masks, len(cfg["dim_mults"]), 2 if cfg.get("input_dropout", 0) > 0 else 0)).cpu()
    err = rel_rms(y, z["y_drop"])
    print(name, dtype, "dropout rel-rms", err)
    assert err <= TOL[dtype]


def rollout_masks(masks, nlev, per_forward):
    out = []
    for f0 in range(0, len(masks), per_forward):
        out += engine_masks(masks[f0:f0 + per_forward], nlev)
    return out


def seeded_unet(dim, mults, cin, cout, seed):
    net = D.Unet(dim=dim, dim_mults=mults, with_time_emb=True, num_input_channels=cin, num_output_channels=cout)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    st = oinit.seeded_state(shapes, seed, gain=1.0)
    for k in st:
        if k.endswith(".norm.g"):
            st[k] = 1.0 + 0.1 * torch.randn(shapes[k], generator=torch.Generator().manual_seed(len(k)))
    return st


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("force_igemm2", [False, True], ids=["default-conv-forms", "second-igemm-form"])
@pytest.mark.parametrize("hw,nb,n_in,n_cond", [((60, 60), 2, 2, 0), ((32, 48), 2, 1, 1)])
def test_dim64_oisst_shape_matches_oracle(hw, nb, n_in, n_cond, force_igemm2, dtype, form_switch):
    """OISST configuration of the reference (dim 64, mults (1,2,4), 60x60): MFMA conv path.  force_igemm2: every conv with
    cout % 128 == 0 (two-source skip convs, fp32 GroupNorm inputs, residual epilogues) through conv_igemm2_kernel, which
    production selects only for large batches."""
    if force_igemm2:
        form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.2,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), n_in + n_cond, 1, seed=51)
    g = torch.Generator().manual_seed(6)
    x = torch.randn(nb, n_in, *hw, generator=g)
    c = torch.rand(nb, n_cond, *hw, generator=g) if n_cond else None
    t = torch.tensor([1.0, 4.5][:nb])
    net = mirror(P, cfg, n_in, n_cond, 1, dtype)
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, c)
    got = net(x.to(DEV), time=t.to(DEV), condition=None if c is None else c.to(DEV)).cpu()
    err = rel_rms(got, want)
    print("resnet-unet dim64", dtype, hw, "rel-rms", err)
    assert err <= TOL[dtype]
    src = nets.DropoutSeeded(9, record=True)
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, c, dropout=src)
    got = net._engine.net_forward(0, x.to(DEV), t.to(DEV), None if c is None else c.to(DEV), dropout_mode=2,
                                  masks=engine_masks(src.masks, 3)).cpu()
    err = rel_rms(got, want)
    print("resnet-unet dim64 dropout", dtype, hw, "rel-rms", err)
    assert err <= TOL[dtype]


@pytest.mark.parametrize("dtype,tol", [("fp16", 1e-2), ("bf16", 3e-2)])
def test_oisst_style_rollout_matches_oracle(dtype, tol):
    """DYffusion with the ResNet-UNet pair, OISST settings in miniature: C=1, no static condition, k>0 extra steps,
    forward_conditioning='data+noise' (injected normal draws), refine off, interpolator MC dropout with injected masks."""
    mcfg_i = dict(dim=64, dim_mults=[1, 2], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.2,
                  resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    mcfg_f = dict(mcfg_i, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    PF, PI = seeded_unet(64, (1, 2), 2, 1, seed=61), seeded_unet(64, (1, 2), 2, 1, seed=62)
    hp = dict(timesteps=4, schedule="before_t1_only", additional_interpolation_steps=2, interpolate_before_t1=True,
              sampling_type="cold", refine_intermediate_predictions=False, forward_conditioning="data+noise",
              time_encoding="dynamics", enable_interpolator_dropout=True)
    F_ = mirror(PF, mcfg_f, 1, 1, 1)
    I_ = mirror(PI, mcfg_i, 2, 0, 1)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 4), max_batch=3, dtype=dtype, **hp)
    g = torch.Generator().manual_seed(12)
    x0 = torch.randn(3, 1, 24, 16, generator=g)
    drop = nets.DropoutSeeded(3, record=True)
    gen = torch.Generator().manual_seed(4)
    draws = []

    def nf(t):
        draws.append(torch.randn(t.shape, generator=gen))
        return draws[-1]

    with torch.no_grad():
        want = sampler.sample_loop(lambda x, t, cnd: nets.resnet_unet_forward(PF, mcfg_f, x, t, cnd),
                                   lambda x, t, cnd: nets.resnet_unet_forward(PI, mcfg_i, x, t, cnd, dropout=drop),
                                   x0, None, hp, noise_fn=nf)
    _, got, _ = m.sample_loop(x0.to(DEV), _masks=rollout_masks(drop.masks, 2, 27), _noise=torch.stack(draws, 0).to(DEV))
    assert sorted(got) == sorted(want)
    worst = max(rel_rms(got[k].cpu(), want[k]) for k in want)
    print("OISST-style rollout", dtype, "worst rel-rms", worst)
    assert worst <= tol


@pytest.mark.parametrize("dtype", DTYPES)
def test_flash_attention_long_sequence_with_dropout(dtype):
    """Bottleneck Attention on the MFMA flash kernel with a sequence that spans many key tiles and is not a multiple
    of the tile sizes (N = 30*26 = 780 tokens), attention-probability dropout injected."""
    cfg = dict(dim=64, dim_mults=[1, 2], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.3,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2), 2, 1, seed=71)
    g = torch.Generator().manual_seed(8)
    x, t = torch.randn(2, 2, 60, 52, generator=g), torch.tensor([2.0, 5.0])
    net = mirror(P, cfg, 2, 0, 1, dtype)
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, None)
    got = net(x.to(DEV), time=t.to(DEV)).cpu()
    print("flash attention N=780", dtype, "eval rel-rms", rel_rms(got, want))
    assert rel_rms(got, want) <= TOL[dtype]
    src = nets.DropoutSeeded(21, record=True)
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, None, dropout=src)
    # sites with p > 0: only the attention blocks -> [linattn l0, linattn l1, mid_attn (b,h,n,n), linattn, linattn]
    masks = [(m if i == 2 else m.permute(0, 2, 3, 1)).contiguous().to(DEV) for i, m in enumerate(src.masks)]
    got = net._engine.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=2, masks=masks).cpu()
    print("flash attention N=780", dtype, "dropout rel-rms", rel_rms(got, want))
    assert rel_rms(got, want) <= TOL[dtype]


def test_bf16_is_refused_for_long_resnet_unet_rollouts():
    """the T = 32 OISST plan in bf16 storage drifts 4.6e-2 - 5.2e-2 per field (measured through round 3 under a 7e-2
    'tolerance').  The engine now refuses a ResNet-UNet plan of more than 32 forwards in bf16 instead of serving it; fp16 (the
    backbone's default) is held to 1e-2 below; allow_bf16_long_rollout=True is the explicit override."""
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    PF, PI = seeded_unet(64, (1, 2, 4), 2, 1, seed=1), seeded_unet(64, (1, 2, 4), 2, 1, seed=2)
    hp = dict(timesteps=7, schedule="before_t1_only", additional_interpolation_steps=25, interpolate_before_t1=True,
              sampling_type="cold", refine_intermediate_predictions=False, forward_conditioning="data", enable_interpolator_dropout=False)
    x0 = torch.randn(2, 1, 60, 60, generator=torch.Generator().manual_seed(0)).to(DEV)
    m = D.DYffusion(mirror(PF, cfg, 1, 1, 1), D.InterpolatorHandle(mirror(PI, cfg, 2, 0, 1), 7), max_batch=2, dtype="bf16", **hp)
    with pytest.raises(NotImplementedError, match="bf16"):
        m.sample(x0)
    m2 = D.DYffusion(mirror(PF, cfg, 1, 1, 1), D.InterpolatorHandle(mirror(PI, cfg, 2, 0, 1), 7), max_batch=2, dtype="bf16",
                     allow_bf16_long_rollout=True, **hp)
    out = m2.sample(x0)
    assert sorted(out) == [f"t{i}_preds" for i in range(1, 8)] and all(bool(torch.isfinite(v).all()) for v in out.values())
    short = dict(hp, additional_interpolation_steps=2)  # T = 9: 9 + 15 forwards, served in bf16
    m3 = D.DYffusion(mirror(PF, cfg, 1, 1, 1), D.InterpolatorHandle(mirror(PI, cfg, 2, 0, 1), 7), max_batch=2, dtype="bf16", **short)
    assert sum(m3.sample(x0)["t7_preds"].shape) > 0 and sum(m3._engine.forward_counts()) <= 32


@pytest.mark.parametrize("dtype,tol", [("fp16", 1e-2)])
def test_fullsize_oisst_rollout_matches_reference_fields(dtype, tol):
    """BASELINE configs[2] at full size (fixture G6-OISST, outputs of the imported reference): 60x60, C=1, ResNet-UNet dim 64
    mults (1,2,4) for both networks, h=7 with k=25 extra interpolation steps -- the T=32 plan of 32 forecaster + 61
    interpolator forwards -- forward_conditioning "data+noise" with the reference's normal draws injected, cold sampling,
    dropout off, NB=1.  All seven fields.  Tolerance: the recursion chains 93 forwards of a network with ~60 16-bit
    roundings per forward: measured 4.6e-2 - 5.2e-2 in bf16 (1e-2 per forward) -- such plans are refused in bf16 since round 4, see the
    test above -- and 6e-3 in fp16, the build held to SURVEY 8c's 1e-2."""
    z = load_npz("fullsize_oisst_fields.npz")
    meta = json.loads(str(z["meta"]))
    fc, ic = meta["forecaster_channels"], meta["interpolator_channels"]
    cfg = dict(meta["model"], resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    PF = seeded_unet(64, (1, 2, 4), fc["inputs"] + fc["cond"], 1, seed=meta["seeds"]["forecaster"])
    PI = seeded_unet(64, (1, 2, 4), ic["inputs"] + ic["cond"], 1, seed=meta["seeds"]["interpolator"])
    F_, I_ = mirror(PF, cfg, fc["inputs"], fc["cond"], 1), mirror(PI, cfg, ic["inputs"], ic["cond"], 1)
    hp = dict(timesteps=7, schedule="before_t1_only", additional_interpolation_steps=25, interpolate_before_t1=True,
              sampling_type="cold", refine_intermediate_predictions=False, forward_conditioning="data+noise",
              time_encoding="dynamics", enable_interpolator_dropout=False)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 7), max_batch=1, dtype=dtype, **hp)
    assert m.num_timesteps == meta["num_timesteps"] == 32
    x0 = torch.from_numpy(z["x0"])
    gen = torch.Generator().manual_seed(meta["seeds"]["noise"])  # the draws the reference's patched randn_like made, in order
    noise = torch.stack([torch.randn(x0.shape, generator=gen) for _ in range(32)], 0)
    _, got, _ = m.sample_loop(x0.to(DEV), _noise=noise.to(DEV))
    assert m._engine.forward_counts() == (32, 61)
    errs = {k: rel_rms(got[k].cpu(), z[k]) for k in sorted(got)}
    print(f"full-size OISST rollout ({dtype}) rel-rms", {k: round(v, 5) for k, v in errs.items()})
    assert sorted(got) == [f"t{i}_preds" for i in range(1, 8)] and max(errs.values()) <= tol


_ATTN_DROP_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from tests.test_gpu_unet_resnet import mirror, seeded_unet
from tests.helpers import apply_test_forms
apply_test_forms()
cfg = dict(dim=64, dim_mults=[1, 2], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.3,
           resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
P = seeded_unet(64, (1, 2), 2, 1, seed=71)
g = torch.Generator().manual_seed(8)
x, t = torch.randn(3, 2, 32, 32, generator=g), torch.tensor([2.0, 5.0, 1.0])
net = mirror(P, cfg, 2, 0, 1)
net(x.cuda(), time=t.cuda())  # creates the engine
net._engine.seed(2024)
y = net._engine.net_forward(0, x.cuda(), t.cuda(), None, dropout_mode=1).cpu()
torch.save(y, {out!r})
"""


def test_flash_attention_engine_dropout_draws_the_masks_of_the_plain_kernel(tmp_path, form_switch):
    """Attention-probability dropout from the engine's generator: the flash kernel hashes one keep word per PAIR of keys
    (16 x 16 = 256 tokens: both query blocks are whole, the paired form runs), the plain kernel (DYF_FLASH_ATTN=0) evaluates
    `rng_keep(q * N + j)` per element.  Same seed -> the same masks, so the outputs agree to rounding (different masks at
    p = 0.3 would differ by tens of percent).  The kernel form is chosen once per process: two subprocesses."""
    import subprocess
    import sys as _sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    outs = {}
    # flash_attention4_kernel (default: pipelined, V through transposing LDS reads), flash_attention3_kernel, flash_attention2_kernel,
    # the plain per-query kernel
    for flash in ("4", "3", "2", "0"):
        out = str(tmp_path / f"y{flash}.pt")
        env = form_switch.env(DYF_FLASH_ATTN=flash)
        subprocess.run([_sys.executable, "-c", _ATTN_DROP_SCRIPT.format(root=root, out=out)], check=True, env=env, timeout=600)
        outs[flash] = torch.load(out)
    for flash in ("4", "3", "2"):
        err = rel_rms(outs[flash], outs["0"])
        print(f"flash form {flash} (paired keep words) vs plain attention kernel, engine dropout: rel-rms", err)
        assert err <= 1e-2
    assert float(outs["4"].std()) > 0


_HALO5_EPI_SCRIPT = r"""
import sys, torch
sys.path.insert(0, {root!r})
from tests.test_gpu_unet_resnet import mirror, seeded_unet
from tests.helpers import apply_test_forms
apply_test_forms()
cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0,
           resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
g = torch.Generator().manual_seed(6)
nb = 40  # 60 x 60 planes at 40 rows: the 3x3 convs run on conv_up_halo_kernel<5> (ragged 16 x 32 tiles)
x, t = torch.randn(nb, 2, 60, 60, generator=g), (torch.arange(nb) % 7 + 1).float()
net = mirror(P, cfg, 2, 0, 1)
net._own_engine(nb, (60, 60))
net._engine.form_log(True)
y = net(x.cuda(), time=t.cuda()).cpu()
forms = net._engine.form_log_read()
assert "conv_up_halo_kernel<5>" in forms, forms
assert "gn_apply_part_kernel" in forms, forms
torch.save(y, {out!r})
"""


def test_halo5_plain_epilogue_instantiation_equals_the_general_one(tmp_path, form_switch):
    """conv_up_halo_kernel<5, true> (only the no-activation / no-dropout epilogue instantiated: what every 3x3 conv of the
    ResNet-UNet launches by default) against the general instantiation (DYF_HALO5_PLAIN_EPI=0, all twelve epilogues): the same
    source path compiled twice, so the forward must agree BITWISE; the default one is pinned to the oracle by every other test of
    this file.  The switch is read once per process: two subprocesses."""
    import subprocess
    import sys as _sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    outs = {}
    for v in ("1", "0"):
        out = str(tmp_path / f"y{v}.pt")
        env = form_switch.env(DYF_HALO5_PLAIN_EPI=v, DYF_GN_FUSED="0")  # the un-fused chain: conv<5> + statistics epilogue + gn_apply_part
        subprocess.run([_sys.executable, "-c", _HALO5_EPI_SCRIPT.format(root=root, out=out)], check=True, env=env, timeout=600)
        outs[v] = torch.load(out)
    assert torch.isfinite(outs["1"]).all() and float(outs["1"].std()) > 0
    assert torch.equal(outs["1"], outs["0"])


@pytest.mark.parametrize("small_tiles", ["1", "0"], ids=["gn16-16x16-tiles", "halo5-16x32-tiles"])
def test_fused_groupnorm_convs_match_the_unfused_chain_and_are_reproducible(small_tiles, form_switch):
    """(round 6: `small_tiles` -- the 60 x 60 / 30 x 30 levels on conv_gn16_kernel, 16 x 16-pixel tiles and three workgroups per CU, the
    default; DYF_GN16=0 keeps them on conv_up_halo_kernel<5, 2>, 16 x 32 tiles.)
    Round 4 (csrc/gn_fused.h): GroupNorm + FiLM + SiLU + dropout (+ residual) inside the producing conv, with the statistics
    exchanged between the workgroups of a sample INSIDE the launch, against the three-kernel chain (DYF_GN_FUSED=0, read per engine)
    on the OISST shapes at 40 rows -- conv_up_halo_kernel<5, 2> on the 60 x 60 and 30 x 30 levels, conv_igemm2_kernel<2, true> (rows of
    the 15 x 15 planes straddle the 256-pixel tiles) -- eval and with the engine's MC dropout (same masks: the streams are keyed by
    element, not by kernel).  The fused form normalises the fp32 accumulators, the chain the 16-bit rounded conv output: they agree
    to 16-bit rounding.  Two runs of the fused form must agree BITWISE (slot-ordered sums, whatever order the workgroups arrive in)."""
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
    g = torch.Generator().manual_seed(6)
    nb = 40
    x, t = torch.randn(nb, 2, 60, 60, generator=g), (torch.arange(nb) % 7 + 1).float()
    outs = {}
    try:
        for fused in ("1", "0"):
            form_switch.setenv("DYF_GN_FUSED", fused)
            form_switch.setenv("DYF_GN16", small_tiles)
            form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")  # the large-batch implicit-GEMM form at 40 rows (as at 300 rows by default)
            net = mirror(P, cfg, 2, 0, 1, "fp16")
            net._own_engine(nb, (60, 60))
            eng = net._engine
            eng.form_log(True)
            y_eval = net(x.to(DEV), time=t.to(DEV)).cpu()
            forms = eng.form_log_read()
            eng.form_log(False)
            print(fused, sorted(forms))
            if fused == "1":
                tiled = "conv_gn16_kernel+gn_fused" if small_tiles == "1" else "conv_up_halo_kernel<5>+gn_fused"
                other = "conv_up_halo_kernel<5>+gn_fused" if small_tiles == "1" else "conv_gn16_kernel+gn_fused"
                assert tiled in forms and other not in forms, sorted(forms)
                # 16 x 32 tiles: the 15 x 15 level and the two-source convs with c1 != c0 stay on the fused implicit GEMM; 16 x 16 tiles take all
                assert ("conv_igemm2_kernel<2>+gn_fused" in forms) == (small_tiles == "0"), sorted(forms)
                # every GroupNorm runs inside its conv (at most the four 64 -> 64 convs of the 30 x 30 level could fall below the tile
                # threshold of conv_up_halo_kernel<5> at 40 rows: 80 tiles against the 64 it takes since round 4)
                assert "gn_apply_part_kernel" not in forms and sum(forms.get("gn_stats_kernel+gn_apply", {}).values()) <= 4, forms
            else:
                assert not any(k.endswith("+gn_fused") for k in forms) and "gn_apply_part_kernel" in forms, sorted(forms)
            eng.seed(11)
            y_d1 = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
            eng.seed(11)
            y_d2 = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
            y_eval2 = net(x.to(DEV), time=t.to(DEV)).cpu()
            assert torch.equal(y_eval, y_eval2) and torch.equal(y_d1, y_d2), "not reproducible run to run"
            outs[fused] = (y_eval, y_d1)
            eng.close()
    finally:
        form_switch.delenv("DYF_IGEMM2_MIN_TILES")
        form_switch.delenv("DYF_GN_FUSED")
        form_switch.delenv("DYF_GN16")
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, None)
    for i, nm in enumerate(("eval", "engine dropout")):
        e = max(rel_rms(outs["1"][i][r], outs["0"][i][r]) for r in range(nb))
        print(f"fused vs un-fused GroupNorm, {nm}: worst row rel-RMS {e:.3e}")
        assert e <= 4e-3
    e1 = max(rel_rms(outs["1"][0][r], want[r]) for r in range(nb))
    e0 = max(rel_rms(outs["0"][0][r], want[r]) for r in range(nb))
    print(f"vs the fp32 oracle: fused {e1:.3e}, un-fused {e0:.3e}")
    assert e1 <= 4e-3


@pytest.mark.parametrize("hw", [(32, 48), (20, 36), (16, 16), (12, 8)], ids=lambda v: "x".join(map(str, v)))
def test_small_tile_fused_conv_on_ragged_and_tiny_planes(hw, form_switch):
    """conv_gn16_kernel (16 x 16 tiles, one statistics slot per workgroup) away from the OISST shapes: planes that tile evenly (32 x 48),
    ragged in both directions (20 x 36 -> 10 x 18 -> 5 x 9), exactly one tile (16 x 16) and smaller than a tile at every level
    (12 x 8 -> 6 x 4 -> 3 x 2: a workgroup's halo is mostly out-of-image zeros, three of its four waves own no valid row).  The tile
    threshold is forced down (3 rows would not reach 64 tiles), the 256-channel level included.  Against the fp32 oracle, against the
    un-fused three-kernel chain on the same engine build, eval and with the engine's MC dropout; two runs agree bitwise."""
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=77)
    g = torch.Generator().manual_seed(sum(hw))
    nb = 3
    x, t = torch.randn(nb, 2, *hw, generator=g), torch.tensor([1.0, 3.0, 6.0])
    outs = {}
    for fused in ("1", "0"):
        form_switch.setenv("DYF_GN_FUSED", fused)
        form_switch.setenv("DYF_GN16_MIN_TILES", "1")
        form_switch.setenv("DYF_GN16_ANY_PLANE", "1")  # (production leaves planes that fill < 60 % of their tiles to the other forms)
        net = mirror(P, cfg, 2, 0, 1, "fp16")
        net._own_engine(nb, hw)
        eng = net._engine
        eng.form_log(True)
        y = net(x.to(DEV), time=t.to(DEV)).cpu()
        forms = eng.form_log_read()
        eng.form_log(False)
        if fused == "1":
            assert "conv_gn16_kernel+gn_fused" in forms, sorted(forms)
            assert "conv_up_halo_kernel<5>+gn_fused" not in forms, sorted(forms)
        else:
            assert not any(k.endswith("+gn_fused") for k in forms), sorted(forms)
        eng.seed(5)
        yd = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
        eng.seed(5)
        yd2 = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
        assert torch.equal(yd, yd2) and torch.equal(y, net(x.to(DEV), time=t.to(DEV)).cpu())
        assert bool(torch.isfinite(y).all()) and bool(torch.isfinite(yd).all())
        outs[fused] = (y, yd)
        eng.close()
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x, t, None)
    e_or, e_un, e_dr = rel_rms(outs["1"][0], want), rel_rms(outs["1"][0], outs["0"][0]), rel_rms(outs["1"][1], outs["0"][1])
    print(f"{hw}: conv_gn16 vs oracle {e_or:.3e} (un-fused {rel_rms(outs['0'][0], want):.3e}); vs the un-fused chain: eval {e_un:.3e}, dropout {e_dr:.3e}")
    assert e_or <= TOL["fp16"] and e_un <= 4e-3 and e_dr <= 4e-3


@pytest.mark.parametrize("dtype", DTYPES)
def test_nearest_upsample_folded_into_the_conv_gather_changes_no_bit(dtype, form_switch):
    """Round 6: `Upsample` of unet.py = nn.Upsample(scale_factor=2, mode="nearest") + 3x3 conv.  When that conv runs on
    conv_up_halo_kernel<5> its halo DMA reads the LOW-resolution tensor at (y >> 1, x >> 1) (ConvArgs::up_nearest): no materialised
    upsample, one launch less.  Same values through the same arithmetic: against DYF_FUSE_NEAREST=0 (up2x_nearest_vec_kernel + conv) the
    forward must agree BITWISE, eval and with the engine's MC dropout, on the OISST shapes (even planes 30 -> 60 and 15 -> 30 are not:
    15 x 15 -> 30 x 30 has an odd source, the upsampled plane is even) and on a plane that is ragged against the 16 x 32 tiles (28 x 60:
    7 x 15 -> 14 x 30 -> 28 x 60)."""
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
    for hw, nb in (((60, 60), 40), ((28, 60), 24)):
        g = torch.Generator().manual_seed(6)
        x, t = torch.randn(nb, 2, *hw, generator=g), (torch.arange(nb) % 7 + 1).float()
        outs = {}
        for fuse in ("1", "0"):
            form_switch.setenv("DYF_FUSE_NEAREST", fuse)
            form_switch.setenv("DYF_HALO5_MIN_TILES", "1")
            net = mirror(P, cfg, 2, 0, 1, dtype)
            net._own_engine(nb, hw)
            eng = net._engine
            eng.form_log(True)
            y = net(x.to(DEV), time=t.to(DEV)).cpu()
            forms = eng.form_log_read()
            eng.form_log(False)
            assert ("conv_up_halo_kernel<5>+nearest_up" in forms) == (fuse == "1"), sorted(forms)
            eng.seed(3)
            yd = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
            outs[fuse] = (y, yd)
            eng.close()
        assert torch.equal(outs["1"][0], outs["0"][0]) and torch.equal(outs["1"][1], outs["0"][1]), hw
        assert bool(torch.isfinite(outs["1"][0]).all()) and float(outs["1"][0].std()) > 0


def _invariant_engine(net, hw, max_batch, dtype="fp16"):
    from dyffusion_amd import _lib as L
    from dyffusion_amd.engine import HipEngine, upload_weights
    cfg = net.engine_net_config()
    eng = HipEngine(cfg, cfg, hw[0], hw[1], max_batch=max_batch, use_graph=False, dtype=dtype, batch_invariant=True, row_groups=1)
    upload_weights(net, eng, L.NET_FORECASTER)
    return eng


@pytest.mark.parametrize("force_igemm2", [False, True], ids=["default-conv-forms", "second-igemm-form"])
def test_batch_invariant_resnet_rows_do_not_depend_on_their_position_on_ragged_planes(force_igemm2, form_switch):
    """conv_igemm2_kernel<2, true> on flattened-M tiles reduces the GroupNorm statistics per 128-row slab of the
    n * plane axis; with plane % 128 != 0 (OISST levels 60 x 60, 30 x 30, 15 x 15) a sample's slab partition -- and so the last bits
    of (mean, 1/std) -- depended on its position in the launch.  A batch_invariant engine now keeps to position-free forms (halo5,
    2-D tiles, plane % 128 == 0; otherwise the three-kernel path): rows [2, 5) of a 6-row launch, run alone at positions 0..2 with
    their global row offset, must reproduce bit for bit, eval and with the engine's MC dropout."""
    if force_igemm2:
        form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")  # the large-batch forms (what 300 rows select) on this small launch
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
    net = mirror(P, cfg, 2, 0, 1, "fp16")
    eng = _invariant_engine(net, (60, 60), 6)
    g = torch.Generator().manual_seed(16)
    x, t = torch.randn(6, 2, 60, 60, generator=g).to(DEV), (torch.arange(6) % 7 + 1).float().to(DEV)
    eng.form_log(True)
    full = eng.net_forward(0, x, t, None).clone()
    forms = eng.form_log_read()
    eng.form_log(False)
    print(sorted(forms))
    if force_igemm2:  # the 15 x 15 and 30 x 30 levels would take the fused flattened-M form; the invariant engine must not
        assert "conv_igemm2_kernel<2>+gn_fused" not in forms, sorted(forms)
    part = eng.net_forward(0, x[2:5].contiguous(), t[2:5].contiguous(), None)
    assert torch.equal(full[2:5], part), float((full[2:5] - part).abs().max())
    eng.seed(3)
    eng.set_row_offset(0)
    full_d = eng.net_forward(0, x, t, None, dropout_mode=1).clone()
    eng.seed(3)
    eng.set_row_offset(2)
    part_d = eng.net_forward(0, x[2:5].contiguous(), t[2:5].contiguous(), None, dropout_mode=1)
    assert torch.equal(full_d[2:5], part_d), float((full_d[2:5] - part_d).abs().max())
    assert not torch.equal(full_d, full)
    eng.close()


def test_fused_groupnorm_timeout_fails_the_same_call_and_the_engine_recovers_unfused(form_switch):
    """the time-out path of the fused GroupNorm (csrc/gn_fused.h).  With the drill switched on
    (dyf_debug_gn_fuse: every granule sweep waits for a tag nobody publishes, bound 20 ms instead of 2 s) the launch must still
    TERMINATE, its output must be NaN-poisoned, and the failure must surface in the SAME call (the wrapper polls dyf_poll_errors after
    the work completed) -- not in the next one.  The engine then runs the three-kernel GroupNorm path: the repeated call is correct
    (equal to an engine that never fused, bit for bit) and the fused form stays off."""
    from dyffusion_amd.engine import EngineError
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
    g = torch.Generator().manual_seed(26)
    nb = 8
    x, t = torch.randn(nb, 2, 60, 60, generator=g).to(DEV), (torch.arange(nb) % 7 + 1).float().to(DEV)
    net = mirror(P, cfg, 2, 0, 1, "fp16")
    net._own_engine(nb, (60, 60))
    eng = net._engine
    assert eng.gn_fuse_state() == (True, 0)
    eng.form_log(True)
    good = net(x, time=t).clone()
    forms = eng.form_log_read()
    assert any(k.endswith("+gn_fused") for k in forms), sorted(forms)
    eng.debug_gn_fuse(timeout_ticks=2_000_000, force_timeout=True)  # 20 ms
    import time as _time
    t0 = _time.perf_counter()
    with pytest.raises(EngineError, match="timed out"):
        net(x, time=t)
    dt = _time.perf_counter() - t0
    print(f"forced time-out: the call failed after {dt:.2f} s")
    assert dt < 30.0
    live, downgrades = eng.gn_fuse_state()
    assert (live, downgrades) == (False, 1)
    eng.debug_gn_fuse(0, False)
    eng.form_log(True)
    again = net(x, time=t).clone()  # no exception: the failure was consumed by the call that caused it
    forms = eng.form_log_read()
    eng.form_log(False)
    assert not any(k.endswith("+gn_fused") for k in forms), sorted(forms)
    assert bool(torch.isfinite(again).all())
    assert rel_rms(again.cpu(), good.cpu()) <= 4e-3
    eng.close()
    # the poisoned output itself: a second engine, polling switched off for one call
    net3 = mirror(P, cfg, 2, 0, 1, "fp16")
    net3._own_engine(nb, (60, 60))
    e3 = net3._engine
    e3.debug_gn_fuse(timeout_ticks=2_000_000, force_timeout=True)
    e3.poll_errors = lambda *a, **k: None  # (instance attribute shadows the method)
    bad = net3(x, time=t)
    torch.cuda.synchronize()
    assert bool(torch.isnan(bad).all()), "a timed-out sweep must poison what it could not normalise"
    del e3.poll_errors
    with pytest.raises(EngineError, match="timed out"):
        e3.poll_errors()
    e3.poll_errors()  # reported once
    e3.close()
    form_switch.setenv("DYF_GN_FUSED", "0")
    try:
        net2 = mirror(P, cfg, 2, 0, 1, "fp16")
        never = net2(x, time=t)
        assert net2._engine.gn_fuse_state()[0] is False
        assert torch.equal(never, again)
        net2._engine.close()
    finally:
        form_switch.delenv("DYF_GN_FUSED")


def test_small_tile_fused_igemm_form_matches_the_large_tile_form_and_the_oracle(form_switch):
    """Round 5: conv_igemm2_kernel<2, true, *, 128> -- the fused-GroupNorm implicit GEMM on 128-pixel tiles (64-row statistics slabs)
    that the launcher takes while the 256-pixel tiles would leave CUs idle: the 15 x 15 level of the OISST shapes at 38 rows (68
    large tiles).  Against the 256-pixel form (DYF_IGEMM2_BM128_BELOW=0, read per launch), eval and with the engine's MC dropout
    (same masks: the streams are keyed by element), against the fp32 oracle on a few rows, and bit-reproducible run to run."""
    cfg = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.3, block_dropout1=0.1, attn_dropout=0.0,
               resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)
    P = seeded_unet(64, (1, 2, 4), 2, 1, seed=5)
    g = torch.Generator().manual_seed(36)
    nb = 38
    x, t = torch.randn(nb, 2, 60, 60, generator=g), (torch.arange(nb) % 7 + 1).float()
    form_switch.setenv("DYF_GN16", "0")  # (since round 6 conv_gn16_kernel takes every fused conv by default: this is the implicit-GEMM pair behind it)
    net = mirror(P, cfg, 2, 0, 1, "fp16")
    net._own_engine(nb, (60, 60))
    eng = net._engine
    outs = {}
    for below in ("0", None):
        if below is None:
            form_switch.delenv("DYF_IGEMM2_BM128_BELOW", raising=False)
        else:
            form_switch.setenv("DYF_IGEMM2_BM128_BELOW", below)
        eng.form_log(True)
        y_eval = net(x.to(DEV), time=t.to(DEV)).cpu()
        forms = eng.form_log_read()
        eng.form_log(False)
        small = "conv_igemm2_kernel<2,bm128>+gn_fused" in forms
        assert small == (below is None), sorted(forms)
        assert "conv_igemm2_kernel<2>+gn_fused" in forms
        eng.seed(11)
        y_d = eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu()
        eng.seed(11)
        assert torch.equal(y_d, eng.net_forward(0, x.to(DEV), t.to(DEV), None, dropout_mode=1).cpu())
        assert torch.equal(y_eval, net(x.to(DEV), time=t.to(DEV)).cpu())
        outs[below] = (y_eval, y_d)
    for i, nm in enumerate(("eval", "engine dropout")):
        e = max(rel_rms(outs[None][i][r], outs["0"][i][r]) for r in range(nb))
        print(f"128- vs 256-pixel tiles, {nm}: worst row rel-RMS {e:.3e}")
        assert e <= 4e-3
    rows = [0, 17, 37]
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, cfg, x[rows], t[rows], None)
    err = max(rel_rms(outs[None][0][r], want[i]) for i, r in enumerate(rows))
    print(f"128-pixel tiles vs the fp32 oracle: {err:.3e}")
    assert err <= TOL["fp16"]
    eng.close()
