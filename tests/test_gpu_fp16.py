"""-m gpu: the fp16 build of the engine (libdyffusion_hip_f16.so: fp16 storage + v_mfma_f32_32x32x16_f16, same sources and
ABI as the bf16 library) and BASELINE configs[4] "Synthetic 512x512x4ch grid, h=32, fp16 MFMA conv/attn".

fp16 carries 11 mantissa bits instead of bf16's 8: the stated tolerances are rel-RMS <= 2.5e-3 per forward and <= 1e-2 per
field over an h=16 rollout (SURVEY 8c's bound for a 16-bit engine), measured values printed.
  * NS full-size forward + rollout (fixture G6) in fp16;
  * ResNet-UNet at the OISST shape in fp16;
  * config 5: ONE 512^2 ResNet-UNet forward (8 input channels, dim 64, mults (1,2,4); bottleneck Attention over 128^2 =
    16 384 tokens on the MFMA flash kernel) against the oracle, fp16 and bf16; the Attention core alone at 16 384 tokens
    against a plain fp32 PyTorch restatement; an h=32 rollout at 512^2: finite, rows independent, graph replay == eager.
"""
import math

import pytest
import torch

import dyffusion_amd as D
from oracle import init as oinit
from oracle import nets
from tests.gpu_common import DEV, seeded_pair
from tests.helpers import jload, load_npz, rel_rms

pytestmark = pytest.mark.gpu


def _unet_simple(P, mk, n_in, n_cond, n_out, dtype):
    net = D.UNet(dim=mk["dim"], with_time_emb=True, upsample_dims=mk.get("upsample_dims"), dropout=mk.get("dropout", 0.0),
                 num_input_channels=n_in, num_output_channels=n_out, num_conditional_channels=n_cond)
    net.engine_dtype = dtype
    net.load_state_dict(P, strict=True)
    return net


def test_fp16_fullsize_ns_forward_and_rollout():
    meta, fields = jload("fullsize_checksums.json"), load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF, PI = seeded_pair(64, 3, 2, seeds=(meta["seeds"]["forecaster"], meta["seeds"]["interpolator"]))
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0, c = torch.randn(1, 3, 221, 42, generator=g), torch.rand(1, 2, 221, 42, generator=g)
    F_, I_ = _unet_simple(PF, mk, 3, 2, 3, "fp16"), _unet_simple(PI, mk, 6, 2, 3, "fp16")
    yF = F_(x0.to(DEV), time=torch.tensor([3.0]).to(DEV), condition=c.to(DEV)).cpu()
    eF = rel_rms(yF, fields["yF"])
    assert F_._engine.dtype == "fp16"
    print("fp16 full-size forecaster forward rel-rms", eF)
    assert eF <= 2.5e-3
    hp = dict(timesteps=16, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=False)
    m = D.DYffusion(F_, D.InterpolatorHandle(I_, 16), max_batch=2, dtype="fp16", **hp)
    out = m.sample(x0.to(DEV), static_condition=c.to(DEV))
    errs = {k: rel_rms(out[f"{k}_preds"].cpu(), fields[k]) for k in ("t1", "t8", "t16")}
    print("fp16 full-size h=16 rollout rel-rms", errs)
    assert max(errs.values()) <= 1e-2


def _seeded_unet(dim, mults, cin, cout, seed, **kw):
    net = D.Unet(dim=dim, dim_mults=mults, with_time_emb=True, num_input_channels=cin, num_output_channels=cout, **kw)
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    st = oinit.seeded_state(shapes, seed, gain=1.0)
    for k in st:
        if k.endswith(".norm.g"):
            st[k] = 1.0 + 0.1 * torch.randn(shapes[k], generator=torch.Generator().manual_seed(len(k)))
    net.load_state_dict(st, strict=True)
    return net, st


CFG = dict(dim=64, dim_mults=[1, 2, 4], with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0,
           resnet_block_groups=8, input_dropout=0.0, upsample_dims=None)


def test_fp16_resnet_unet_oisst_shape():
    net, P = _seeded_unet(64, (1, 2, 4), 2, 1, seed=51)
    net.engine_dtype = "fp16"
    g = torch.Generator().manual_seed(6)
    x, t = torch.randn(2, 2, 60, 60, generator=g), torch.tensor([1.0, 4.5])
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, CFG, x, t, None)
    err = rel_rms(net(x.to(DEV), time=t.to(DEV)).cpu(), want)
    print("fp16 resnet-unet 60x60 rel-rms", err)
    assert err <= 4e-3


@pytest.mark.parametrize("dtype,tol", [("fp16", 4e-3), ("bf16", 2e-2)])
def test_config5_512x512_forward_matches_oracle(dtype, tol):
    """BASELINE configs[4]: one interpolator forward at 512^2 (925 GFLOP; 148 of them in the bottleneck Attention over
    16 384 tokens, which the oracle materialises as a 4 x 16 384^2 fp32 matrix)."""
    net, P = _seeded_unet(64, (1, 2, 4), 8, 4, seed=81)
    net.engine_dtype = dtype
    g = torch.Generator().manual_seed(16)
    x, t = torch.randn(1, 8, 512, 512, generator=g), torch.tensor([3.0])
    with torch.no_grad():
        want = nets.resnet_unet_forward(P, CFG, x, t, None)
    got = net(x.to(DEV), time=t.to(DEV)).cpu()
    err = rel_rms(got, want)
    print(f"config 5 ({dtype}) 512^2 forward rel-rms {err:.3e}")
    assert bool(torch.isfinite(got).all()) and err <= tol


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_attention_core_16384_tokens(dtype):
    """Attention (attention.py:62-72) over 128^2 = 16 384 tokens, 4 heads x 32: flash MFMA kernel vs plain fp32 PyTorch on
    the same 16-bit operands (the 16 384^2 score matrix is materialised head by head on the GPU)."""
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    eng = D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False, dtype=dtype)
    g = torch.Generator().manual_seed(2)
    n = 16384
    qkv = (torch.randn(1, n, 384, generator=g) * 1.5).to(DEV).to(eng.torch_dtype)
    got = eng.op_attention(qkv).float()
    q, k, v = (qkv.float()[0, :, i * 128:(i + 1) * 128].reshape(n, 4, 32).permute(1, 0, 2) for i in range(3))
    want = torch.empty(4, n, 32, device=DEV)
    for h in range(4):
        want[h] = torch.softmax(q[h] @ k[h].T * (32 ** -0.5), dim=-1) @ v[h]
    want = want.permute(1, 0, 2).reshape(1, n, 128)
    err = rel_rms(got.cpu(), want.cpu())
    print(f"attention core, 16 384 tokens ({dtype}) rel-rms {err:.3e}")
    assert err <= (1.5e-3 if dtype == "fp16" else 8e-3)


def _attention_reference(qkv):
    n = qkv.shape[1]
    q, k, v = (qkv.float()[0, :, i * 128:(i + 1) * 128].reshape(n, 4, 32).permute(1, 0, 2) for i in range(3))
    want = torch.stack([torch.softmax(q[h] @ k[h].T * (32 ** -0.5), dim=-1) @ v[h] for h in range(4)])
    return want.permute(1, 0, 2).reshape(1, n, 128)


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_attention_core_sequence_lengths_and_score_ranges(dtype):
    """The pipelined flash kernel (flash_attention4_kernel) against plain fp32 PyTorch on the same 16-bit operands:
      * token counts 1 ... 1 100 (no whole 64-key tile; whole tiles only; whole tiles + a partial last tile of <= 32 and > 32 keys;
        four- and eight-wave workgroups: >= 512 tokens);
      * score ranges that take each branch of its lazy maximum: |scores| small (m stays exactly 0: no bias products), scores with
        a spread of +-90 in the log2 domain (the sum check fires, m moves, the scores in flight are shifted), every score near
        -165 (m != 0 from the first sub-tile on), every score near +165."""
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    eng = D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False, dtype=dtype)
    tol = 2e-3 if dtype == "fp16" else 1e-2
    worst = 0.0
    for n in (1, 31, 64, 100, 128, 200, 512, 577, 1100):
        for case in ("small", "spread", "low", "high"):
            g = torch.Generator().manual_seed(1000 * n + len(case))
            x = torch.randn(1, n, 384, generator=g)
            if case == "spread":
                x[:, :, :256] *= 4.0
            elif case in ("low", "high"):  # q near +4.5, k near -+4.5 in every channel: q.k * 32^-1/2 * log2(e) near -+165
                x[:, :, :128] = 4.5 + 0.1 * x[:, :, :128]  # (below -128 the first rescale factor 2^-delta would overflow)
                x[:, :, 128:256] = (-4.5 if case == "low" else 4.5) + 0.1 * x[:, :, 128:256]
            qkv = x.to(DEV).to(eng.torch_dtype)
            got = eng.op_attention(qkv).float()
            assert bool(torch.isfinite(got).all()), (n, case)
            err = rel_rms(got.cpu(), _attention_reference(qkv).cpu())
            worst = max(worst, err)
            assert err <= tol, (n, case, err)
    print(f"attention core, 9 token counts x 4 score ranges ({dtype}): worst rel-rms {worst:.3e}")


def test_config5_h32_rollout_properties():
    """h=32 rollout at 512^2 x 4 channels (fp16, dropout on in the interpolator): every field finite, rows independent
    (2-row batch == two 1-row batches with row offsets; engines created with batch_invariant=True: kernel forms chosen for max_batch), hipGraph replay == eager launches, bit for bit."""
    F_, _ = _seeded_unet(64, (1, 2, 4), 4, 4, seed=91)
    I_, _ = _seeded_unet(64, (1, 2, 4), 8, 4, seed=92, block_dropout=0.1, attn_dropout=0.1)
    hp = dict(timesteps=32, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              sampling_type="cold", refine_intermediate_predictions=False, enable_interpolator_dropout=True)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(2, 4, 512, 512, generator=g).to(DEV)

    def run(rows, offset, use_graph):
        m = D.DYffusion(F_, D.InterpolatorHandle(I_, 32), max_batch=2, dtype="fp16", use_graph=use_graph, batch_invariant=True, **hp)
        m.seed(5)
        m.set_row_offset(offset)
        return {k: v.clone() for k, v in m.sample(x0[rows]).items()}

    full = run(slice(0, 2), 0, True)
    assert len(full) == 32 and all(bool(torch.isfinite(v).all()) for v in full.values())
    eager = run(slice(0, 2), 0, False)
    r0, r1 = run(slice(0, 1), 0, True), run(slice(1, 2), 1, True)
    for k in full:
        assert torch.equal(full[k], eager[k]), k
        assert torch.equal(full[k][:1], r0[k]) and torch.equal(full[k][1:], r1[k]), k
    assert not torch.equal(full["t32_preds"][0], full["t32_preds"][1])


@pytest.mark.parametrize("p", [0.1, 0.3, 0.6])
def test_attention_probability_dropout_quad_form_keep_rate_and_unbiasedness(p):
    """Round 5: the engine's dropout on the softmax probabilities draws ONE 32-bit word per four consecutive keys and compares 8 bits
    each with k = floor((1 - p) * 256) (csrc/common.h rng_keep8): the keep rate is k / 256 -- within 2^-8 of 1 - p -- and survivors are
    scaled by 256 / k, so E[dropout(P)] = P exactly as for nn.Dropout.  With V = one-hot rows the output IS the (dropped, scaled)
    probability mass per key block: checked here through uniform attention (q = 0: P = 1 / N) and a V whose channel c marks the
    keys j % 32 == c -- out[i, c] = scale * (kept keys of class c) / N."""
    import dyffusion_amd as D
    cfg = D.resnet_net_config(in_channels=2, cond_channels=0, out_channels=1, dim=64, dim_mults=(1, 2))
    eng = D.HipEngine(cfg, cfg, 16, 16, max_batch=2, use_graph=False, dtype="fp16")
    n, N = 2, 1024
    qkv = torch.zeros(n, N, 3 * 128, dtype=torch.float16)
    cls = torch.arange(N) % 32
    for h in range(4):
        qkv[:, torch.arange(N), 2 * 128 + h * 32 + cls] = 1.0  # V: one-hot class of the key
    eng.seed(7)
    out = eng.op_attention(qkv.to(DEV), p_drop=p).float().cpu()  # (n, N, 128): per query the kept mass per class, scaled
    k = max(1, int((1.0 - p) * 256.0))
    keep_rate, scale = k / 256.0, 256.0 / k
    per_query = out.reshape(n, N, 4, 32).sum(-1)  # scale * kept / N per (row, query, head)
    mean = float(per_query.mean())
    # n * N * 4 queries x N keys each: the mean kept fraction has a standard error of sqrt(r (1 - r) / 8.4e6) ~ 1.6e-4
    kept_frac = mean / scale
    print(f"p={p}: kept fraction {kept_frac:.5f} (k/256 = {keep_rate:.5f}, 1-p = {1 - p:.5f}); E[scaled mass] = {mean:.5f}")
    assert abs(kept_frac - keep_rate) < 1.2e-3, (kept_frac, keep_rate)
    assert abs(keep_rate - (1.0 - p)) <= 1.0 / 256.0
    assert abs(mean - 1.0) < 2.5e-3  # unbiased: the scale is the inverse of the ACTUAL keep rate (fp16 output rounding included)
    # rows / heads / queries draw different masks
    assert float(per_query.std()) > 0
    eng.close()
