"""LinearAttention core on the MFMA kernels (linattn_ctx_mfma / merge / out_mfma) against a torch fp32 restatement of
attention.py:28-49 (reference: softmax over d for q, over n for k, q * scale, v / (h*w), context = k v^T, out = context^T q)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv):
    n, hw, _ = qkv.shape
    q, k, v = qkv.float().view(n, hw, 3, 4, 32).permute(2, 0, 3, 4, 1)  # each (n, heads, 32, hw)
    q = q.softmax(dim=-2) * 32 ** -0.5
    k = k.softmax(dim=-1)
    v = v / hw
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q)  # (n, heads, 32, hw)
    return out.permute(0, 3, 1, 2).reshape(n, hw, 128)


@pytest.fixture(scope="module")
def engine():
    import dyffusion_amd as D
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    return D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False)


# hw: one partial wave, exactly one wave, several workgroups with idle / partial tail waves (3600 = OISST 60 x 60)
@pytest.mark.parametrize("n,hw", [(2, 64), (1, 256), (3, 225), (2, 1024), (2, 3600), (1, 1032), (1, 5000)])
def test_linear_attention_matches_fp32_reference(engine, n, hw):
    g = torch.Generator().manual_seed(n * 7919 + hw)
    qkv = (1.5 * torch.randn(n, hw, 384, generator=g)).to(torch.bfloat16)
    qkv[:, :, 128:160] += 3.0 * torch.randn(n, 1, 32, generator=g).to(torch.bfloat16)  # per-channel offsets in k (head 0)
    got = engine.op_linear_attention(qkv.cuda()).float().cpu()
    want = _ref(qkv)
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -8 * want.abs().max().item() + 1e-6, err  # output is bf16: half an ulp of the largest value


def _ref_block(xn, xres, wqkv, wout, bout):
    """to_qkv (1x1, no bias) -> the core above -> to_out (1x1 + bias) + residual, all fp32 (attention.py:22-49, unet.py Residual)."""
    qkv = xn.float() @ wqkv.t()
    return _ref(qkv) @ wout.t() + bout + xres.float()


# hw: a single pixel, one short / one exact / one-and-a-bit 32-pixel group, the OISST levels (15^2, 30^2, 60^2 = several
# workgroups with a 16-pixel tail group), exactly one workgroup, one pixel more, a workgroup whose later waves are idle
@pytest.mark.parametrize("c", [64, 128])
@pytest.mark.parametrize("n,hw", [(2, 1), (2, 31), (1, 32), (3, 33), (3, 225), (2, 900), (1, 1024), (1, 1025), (2, 3600), (1, 5000)])
def test_fused_linear_attention_block_matches_fp32_reference(engine, n, hw, c):
    g = torch.Generator().manual_seed(n * 7919 + hw + c)
    xn = torch.randn(n, hw, c, generator=g).to(torch.bfloat16)       # LayerNorm output: unit variance per pixel
    xres = (2.0 * torch.randn(n, hw, c, generator=g)).to(torch.bfloat16)
    wqkv = torch.randn(384, c, generator=g) * (1.5 / c ** 0.5)
    wqkv[128:160] *= 3.0                                               # a head whose k logits spread widely over the pixels
    wout = torch.randn(c, 128, generator=g) * (8.0 / 128 ** 0.5)       # the core output is O(1 / hw ... 1): keep to_out visible
    bout = torch.randn(c, generator=g)
    want = _ref_block(xn, xres, wqkv, wout, bout)
    att = want - xres.float() - bout                                   # the attention branch alone
    # (1) the branch by itself (zero residual and bias): its own 16-bit operands (weights, v, k', q', core output) and the 16-bit store
    alone = engine.op_linear_attention_fused(xn.cuda(), torch.zeros_like(xres).cuda(), wqkv, wout, torch.zeros_like(bout)).float().cpu()
    assert torch.isfinite(alone).all()
    branch_err = (alone - att).abs().max().item()
    print(f"fused linattn c={c} n={n} hw={hw}: branch max {att.abs().max().item():.3e}, err {branch_err:.3e}")
    assert branch_err <= 1.5e-2 * att.abs().max().item(), branch_err
    # (2) with bias and residual: additionally half an ulp of the largest stored value
    got = engine.op_linear_attention_fused(xn.cuda(), xres.cuda(), wqkv, wout, bout).float().cpu()
    assert torch.isfinite(got).all()
    err = (got - want).abs().max().item()
    assert err <= 2.0 ** -8 * want.abs().max().item() + 1.5e-2 * att.abs().max().item() + 1e-6, err


@pytest.mark.parametrize("c,n,hw", [(64, 2, 3600), (128, 3, 900), (64, 1, 5000), (128, 2, 225)])
def test_fused_linear_attention_block_sizes_agree(engine, c, n, hw, form_switch):
    """Round 5: the fused block's workgroups walk 32, 16 or 8 groups of 32 pixels (DYF_LINATTN_GPB, read per launch; the launcher takes
    the smaller blocks while a launch is under 512 workgroups -- the few-rows regime).  The blocks only change how the online softmax
    over the pixels is cut into partials: every size within the tolerance of the fp32 reference, and within rounding of one another."""
    g = torch.Generator().manual_seed(n * 31 + hw + c)
    xn = torch.randn(n, hw, c, generator=g).to(torch.bfloat16)
    xres = (2.0 * torch.randn(n, hw, c, generator=g)).to(torch.bfloat16)
    wqkv = torch.randn(384, c, generator=g) * (1.5 / c ** 0.5)
    wqkv[128:160] *= 3.0
    wout = torch.randn(c, 128, generator=g) * (8.0 / 128 ** 0.5)
    bout = torch.randn(c, generator=g)
    want = _ref_block(xn, xres, wqkv, wout, bout)
    att = want - xres.float() - bout
    outs = {}
    for gpb in ("32", "16", "8"):
        form_switch.setenv("DYF_LINATTN_GPB", gpb)
        engine.form_log(True)
        outs[gpb] = engine.op_linear_attention_fused(xn.cuda(), xres.cuda(), wqkv, wout, bout).float().cpu()
        forms = engine.form_log_read()
        engine.form_log(False)
        assert f"linattn_fused_kernels<gpb={gpb}>" in forms, sorted(forms)
        err = (outs[gpb] - want).abs().max().item()
        assert err <= 2.0 ** -8 * want.abs().max().item() + 1.5e-2 * att.abs().max().item() + 1e-6, (gpb, err)
    for gpb in ("16", "8"):
        d = (outs[gpb] - outs["32"]).abs().max().item()
        assert d <= 2.0 ** -7 * want.abs().max().item(), (gpb, d)
