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
