"""-m gpu: the conv kernels (direct and MFMA implicit GEMM) against torch.conv2d on the same bf16-rounded operands.

Through the C ABI test seam dyf_op_conv2d.  Tolerance: output is rounded to bf16 (rel 2^-9) after an fp32
accumulation whose order differs from ATen's; we require max |err| <= 1.5 * 2^-8 * max|y| + 1e-3 and rel-RMS <= 4e-3.
"""
import pytest
import torch
import torch.nn.functional as F

from tests.helpers import max_abs, rel_rms

pytestmark = pytest.mark.gpu

CASES = [
    # n, h, w, cin, cout, k, stride, pad
    (2, 16, 16, 64, 128, 3, 1, 1),
    (1, 32, 32, 64, 128, 4, 2, 1),
    (3, 8, 8, 128, 64, 2, 2, 0),
    (2, 8, 8, 192, 64, 1, 1, 0),
    (1, 20, 12, 128, 256, 3, 1, 1),     # M = 240: not a multiple of the 128-row tile
    (5, 4, 4, 512, 512, 2, 2, 0),       # tiles span several samples
    (1, 64, 64, 256, 64, 3, 1, 1),      # 256x64 tile variant
    (2, 9, 7, 64, 64, 3, 1, 1),
    # plain 3x3 convs with 64 / 128 output channels on the halo kernel's SP = 5 form (>= 256 tiles of 16 x 32 pixels): the
    # ResNet-UNet levels at the OISST plane sizes (ragged: 60 = 3*16 + 12 = 32 + 28), a plane narrower than a tile, two chunks
    (32, 60, 60, 64, 64, 3, 1, 1),
    (70, 30, 30, 128, 128, 3, 1, 1),
    (140, 20, 37, 64, 128, 3, 1, 1),
    (300, 17, 9, 128, 64, 3, 1, 1),
]


@pytest.fixture(scope="module")
def engine():
    import dyffusion_amd as D
    cfg = D.net_config(in_channels=3, cond_channels=0, out_channels=3, dim=64, upsample_dims=[64, 64])
    return D.HipEngine(cfg, cfg, 16, 16, max_batch=1, use_graph=False)


def reference(x_nhwc, w, stride, pad, scale=None, shift=None, act=0):
    x = x_nhwc.float().permute(0, 3, 1, 2)
    wq = w.to(torch.bfloat16).float()
    y = F.conv2d(x, wq, None, stride, pad)
    if scale is not None:
        y = y * scale[:, :, None, None] + shift[:, :, None, None]
    if act == 1:
        y = F.relu(y)
    elif act == 2:
        y = F.leaky_relu(y, 0.2)
    return y.permute(0, 2, 3, 1)


@pytest.mark.parametrize("path", [0, 1], ids=["direct", "mfma"])
@pytest.mark.parametrize("case", CASES, ids=lambda c: "x".join(map(str, c)))
def test_conv_matches_torch(engine, case, path):
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(hash(case) % 1000)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    for act, use_coef in [(0, False), (2, True), (1, True)]:
        y = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda() if use_coef else None,
                             shift.cuda() if use_coef else None, act=act, path=path)
        want = reference(x, wt, stride, pad, scale if use_coef else None, shift if use_coef else None, act)
        got = y.float().cpu()
        tol = 1.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
        assert max_abs(got, want) <= tol, (case, act, max_abs(got, want), tol)
        assert rel_rms(got, want) <= 4e-3


IGEMM2_CASES = [
    (2, 16, 16, 64, 128, 3, 1, 1),
    (1, 32, 32, 64, 128, 4, 2, 1),
    (1, 20, 12, 128, 256, 3, 1, 1),     # M = 240 < one 256-row tile, raster tiling
    (5, 4, 4, 512, 512, 2, 2, 0),       # a tile spans several samples
    (3, 33, 17, 64, 128, 3, 1, 1),      # odd plane: raster tiling, 7 tiles with a ragged last one
    (1, 64, 64, 128, 128, 1, 1, 0),
    (2, 32, 48, 192, 256, 3, 1, 1),     # 2-D tiles (16 x 16 output tiles), three 64-channel chunks
    # cout % 128 != 0 -> conv_igemm2_kernel<1> (256 x 64 tiles, waves 4 x 1)
    (2, 16, 16, 64, 64, 3, 1, 1),
    (1, 60, 60, 64, 64, 3, 1, 1),       # the OISST level-0 plane: raster tiling, ragged last tile
    (3, 15, 15, 128, 192, 3, 1, 1),
    (1, 32, 32, 64, 320, 1, 1, 0),
    # 3x3 / stride 1 / pad 1 on raster tiles = the SH3 form (one gather per window row, dx = 0 / 2 from the neighbouring LDS rows):
    # several tiles with sample boundaries inside, 2 and 4 chunks, tile rows that start / end mid image row, a 2-pixel-wide plane
    (7, 30, 30, 128, 128, 3, 1, 1),
    (9, 15, 15, 256, 256, 3, 1, 1),
    (5, 23, 2, 64, 128, 3, 1, 1),
    (2, 16, 19, 128, 64, 3, 1, 1),
]


@pytest.mark.parametrize("case", IGEMM2_CASES, ids=lambda c: "x".join(map(str, c)))
def test_second_igemm_form_matches_torch(engine, case, form_switch):
    """conv_igemm2_kernel (256 x 128 tiles, weights streamed in fragment order) is selected by tile count in production;
    DYF_IGEMM2_MIN_TILES=1 forces it on these small problems.  Checked against torch and against the 128 x 128 form."""
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1000000000")
    first = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda(), shift.cuda(), act=2, path=1).float().cpu()
    form_switch.setenv("DYF_IGEMM2_MIN_TILES", "1")
    for act, use_coef in [(0, False), (2, True), (1, True)]:
        y = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda() if use_coef else None,
                             shift.cuda() if use_coef else None, act=act, path=1)
        want = reference(x, wt, stride, pad, scale if use_coef else None, shift if use_coef else None, act)
        got = y.float().cpu()
        tol = 1.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
        assert max_abs(got, want) <= tol, (case, act, max_abs(got, want), tol)
        assert rel_rms(got, want) <= 4e-3
        if act == 2:
            assert rel_rms(got, first) <= 2.5e-3  # the two MFMA forms differ by summation order only
            if k == 3 and stride == 1 and pad == 1 and not (w % 16 == 0 and h % 16 == 0):
                # the SH3 form ran: it must reproduce the one-gather-per-tap form BIT FOR BIT (same operands, same summation order)
                form_switch.setenv("DYF_IGEMM2_SH3", "0")
                per_tap = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda(), shift.cuda(), act=act, path=1).float().cpu()
                form_switch.delenv("DYF_IGEMM2_SH3")
                assert torch.equal(got, per_tap), float((got - per_tap).abs().max())


HALO3_CASES = [(2, 16, 16, 64, 256, 3, 1, 1), (1, 32, 48, 128, 256, 3, 1, 1), (3, 8, 16, 192, 512, 3, 1, 1),
               (1, 64, 32, 64, 256, 3, 1, 1),
               # planes that tile by 4 x 32 run the rows form (conv_halo_rows_kernel<2>): 3 chunks, 2 column tiles, 2 blocks
               (2, 32, 64, 192, 512, 3, 1, 1), (1, 8, 32, 64, 256, 3, 1, 1),
               # 4 x 4 / stride 2 / pad 1 on the space-to-depth view (conv_up_halo_kernel<3>)
               (2, 16, 32, 64, 256, 4, 2, 1), (1, 64, 64, 128, 256, 4, 2, 1), (3, 32, 32, 256, 512, 4, 2, 1),
               # ... with 128-channel workgroups on 16 x 16 tiles (conv_up_halo_kernel<4>)
               (2, 32, 32, 64, 128, 4, 2, 1), (1, 64, 96, 128, 384, 4, 2, 1), (3, 128, 128, 64, 128, 4, 2, 1)]


@pytest.mark.parametrize("case", HALO3_CASES, ids=lambda c: "x".join(map(str, c)))
def test_plain_convs_on_the_halo_kernel_match_torch(engine, case, form_switch):
    """conv_up_halo_kernel<2> / <3>: plain 3x3 / s1 / p1 and 4x4 / s2 / p1 convs (cout % 256 == 0) with the window in LDS
    and zero-filled borders; production uses them from 512 tiles on, DYF_HALO3_MIN_TILES=1 forces them here.  Borders
    checked separately."""
    n, h, w, cin, cout, k, stride, pad = case
    g = torch.Generator().manual_seed(sum(case) + 1)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, k, k, generator=g) / (cin * k * k) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_HALO3", "0")
    other = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda(), shift.cuda(), act=1, path=1).float().cpu()
    form_switch.setenv("DYF_HALO3", "1")
    form_switch.setenv("DYF_HALO3_MIN_TILES", "1")
    y = engine.op_conv2d(x.cuda(), wt, stride, pad, scale.cuda(), shift.cuda(), act=1, path=1).float().cpu()
    want = reference(x, wt, stride, pad, scale, shift, 1)
    tol = 1.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
    assert max_abs(y, want) <= tol
    assert rel_rms(y, want) <= 4e-3
    for sl in [(slice(None), 0), (slice(None), -1), (slice(None), slice(None), 0), (slice(None), slice(None), -1)]:
        assert rel_rms(y[sl], want[sl]) <= 5e-3, sl
    assert rel_rms(y, other) <= 2.5e-3  # vs the implicit-GEMM forms: summation order only


def test_mfma_and_direct_agree_closely(engine):
    # both accumulate in fp32 from identical bf16 operands: they differ only by summation order (+ final rounding)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(2, 24, 24, 128, generator=g).to(torch.bfloat16).cuda()
    wt = torch.randn(128, 128, 3, 3, generator=g) / 34.0
    a = engine.op_conv2d(x, wt, 1, 1, path=0).float()
    b = engine.op_conv2d(x, wt, 1, 1, path=1).float()
    assert rel_rms(a.cpu(), b.cpu()) <= 2.5e-3
    assert float((a != b).float().mean()) < 0.2  # most outputs round to the same bf16 value


def test_mfma_path_rejects_unsupported_channels(engine):
    x = torch.zeros(1, 8, 8, 32, dtype=torch.bfloat16).cuda()
    with pytest.raises(NotImplementedError):
        engine.op_conv2d(x, torch.zeros(64, 32, 3, 3), 1, 1, path=1)


UPCASES = [(2, 16, 16, 64, 128), (1, 32, 16, 128, 64), (3, 8, 16, 192, 128), (1, 16, 32, 64, 64), (2, 48, 32, 128, 256),
           # halo form + border-ring kernel: ragged ring tiles (47 and 48 pixels), 9 samples = one full group of 8 + 1,
           # corner workgroups with fewer than 64 samples; two input chunks from two sources are covered by the network tests
           (9, 32, 48, 64, 64), (1, 64, 32, 128, 128),
           # planes that tile by 4 x 32 run the rows form (conv_halo_rows_kernel<0>); w = 48 above stays on conv_up_halo_kernel<0>
           (2, 40, 64, 192, 128), (9, 32, 32, 64, 64)]


@pytest.mark.parametrize("case", UPCASES, ids=lambda c: "x".join(map(str, c)))
def test_fused_upsample_conv_matches_torch(engine, case):
    """Upsample(x2, bilinear) + Conv2d(3x3, pad 1) by phase decomposition (incl. the border correction taps) vs ATen."""
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case))
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    y = engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu()
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear")
    want = F.relu(F.conv2d(up, wt, None, 1, 1) * scale[:, :, None, None] + shift[:, :, None, None]).permute(0, 2, 3, 1)
    # the engine rounds the COMBINED stencil weights to bf16 (the reference rounds nothing): slightly looser than conv
    assert rel_rms(y, want) <= 6e-3, rel_rms(y, want)
    # borders and corners carry the correction taps: check them separately so a wrong correction cannot hide
    for sl in [(slice(None), 0), (slice(None), -1), (slice(None), slice(None), 0), (slice(None), slice(None), -1)]:
        assert rel_rms(y[sl], want[sl]) <= 8e-3, (sl, rel_rms(y[sl], want[sl]))
    assert max_abs(y, want) <= 3 * 2 ** -8 * float(want.abs().max()) + 2e-3


def _form_log(engine_obj, fn):
    engine_obj.form_log(True)
    out = fn()
    forms = engine_obj.form_log_read()
    engine_obj.form_log(False)
    return out, forms


# (n, h, w, cin, cout) of fused x2-upsample convs at FEW ROWS: the decoder shapes of unet_simple at one / two rows (dec3: 32 x 32
# low-res, 8 chunks; dec4: 64 x 64, 4 chunks) and shapes where the split is 2 / the ring has ragged tiles / three samples share the
# border kernel's sample pairs unevenly
SPLIT_UPCASES = [(1, 32, 32, 512, 128, 8), (2, 64, 64, 256, 128, 4), (3, 32, 64, 256, 64, 2), (5, 48, 32, 128, 64, 2), (1, 32, 32, 128, 128, 2),
                 (1, 32, 32, 512, 128, 0)]  # last entry: factor 0 = whatever the launcher's cost model picks (8 for dec3 at one row)


@pytest.mark.parametrize("case", SPLIT_UPCASES, ids=lambda c: "x".join(map(str, c)))
def test_rows_splitk_upsample_conv_matches_torch_and_the_unsplit_kernel(engine, case, form_switch):
    """Round 5: conv_halo_rows_splitk_kernel<0> (a tile's 64-channel chunks dealt to 2 / 4 / 8 workgroups, raw fp32 partials,
    conv_splitk_finish4_kernel) and up_border_split_kernel (the border ring's K chain dealt to the 8 waves of a workgroup) -- what a
    decoder launch of a few rows takes -- against ATen and against the one-workgroup-per-tile kernels they replace."""
    n, h, w, cin, cout, factor = case
    g = torch.Generator().manual_seed(sum(case[:5]) + 5)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_ROWS_TR", "4")  # (short tiles are another round-5 form: tested below)
    form_switch.setenv("DYF_HALO_SPLITK", "0")
    form_switch.setenv("DYF_UP_BORDER_SPLIT_ROWS", "0")
    form_switch.setenv("DYF_UP_BORDER_RING4", "0")
    old, forms0 = _form_log(engine, lambda: engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu())
    assert "conv_halo_rows_kernel<0>" in forms0 and not any("split" in k or "ring4" in k for k in forms0), sorted(forms0)
    # the many-rows form of the border ring (four samples per workgroup, a four-slot ring per wave, no K split): same sums, other order
    form_switch.setenv("DYF_UP_BORDER_RING4", "1")  # (opt-in: measured slower than the two-slot kernel at 80 rows, kept as an experiment)
    y4, forms4 = _form_log(engine, lambda: engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu())
    assert "up_border_ring4_kernel" in forms4, sorted(forms4)
    assert torch.equal(y4[:, 1:-1, 1:-1], old[:, 1:-1, 1:-1])  # the interior does not see the ring
    for sl in [(slice(None), 0), (slice(None), -1), (slice(None), slice(None), 0), (slice(None), slice(None), -1)]:
        assert rel_rms(y4[sl], old[sl]) <= 2.5e-3, (sl, rel_rms(y4[sl], old[sl]))
    form_switch.delenv("DYF_UP_BORDER_RING4")
    form_switch.delenv("DYF_HALO_SPLITK")
    form_switch.delenv("DYF_UP_BORDER_SPLIT_ROWS")
    if factor:
        form_switch.setenv("DYF_HALO_SPLITK_FORCE", str(factor))
    y, forms = _form_log(engine, lambda: engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu())
    assert "conv_halo_rows_kernel<0>+splitk" in forms and "up_border_split_kernel" in forms, sorted(forms)
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear")
    want = F.relu(F.conv2d(up, wt, None, 1, 1) * scale[:, :, None, None] + shift[:, :, None, None]).permute(0, 2, 3, 1)
    assert rel_rms(y, want) <= 6e-3, rel_rms(y, want)
    for sl in [(slice(None), 0), (slice(None), -1), (slice(None), slice(None), 0), (slice(None), slice(None), -1)]:
        assert rel_rms(y[sl], want[sl]) <= 8e-3, (sl, rel_rms(y[sl], want[sl]))
        assert rel_rms(y[sl], old[sl]) <= 2.5e-3, (sl, rel_rms(y[sl], old[sl]))  # the ring: summation order only
    assert max_abs(y, want) <= 3 * 2 ** -8 * float(want.abs().max()) + 2e-3
    assert rel_rms(y, old) <= 2.5e-3
    # ... and the split kernels reproduce themselves bit for bit (fixed summation order: splits / waves are added by index)
    y2 = engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu()
    assert torch.equal(y, y2)


@pytest.mark.parametrize("case", [(2, 32, 32, 1024, 256, 8), (1, 8, 64, 256, 512, 4), (3, 16, 32, 128, 256, 2)], ids=lambda c: "x".join(map(str, c)))
def test_rows_splitk_plain_conv_matches_torch_and_the_unsplit_kernel(engine, case, form_switch):
    """conv_halo_rows_splitk_kernel<2>: the plain 3x3 / 256-channel-block form (dec2 of unet_simple: 16 chunks, 8 tiles per row) at few
    rows, forced onto the rows kernel as the engine's tile thresholds would at >= 10 rows."""
    n, h, w, cin, cout, factor = case
    g = torch.Generator().manual_seed(sum(case[:5]) + 9)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_HALO3_MIN_TILES", "1")
    form_switch.setenv("DYF_ROWS_TR", "4")
    form_switch.setenv("DYF_HALO_SPLITK", "0")
    old, forms0 = _form_log(engine, lambda: engine.op_conv2d(x.cuda(), wt, 1, 1, scale.cuda(), shift.cuda(), act=2, path=1).float().cpu())
    assert "conv_halo_rows_kernel<2>" in forms0, sorted(forms0)
    form_switch.delenv("DYF_HALO_SPLITK")
    form_switch.setenv("DYF_HALO_SPLITK_FORCE", str(factor))
    y, forms = _form_log(engine, lambda: engine.op_conv2d(x.cuda(), wt, 1, 1, scale.cuda(), shift.cuda(), act=2, path=1).float().cpu())
    assert "conv_halo_rows_kernel<2>+splitk" in forms, sorted(forms)
    want = reference(x, wt, 1, 1, scale, shift, 2)
    tol = 1.5 * 2 ** -8 * float(want.abs().max()) + 1e-3
    assert max_abs(y, want) <= tol
    assert rel_rms(y, want) <= 4e-3
    for sl in [(slice(None), 0), (slice(None), -1), (slice(None), slice(None), 0), (slice(None), slice(None), -1)]:
        assert rel_rms(y[sl], want[sl]) <= 5e-3, sl
    assert rel_rms(y, old) <= 2.5e-3


@pytest.mark.parametrize("tr", [2, 1])
@pytest.mark.parametrize("case", [(2, 32, 32, 256, 128), (1, 64, 64, 128, 64), (3, 40, 64, 192, 128)], ids=lambda c: "x".join(map(str, c)))
def test_short_tile_rows_forms_reproduce_the_four_row_tiles_bitwise_upsample(engine, case, tr, form_switch):
    """Round 5: conv_halo_rows_tr_kernel<0, 2 / 1> -- the fused x2-upsample conv on tiles of 2 / 1 rows per wave (2 / 4 x the
    workgroups for under-filled launches).  Same operands, same K order per output element as the four-row tiles: bit-identical."""
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case) + 21)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_HALO_SPLITK", "0")
    form_switch.setenv("DYF_ROWS_TR", "4")
    ref, forms0 = _form_log(engine, lambda: engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu())
    assert "conv_halo_rows_kernel<0>" in forms0 and not any("+tr" in k for k in forms0), sorted(forms0)
    form_switch.setenv("DYF_ROWS_TR", str(tr))
    y, forms = _form_log(engine, lambda: engine.op_upconv2d(x.cuda(), wt, scale.cuda(), shift.cuda(), act=1).float().cpu())
    assert f"conv_halo_rows_kernel<0>+tr{tr}" in forms, sorted(forms)
    assert torch.equal(y, ref), float((y - ref).abs().max())
    up = F.interpolate(x.float().permute(0, 3, 1, 2), scale_factor=2, mode="bilinear")
    want = F.relu(F.conv2d(up, wt, None, 1, 1) * scale[:, :, None, None] + shift[:, :, None, None]).permute(0, 2, 3, 1)
    assert rel_rms(y, want) <= 6e-3


@pytest.mark.parametrize("tr", [2, 1])
@pytest.mark.parametrize("case", [(2, 32, 32, 256, 256), (1, 8, 64, 128, 512)], ids=lambda c: "x".join(map(str, c)))
def test_short_tile_rows_forms_reproduce_the_four_row_tiles_bitwise_plain(engine, case, tr, form_switch):
    """conv_halo_rows_tr_kernel<2, 2 / 1>: the plain 3x3 / 256-channel-block form on short tiles."""
    n, h, w, cin, cout = case
    g = torch.Generator().manual_seed(sum(case) + 23)
    x = torch.randn(n, h, w, cin, generator=g).to(torch.bfloat16)
    wt = torch.randn(cout, cin, 3, 3, generator=g) / (cin * 9) ** 0.5
    scale = 1.0 + 0.3 * torch.randn(n, cout, generator=g)
    shift = 0.2 * torch.randn(n, cout, generator=g)
    form_switch.setenv("DYF_HALO3_MIN_TILES", "1")
    form_switch.setenv("DYF_HALO_SPLITK", "0")
    form_switch.setenv("DYF_ROWS_TR", "4")
    ref, forms0 = _form_log(engine, lambda: engine.op_conv2d(x.cuda(), wt, 1, 1, scale.cuda(), shift.cuda(), act=2, path=1).float().cpu())
    assert "conv_halo_rows_kernel<2>" in forms0 and not any("+tr" in k for k in forms0), sorted(forms0)
    form_switch.setenv("DYF_ROWS_TR", str(tr))
    y, forms = _form_log(engine, lambda: engine.op_conv2d(x.cuda(), wt, 1, 1, scale.cuda(), shift.cuda(), act=2, path=1).float().cpu())
    assert f"conv_halo_rows_kernel<2>+tr{tr}" in forms, sorted(forms)
    assert torch.equal(y, ref), float((y - ref).abs().max())
    want = reference(x, wt, 1, 1, scale, shift, 2)
    assert rel_rms(y, want) <= 4e-3
