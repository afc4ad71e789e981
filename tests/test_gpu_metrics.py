"""On-device ensemble metrics (dyf_ensemble_metrics) against oracle/metrics.py."""
import numpy as np
import pytest
import torch

from oracle import metrics as om

pytestmark = pytest.mark.gpu


def _engine():
    import dyffusion_amd as D
    from dyffusion_amd.engine import HipEngine

    net = D.UNet(dim=8, with_time_emb=True, upsample_dims=[64, 64], num_input_channels=1, num_output_channels=1,
                 num_conditional_channels=0, spatial_shape=(16, 16))
    return HipEngine(net.engine_net_config(), net.engine_net_config(), height=16, width=16, max_batch=1, use_graph=False)


@pytest.mark.parametrize("shape", [(20, 4, 3, 221, 42), (1, 2, 1, 9, 7), (64, 1, 2, 33, 5), (5, 3, 2, 60, 60),
                                   # > 64 members: the tiled kernel (fewer points per workgroup, several threads per point)
                                   (65, 1, 2, 33, 5), (256, 2, 3, 23, 11), (1000, 1, 1, 17, 9), (4097, 1, 1, 5, 3)])
def test_ensemble_metrics_match_oracle(shape):
    from dyffusion_amd.metrics import evaluate_ensemble_prediction

    eng = _engine()
    g = torch.Generator().manual_seed(sum(shape))
    truth = torch.randn(*shape[1:], generator=g)
    preds = truth[None] * 0.7 + torch.randn(*shape, generator=g) * 0.6 + 0.1
    got = evaluate_ensemble_prediction(preds.cuda(), truth.cuda(), eng)
    ref = om.evaluate_ensemble_prediction(preds.numpy(), truth.numpy())
    for k in ("mse", "ssr", "crps"):
        if shape[0] == 1 and k == "ssr":
            assert got[k] == 0.0 and ref[k] == 0.0  # one member: zero spread
            continue
        assert got[k] == pytest.approx(ref[k], rel=2e-5), k


def test_eval_ensemble_predictions_keys_and_average():
    from dyffusion_amd.metrics import eval_ensemble_predictions

    eng = _engine()
    g = torch.Generator().manual_seed(3)
    res = {}
    for k in (1, 2, 3):
        t = torch.randn(2, 3, 11, 13, generator=g)
        res[f"t{k}_targets"] = t.cuda()
        res[f"t{k}_preds"] = (t[None] + 0.3 * k * torch.randn(6, 2, 3, 11, 13, generator=g)).cuda()
    out = eval_ensemble_predictions(res, eng, split="val", infix="6ens_mems/")
    for k in (1, 2, 3):
        ref = om.evaluate_ensemble_prediction(res[f"t{k}_preds"].cpu().numpy(), res[f"t{k}_targets"].cpu().numpy())
        for m in ("mse", "ssr", "crps"):
            assert out[f"val/6ens_mems/t{k}/{m}"] == pytest.approx(ref[m], rel=2e-5)
    for m in ("mse", "ssr", "crps"):
        assert out[f"val/6ens_mems/avg/{m}"] == pytest.approx(np.mean([out[f"val/6ens_mems/t{k}/{m}"] for k in (1, 2, 3)]))


def test_argument_checks():
    eng = _engine()
    with pytest.raises(ValueError):
        eng.ensemble_metrics(torch.zeros(3, 2, 5).cuda(), torch.zeros(2, 4).cuda())
    with pytest.raises(ValueError):
        eng.ensemble_metrics(torch.zeros(15361, 2, 5).cuda(), torch.zeros(2, 5).cuda())  # beyond the 60 KB staging tile
