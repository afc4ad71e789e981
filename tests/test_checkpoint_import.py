"""Lightning-checkpoint import (dyffusion_amd/checkpoint.py) against the state-dict layout of the reference's Lightning
modules (tests/golden/ckpt_keys.json, generated from the imported reference by make_golden.py ckpt)."""
import torch

import dyffusion_amd as D
from dyffusion_amd.checkpoint import load_networks_from_checkpoints, split_lightning_state_dict
from tests.helpers import jload


def _fake(shapes, seed):
    g = torch.Generator().manual_seed(seed)
    return {k: (torch.randn(*s, generator=g) if len(s) else torch.tensor(7)) for k, s in shapes.items()}


def _mirrors(fx):
    mk = fx["model_kwargs"]
    f_in = fx["forecasting"]["model.model.init_conv.weight"][1]
    i_in = fx["interpolation"]["model.init_conv.weight"][1]
    cout = fx["forecasting"]["model.model.readout.0.weight"][1]
    kw = dict(dim=mk["dim"], with_time_emb=True, upsample_dims=mk["upsample_dims"], dropout=mk["dropout"],
              num_output_channels=cout, num_conditional_channels=0)
    return D.UNet(num_input_channels=f_in, **kw), D.UNet(num_input_channels=i_in, **kw)


def test_forecasting_checkpoint_fills_both_networks(tmp_path):
    fx = jload("ckpt_keys.json")
    sd = _fake(fx["forecasting"], 1)
    path = tmp_path / "last.ckpt"
    torch.save({"state_dict": sd, "epoch": 3, "global_step": 77}, path)
    F, I = _mirrors(fx)
    meta = load_networks_from_checkpoints(F, I, forecaster_ckpt=str(path))
    assert meta == {"epoch": 3, "global_step": 77}
    for k, v in F.state_dict().items():
        assert torch.equal(v, sd["model.model." + k]), k
    for k, v in I.state_dict().items():
        assert torch.equal(v, sd["model.interpolator.model." + k]), k
    # every tensor of the checkpoint was consumed
    used = {"model.model." + k for k in F.state_dict()} | {"model.interpolator.model." + k for k in I.state_dict()}
    assert used == set(sd)


def test_interpolator_taken_from_its_own_run_when_given():
    fx = jload("ckpt_keys.json")
    sd_f, sd_i = _fake(fx["forecasting"], 2), _fake(fx["interpolation"], 3)
    F, I = _mirrors(fx)
    load_networks_from_checkpoints(F, I, forecaster_ckpt={"state_dict": sd_f}, interpolator_ckpt={"state_dict": sd_i})
    for k, v in I.state_dict().items():
        assert torch.equal(v, sd_i["model." + k]), k
    for k, v in F.state_dict().items():
        assert torch.equal(v, sd_f["model.model." + k]), k


def test_split_roles():
    fx = jload("ckpt_keys.json")
    parts = split_lightning_state_dict({"state_dict": _fake(fx["forecasting"], 4)})
    assert set(parts) == {"forecaster", "interpolator"}
    parts = split_lightning_state_dict(_fake(fx["interpolation"], 5))
    assert set(parts) == {"interpolator"}
    plain = {k[len("model."):]: v for k, v in _fake(fx["interpolation"], 6).items()}
    assert set(split_lightning_state_dict(plain)) == {"model"}
