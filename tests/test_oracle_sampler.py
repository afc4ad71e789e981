"""Oracle sampling loop vs `predict()` outputs of the imported reference (fixture G4)."""
import json

import pytest
import torch

from oracle import nets, sampler
from tests.helpers import load_npz, max_abs, split_state

NAMES = ["sample_cold_refine", "sample_cold_norefine", "sample_naive", "sample_k2_data", "sample_k2_coldlast",
         "sample_k2_onlydyn", "sample_k2_plus2", "sample_ens3", "sample_dropout", "sample_datanoise", "sample_linear", "sample_fractional_refine",
         "sample_log_cold", "sample_log_naive"]


def run_oracle(z):
    hp = json.loads(str(z["hp"]))
    PF, PI = split_state(z, "F"), split_state(z, "I")
    N, B = hp["num_predictions"], hp["B"]
    x0 = torch.from_numpy(z["x0"]).repeat(N, 1, 1, 1)
    c = torch.from_numpy(z["c"]).repeat(N, 1, 1, 1)
    mcfg = hp["model"]
    drop = nets.DropoutSeeded(hp["dropout_seed"]) if hp.get("enable_interpolator_dropout") else nets.DropoutOff()

    def F_(x, t, cond):
        return nets.unet_simple_forward(PF, mcfg, x, t, cond)

    def I_(x, t, cond):
        return nets.unet_simple_forward(PI, mcfg, x, t, cond, dropout=drop)

    gen = torch.Generator().manual_seed(hp.get("noise_seed", 0))
    out = sampler.sample_loop(F_, I_, x0, c, hp, noise_fn=lambda t: torch.randn(t.shape, generator=gen))
    return sampler.reshape_ensemble(out, N), hp


@pytest.mark.parametrize("name", NAMES)
def test_sample_loop_matches_reference(name):
    z = load_npz(name + ".npz")
    got, hp = run_oracle(z)
    want = {k[len("out::"):]: v for k, v in z.items() if k.startswith("out::")}
    assert sorted(got) == sorted(want)
    for k in want:
        assert tuple(got[k].shape) == tuple(want[k].shape), k
        assert max_abs(got[k], want[k]) <= 5e-6 * max(1.0, float(abs(want[k]).max())), k


def test_eval_counts():
    base = dict(timesteps=16, schedule="before_t1_only", interpolate_before_t1=True, sampling_type="cold",
                refine_intermediate_predictions=True)
    assert sampler.count_net_evals(base) == {"forecaster": 16, "interpolator": 44}  # NS 60 forwards
    oisst = dict(timesteps=7, schedule="before_t1_only", additional_interpolation_steps=25, interpolate_before_t1=True,
                 sampling_type="cold", refine_intermediate_predictions=False)
    assert sampler.count_net_evals(oisst) == {"forecaster": 32, "interpolator": 61}


def test_fullsize_rollout_matches_reference_checksums():
    """Fixture G6: the BASELINE config-2 rollout (NS 221x42, h=16, cold, refine, dropout off), NB=1."""
    import numpy as np
    from oracle import init as oinit
    from tests.helpers import jload, load_npz, rel_rms

    meta = jload("fullsize_checksums.json")
    fields = load_npz("fullsize_ns_fields.npz")
    mk = meta["model"]
    PF = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 5, 3), meta["seeds"]["forecaster"])
    PI = oinit.seeded_state(oinit.unet_simple_param_shapes(64, 8, 3), meta["seeds"]["interpolator"])
    g = torch.Generator().manual_seed(meta["seeds"]["inputs"])
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    cfg = dict(timesteps=16, schedule="before_t1_only", interpolate_before_t1=True, sampling_type="cold",
               refine_intermediate_predictions=True, forward_conditioning="none", time_encoding="dynamics",
               num_input_channels=3)
    with torch.no_grad():
        out = sampler.sample_loop(lambda x, t, cond: nets.unet_simple_forward(PF, mk, x, t, cond),
                                  lambda x, t, cond: nets.unet_simple_forward(PI, mk, x, t, cond), x0, c, cfg)
    assert sorted(out) == sorted(meta["rollout"])
    for k, want in meta["rollout"].items():
        assert float(out[k].mean()) == pytest.approx(want["mean"], abs=5e-5), k
        assert np.allclose([float(out[k][tuple(p)]) for p in meta["probes"]], want["probes"], atol=2e-4), k
    for k in ("t1", "t8", "t16"):
        assert rel_rms(out[f"{k}_preds"], fields[k]) < 1e-4, k  # SURVEY 8c tolerance for an fp32 restatement
