"""Import the upstream reference (read-only at /root/reference) behind the offline stub shim.

TEST INFRASTRUCTURE ONLY.  Used by tests/golden/make_golden.py (run in the build container,
where /root/reference exists) to produce golden vectors.  Nothing on the product path and
nothing in `-m gpu` tests / bench.py / smoke() imports this module: /root/reference does not
exist on the GPU box.
"""
import os
import sys

REFERENCE_ROOT = os.environ.get("DYF_REFERENCE_ROOT", "/root/reference")
_STUBS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "_stubs")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_ROOT, "src", "diffusion"))


def activate():
    """Put the stubs, then the reference root, at the front of sys.path (idempotent)."""
    if not reference_available():
        raise RuntimeError(f"reference not found at {REFERENCE_ROOT}")
    os.environ.setdefault("TQDM_DISABLE", "1")
    for p in (REFERENCE_ROOT, _STUBS):
        if p in sys.path:
            sys.path.remove(p)
        sys.path.insert(0, p)


def build_reference_dyffusion(system="navier-stokes", model="unet_simple", model_kwargs=None, horizon=16,
                              diffusion_kwargs=None, num_predictions=1, box_size=60, seed=0,
                              randomize_norm_stats=True):
    """Construct the reference's MultiHorizonForecastingDYffusion + InterpolationExperiment pair
    (recipe: SURVEY.md Appendix D) with seeded random-init weights.  Returns (experiment, interpolator_exp)."""
    activate()
    import torch
    from omegaconf import DictConfig
    from src.experiment_types.forecasting_multi_horizon import MultiHorizonForecastingDYffusion
    from src.experiment_types.interpolation import InterpolationExperiment

    torch.manual_seed(seed)
    if system == "oisst":
        dm = DictConfig(_target_="src.datamodules.oisstv2.OISSTv2DataModule", box_size=box_size, horizon=horizon,
                        window=1, prediction_horizon=None)
    else:
        dm = DictConfig(_target_="src.datamodules.physical_systems_benchmark.PhysicalSystemsBenchmarkDataModule",
                        physical_system=system, horizon=horizon, window=1, prediction_horizon=None)
    targets = {"unet_simple": "src.models.unet_simple.UNet", "unet_resnet": "src.models.unet.Unet",
               "cnn_simple": "src.models.simple_conv_net.SimpleConvNet"}
    mk = dict(model_kwargs or {})
    mc = DictConfig(_target_=targets[model], loss_function="mse", name="", verbose=False, **mk)
    ipol = InterpolationExperiment(model_config=mc, datamodule_config=dm, enable_inference_dropout=True,
                                   num_predictions=1, verbose=False)
    dk = dict(timesteps=horizon, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
              additional_interpolation_steps=0, sampling_type="cold", time_encoding="dynamics",
              enable_interpolator_dropout=True, refine_intermediate_predictions=True, loss_function="l1",
              verbose=False)
    dk.update(diffusion_kwargs or {})
    dc = DictConfig(_target_="src.diffusion.dyffusion.DYffusion", interpolator=ipol, **dk)
    exp = MultiHorizonForecastingDYffusion(model_config=mc, datamodule_config=dm, diffusion_config=dc,
                                           num_predictions=num_predictions, verbose=False).eval()
    if randomize_norm_stats:
        g = torch.Generator().manual_seed(seed + 1000)
        for net in (exp.model.model, ipol.model):
            for m in net.modules():
                if isinstance(m, torch.nn.BatchNorm2d):
                    m.running_mean.copy_(0.1 * torch.randn(m.running_mean.shape, generator=g))
                    m.running_var.copy_(0.5 + torch.rand(m.running_var.shape, generator=g))
                    m.bias.data.copy_(0.05 * torch.randn(m.bias.shape, generator=g))
    return exp, ipol
