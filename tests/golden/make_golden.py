"""Generate the golden fixtures in this directory from the IMPORTED reference (/root/reference).

Run once in the build container:  python tests/golden/make_golden.py
The reference cannot travel to the GPU box; these small .npz/.json files (inputs + expected outputs of the
reference's own code) are what pins the oracle (oracle/*.py), which in turn is the checker for the HIP path.
Fixture kinds: G1 schedules.json, G3 net_*.npz, G4 sample_*.npz, G6 fullsize_checksums.json; metrics_*.npz
(ensemble mse / spread-skill of src/utilities/evaluation.py).
"""
import contextlib
import json
import os
import sys
import zlib

import numpy as np
import torch

_DIR = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.abspath(os.path.join(_DIR, "..", "..")))
# DYF_GOLDEN_OUT=<dir>: write the fixtures somewhere else (tests/test_golden_regen.py regenerates into a scratch directory and
# compares with the committed files bit for bit)
HERE = os.environ.get("DYF_GOLDEN_OUT") or _DIR

from oracle import init as oinit  # noqa: E402
from oracle import ref_import  # noqa: E402
from oracle.nets import DropoutSeeded  # noqa: E402

ref_import.activate()
torch.set_num_threads(max(1, os.cpu_count() or 1))


@contextlib.contextmanager
def patched_dropout(source):
    """Route every nn.Dropout of the reference through `source` (so masks are seeded / recordable)."""
    orig = torch.nn.Dropout.forward

    def fwd(self, x):
        if not self.training or source is None:
            return x
        return source.apply(x, self.p)

    torch.nn.Dropout.forward = fwd
    try:
        yield
    finally:
        torch.nn.Dropout.forward = orig


@contextlib.contextmanager
def patched_randn_like(seed):
    gen = torch.Generator().manual_seed(seed)
    orig = torch.randn_like
    draws = []

    def fake(t, **kw):
        z = torch.randn(t.shape, generator=gen)
        draws.append(z)
        return z

    torch.randn_like = fake
    try:
        yield draws
    finally:
        torch.randn_like = orig


def np_state(sd):
    return {f"P::{k}": v.detach().cpu().numpy() for k, v in sd.items()}


def load_seeded(net, seed):
    shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
    st = oinit.seeded_state(shapes, seed)
    net.load_state_dict(st, strict=True)
    return shapes


# ------------------------------------------------------------------------------------------------ G1
def gen_schedules():
    from src.diffusion.dyffusion import BaseDYffusion

    cases = []
    grid = [
        dict(h=4, schedule="before_t1_only", k=0, fac=0, before_t1=True),
        dict(h=5, schedule="before_t1_only", k=1, fac=0, before_t1=True),
        dict(h=5, schedule="before_t1_only", k=2, fac=0, before_t1=True),
        dict(h=7, schedule="before_t1_only", k=25, fac=0, before_t1=True),
        dict(h=16, schedule="before_t1_only", k=0, fac=0, before_t1=True),
        dict(h=64, schedule="before_t1_only", k=0, fac=0, before_t1=True),
        dict(h=5, schedule="linear", k=0, fac=1, before_t1=False),
        dict(h=5, schedule="linear", k=0, fac=1, before_t1=True),
        dict(h=5, schedule="linear", k=0, fac=2, before_t1=False),
        dict(h=6, schedule="linear", k=0, fac=2, before_t1=True),
    ]
    names = [None, "only_dynamics", "only_dynamics_plus2", "only_dynamics_plus_discrete2", "every2nd", "every5th",
             "first3", "first0.5", "every1", "first1"]

    class _Bare(BaseDYffusion):  # the schedule logic lives in BaseDYffusion; no interpolator needed
        def _interpolate(self, *a, **k):
            raise NotImplementedError

        def p_losses(self, *a, **k):
            raise NotImplementedError

    for g in grid:
        exp, _ = ref_import.build_reference_dyffusion(
            system="spring-mesh", model="cnn_simple",
            model_kwargs=dict(dim=4, with_time_emb=True, kernel_sizes=[3], dropout=0.0), horizon=g["h"],
            diffusion_kwargs=dict(schedule=g["schedule"], additional_interpolation_steps=g["k"],
                                  additional_interpolation_steps_factor=g["fac"],
                                  interpolate_before_t1=g["before_t1"]))
        dy = exp.model
        case = dict(g)
        case["num_timesteps"] = dy.num_timesteps
        case["d_to_i"] = {str(d): float(dy.diffusion_step_to_interpolation_step(d)) for d in range(1, dy.num_timesteps)}
        case["dynamical_steps"] = {str(d): float(i) for d, i in dy.dynamical_steps.items()}
        case["artificial_steps"] = {str(d): float(i) for d, i in dy.artificial_interpolation_steps.items()}
        case["schedules"] = {}
        for nm in names:
            try:
                dy.sampling_schedule = nm if nm is not None else dy.full_sampling_schedule
                val = [float(s) for s in dy.sampling_schedule]
                ints = all(isinstance(s, int) for s in dy.sampling_schedule)
                case["schedules"][str(nm)] = dict(ok=True, steps=val, all_int=ints)
            except Exception as e:  # invalid for this T: record that the reference refuses it
                case["schedules"][str(nm)] = dict(ok=False, error=type(e).__name__)
        cases.append(case)
    with open(os.path.join(HERE, "schedules.json"), "w") as f:
        json.dump(cases, f, indent=1)
    print("schedules.json:", len(cases), "cases")


# ------------------------------------------------------------------------------------------------ G3
def gen_nets():
    from omegaconf import DictConfig
    from src.models.simple_conv_net import SimpleConvNet
    from src.models.unet_simple import UNet

    torch.manual_seed(0)
    specs = [
        ("net_unet_simple_a", dict(dim=8, upsample_dims=[64, 64], n_in=6, n_cond=2, n_out=3, hw=(23, 11), nb=2,
                                   dropout=0.15)),
        ("net_unet_simple_b", dict(dim=4, upsample_dims=[64, 64], n_in=4, n_cond=1, n_out=4, hw=(10, 10), nb=3,
                                   dropout=0.3)),
        ("net_unet_simple_c", dict(dim=8, upsample_dims=[128, 64], n_in=3, n_cond=0, n_out=2, hw=(40, 17), nb=1,
                                   dropout=0.1)),
        # outer_sample_mode="nearest" (unet_simple.py:172-179): both outer resamples pick the floor(dst * in/out) source
        ("net_unet_simple_d", dict(dim=8, upsample_dims=[64, 64], n_in=5, n_cond=1, n_out=3, hw=(23, 11), nb=2,
                                   dropout=0.2, mode="nearest")),
        # input_dropout > 0 (unet_simple.py:116, 168): a Dropout on init_conv's output, the FIRST dropout site of a forward
        ("net_unet_simple_e", dict(dim=8, upsample_dims=[64, 64], n_in=4, n_cond=1, n_out=3, hw=(23, 11), nb=2,
                                   dropout=0.15, input_dropout=0.1)),
    ]
    only = os.environ.get("DYF_GOLDEN_ONLY")
    for name, sp in specs:
        if only and name != only:
            continue
        mode = sp.get("mode", "bilinear")
        net = UNet(dim=sp["dim"], with_time_emb=True, outer_sample_mode=mode, upsample_dims=sp["upsample_dims"],
                   dropout=sp["dropout"], input_dropout=sp.get("input_dropout", 0.0), num_input_channels=sp["n_in"],
                   num_output_channels=sp["n_out"], num_conditional_channels=sp["n_cond"], spatial_shape=sp["hw"],
                   loss_function="mse", verbose=False).eval()
        shapes = load_seeded(net, seed=11)
        ref_shapes = oinit.unet_simple_param_shapes(sp["dim"], sp["n_in"] + sp["n_cond"], sp["n_out"])
        assert {k: tuple(v) for k, v in ref_shapes.items()} == shapes, "oracle shape table != reference state_dict"
        g = torch.Generator().manual_seed(5)
        x = torch.randn(sp["nb"], sp["n_in"], *sp["hw"], generator=g)
        c = torch.rand(sp["nb"], sp["n_cond"], *sp["hw"], generator=g) if sp["n_cond"] else None
        t = torch.tensor([1.0, 2.5, 0.3333][: sp["nb"]])
        with torch.no_grad():
            y_eval = net(x, time=t, condition=c)
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.train()
            with patched_dropout(DropoutSeeded(seed=77)):
                y_drop = net(x, time=t, condition=c)
        arrs = dict(np_state(net.state_dict()), x=x.numpy(), t=t.numpy(), y_eval=y_eval.numpy(),
                    y_drop=y_drop.numpy(), dropout_seed=np.int64(77),
                    cfg=json.dumps(dict(dim=sp["dim"], upsample_dims=sp["upsample_dims"], outer_sample_mode=mode,
                                        with_time_emb=True, dropout=sp["dropout"], input_dropout=sp.get("input_dropout", 0.0))))
        if c is not None:
            arrs["c"] = c.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        print(name, "y_eval std", float(y_eval.std()), "y_drop std", float(y_drop.std()))

    if only:
        return
    # SimpleConvNet (spring-mesh plumbing config)
    net = SimpleConvNet(dim=8, with_time_emb=True, kernel_sizes=[9, 7, 5, 3], dropout=0.1, num_input_channels=8,
                        num_output_channels=4, num_conditional_channels=1, spatial_shape=(10, 10),
                        loss_function="mse", verbose=False).eval()
    shapes = load_seeded(net, seed=12)
    assert {k: tuple(v) for k, v in oinit.simple_conv_net_param_shapes(8, 9, 4, [9, 7, 5, 3]).items()} == shapes
    g = torch.Generator().manual_seed(6)
    x = torch.randn(2, 8, 10, 10, generator=g)
    c = torch.rand(2, 1, 10, 10, generator=g)
    t = torch.tensor([1.0, 3.0])
    with torch.no_grad():
        y_eval = net(x, time=t, condition=c)
    np.savez_compressed(os.path.join(HERE, "net_simple_conv.npz"), **np_state(net.state_dict()), x=x.numpy(),
                        c=c.numpy(), t=t.numpy(), y_eval=y_eval.numpy(),
                        cfg=json.dumps(dict(dim=8, kernel_sizes=[9, 7, 5, 3], with_time_emb=True, dropout=0.1,
                                            residual=True)))
    print("net_simple_conv y std", float(y_eval.std()))


def gen_resnet_unets():
    """G3 for src.models.unet.Unet (OISST / synthetic backbone): tiny nets, eval + seeded dropout."""
    from src.models.unet import Unet

    specs = [
        ("net_unet_resnet_a", dict(dim=8, mults=(1, 2, 4), n_in=2, n_cond=1, n_out=1, hw=(12, 12), nb=2,
                                   bd=0.3, bd1=0.1, ad=0.2)),
        ("net_unet_resnet_b", dict(dim=16, mults=(1, 2), n_in=1, n_cond=0, n_out=1, hw=(20, 12), nb=2,
                                   bd=0.6, bd1=0.2, ad=0.6)),
        # input_dropout > 0 (unet.py:162-163, 276-277): dropout_input_for_residual and dropout_input on init_conv's output
        ("net_unet_resnet_c", dict(dim=8, mults=(1, 2), n_in=2, n_cond=1, n_out=1, hw=(12, 16), nb=2,
                                   bd=0.3, bd1=0.1, ad=0.2, ind=0.15)),
        # options no shipped config sets (unet.py:127-135): keep_spatial_dims, double_conv_layer=False, learned_sinusoidal_cond
        ("net_unet_resnet_d", dict(dim=8, mults=(1, 2, 2), n_in=2, n_cond=1, n_out=1, hw=(10, 14), nb=2, bd=0.3, bd1=0.1, ad=0.2,
                                   extra=dict(keep_spatial_dims=True))),
        ("net_unet_resnet_e", dict(dim=8, mults=(1, 2), n_in=2, n_cond=0, n_out=2, hw=(12, 12), nb=2, bd=0.3, bd1=0.2, ad=0.1,
                                   extra=dict(double_conv_layer=False))),
        # (the outer resampler of unet.Unet cannot be pinned: the reference constructor raises AttributeError -- unet.py:155 reads
        # self.outer_sample_mode, which is never set -- so upsample_dims / outer_sample_mode stay unsupported here as well)
        ("net_unet_resnet_g", dict(dim=8, mults=(1, 2), n_in=2, n_cond=1, n_out=1, hw=(12, 16), nb=2, bd=0.2, bd1=0.0, ad=0.0,
                                   extra=dict(learned_sinusoidal_cond=True, learned_sinusoidal_dim=16))),
    ]
    only = os.environ.get("DYF_GOLDEN_ONLY")
    for name, sp in specs:
        if only and name != only:
            continue
        net = Unet(dim=sp["dim"], dim_mults=sp["mults"], with_time_emb=True, block_dropout=sp["bd"],
                   block_dropout1=sp["bd1"], attn_dropout=sp["ad"], input_dropout=sp.get("ind", 0.0), num_input_channels=sp["n_in"],
                   num_output_channels=sp["n_out"], num_conditional_channels=sp["n_cond"], spatial_shape=sp["hw"],
                   loss_function="mse", verbose=False, **sp.get("extra", {})).eval()
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        st = oinit.seeded_state(shapes, seed=13)
        for k in st:  # LayerNorm gains `g` are 4-D (1,C,1,1): treat as gains, not as conv weights
            if k.endswith(".weights"):  # LearnedSinusoidalPosEmb frequencies: randn like the reference's init
                st[k] = torch.randn(shapes[k], generator=torch.Generator().manual_seed(5))
            if k.endswith(".norm.g"):
                st[k] = 1.0 + 0.1 * torch.randn(shapes[k], generator=torch.Generator().manual_seed(zlib.crc32(k.encode()) % 997))  # (not hash(k): str hashes are salted per process)
        net.load_state_dict(st, strict=True)
        g = torch.Generator().manual_seed(7)
        x = torch.randn(sp["nb"], sp["n_in"], *sp["hw"], generator=g)
        c = torch.rand(sp["nb"], sp["n_cond"], *sp["hw"], generator=g) if sp["n_cond"] else None
        t = torch.tensor([1.0, 2.5])
        with torch.no_grad():
            y_eval = net(x, time=t, condition=c)
            for m in net.modules():
                if isinstance(m, torch.nn.Dropout):
                    m.train()
            with patched_dropout(DropoutSeeded(seed=78)):
                y_drop = net(x, time=t, condition=c)
        arrs = dict(np_state(net.state_dict()), x=x.numpy(), t=t.numpy(), y_eval=y_eval.numpy(), y_drop=y_drop.numpy(),
                    dropout_seed=np.int64(78),
                    cfg=json.dumps(dict(dim=sp["dim"], dim_mults=list(sp["mults"]), with_time_emb=True,
                                        block_dropout=sp["bd"], block_dropout1=sp["bd1"], attn_dropout=sp["ad"],
                                        resnet_block_groups=8, input_dropout=sp.get("ind", 0.0),
                                        **{"upsample_dims": None, **sp.get("extra", {})})))
        if c is not None:
            arrs["c"] = c.numpy()
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **arrs)
        print(name, "y_eval std", float(y_eval.std()), "y_drop std", float(y_drop.std()), "params",
              sum(v.numel() for v in net.state_dict().values()))


# ------------------------------------------------------------------------------------------------ G4
def gen_samples():
    base_model = dict(dim=4, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True,
                      input_dropout=0.0, dropout=0.2)
    variants = [
        ("sample_cold_refine", dict(h=4), dict()),
        ("sample_cold_norefine", dict(h=4), dict(refine_intermediate_predictions=False)),
        ("sample_naive", dict(h=4), dict(sampling_type="naive", refine_intermediate_predictions=False)),
        ("sample_k2_data", dict(h=5), dict(additional_interpolation_steps=2, forward_conditioning="data",
                                            refine_intermediate_predictions=False)),
        ("sample_k2_coldlast", dict(h=5), dict(additional_interpolation_steps=2, use_cold_sampling_for_last_step=True,
                                                refine_intermediate_predictions=True)),
        ("sample_k2_onlydyn", dict(h=5), dict(additional_interpolation_steps=2, sampling_schedule="only_dynamics",
                                               refine_intermediate_predictions=False)),
        ("sample_k2_plus2", dict(h=5), dict(additional_interpolation_steps=2, sampling_schedule="only_dynamics_plus2",
                                             refine_intermediate_predictions=False, time_encoding="discrete")),
        ("sample_ens3", dict(h=4, N=3), dict()),
        ("sample_dropout", dict(h=4, N=2, dropout_seed=31), dict(enable_interpolator_dropout=True)),
        ("sample_datanoise", dict(h=4, noise_seed=41), dict(forward_conditioning="data+noise",
                                                            additional_interpolation_steps=1,
                                                            refine_intermediate_predictions=False,
                                                            time_encoding="normalized")),
        ("sample_linear", dict(h=5), dict(schedule="linear", additional_interpolation_steps_factor=1,
                                          interpolate_before_t1=False, refine_intermediate_predictions=False)),
        # refinement pass at fractional prediction times (dyffusion.py:412-421): extra keys "t0.5_preds", "t1.5_preds", ...
        ("sample_fractional_refine", dict(h=4), dict(prediction_timesteps=[0.5, 1, 1.5, 2, 3, 3.5])),
        # log_every_t: the per-step intermediates of sample_loop (dyffusion.py:396-406): t{k}_preds2, intermediate_{s}_x0hat,
        # xipol_{s}_dmodel, xipol_{s}_dmodel2
        ("sample_log_cold", dict(h=4), dict(log_every_t=1, additional_interpolation_steps=1)),
        ("sample_log_naive", dict(h=4), dict(log_every_t=1, sampling_type="naive", refine_intermediate_predictions=False)),
    ]
    only = os.environ.get("DYF_GOLDEN_ONLY")
    for name, meta, dk in variants:
        if only and name not in only.split(","):
            continue
        h, N = meta["h"], meta.get("N", 1)
        dkw = dict(enable_interpolator_dropout=False)
        dkw.update(dk)
        exp, ipol = ref_import.build_reference_dyffusion(system="spring-mesh", model="unet_simple",
                                                         model_kwargs=base_model, horizon=h, diffusion_kwargs=dkw,
                                                         num_predictions=N)
        load_seeded(exp.model.model, seed=21)
        load_seeded(ipol.model, seed=22)
        g = torch.Generator().manual_seed(9)
        B = 2
        x0 = torch.randn(B, 4, 10, 10, generator=g)
        c = torch.rand(B, 1, 10, 10, generator=g)
        x0_t = torch.stack([x0] * N, 0).reshape(N * B, 4, 10, 10)  # "N B ... -> (N B) ..." ensemble-major
        c_t = torch.stack([c] * N, 0).reshape(N * B, 1, 10, 10)
        src = DropoutSeeded(seed=meta["dropout_seed"]) if "dropout_seed" in meta else None
        with torch.no_grad(), patched_dropout(src), patched_randn_like(meta.get("noise_seed", 0)):
            out = exp.predict(x0_t, condition=c_t)
        arrs = {f"out::{k}": v.numpy() for k, v in out.items()}
        arrs.update({f"F::{k}": v.numpy() for k, v in exp.model.model.state_dict().items()})
        arrs.update({f"I::{k}": v.numpy() for k, v in ipol.model.state_dict().items()})
        dyn = exp.model
        hp = dict(timesteps=h, num_input_channels=4, num_predictions=N, B=B, model=base_model,
                  sampling_schedule_resolved=[float(s) for s in dyn.sampling_schedule], **{
                      k: dkw.get(k, d) for k, d in dict(
                          schedule="before_t1_only", additional_interpolation_steps=0,
                          additional_interpolation_steps_factor=0, interpolate_before_t1=True, sampling_type="cold",
                          sampling_schedule=None, time_encoding="dynamics", refine_intermediate_predictions=True,
                          use_cold_sampling_for_last_step=False, forward_conditioning="none",
                          enable_interpolator_dropout=False, prediction_timesteps=None, log_every_t=None).items()})
        hp.update({k: v for k, v in meta.items() if k.endswith("_seed")})
        np.savez_compressed(os.path.join(HERE, name + ".npz"), x0=x0.numpy(), c=c.numpy(), hp=json.dumps(hp), **arrs)
        print(name, {k: tuple(v.shape) for k, v in out.items()}, "std t_last", float(out[f"t{h}_preds"].std()))


# ------------------------------------------------------------------------------------------------ forecaster objective
def gen_plosses():
    """`DYffusion.p_losses` of the imported reference in eval mode (the validation objective; dyffusion.py:496-567): per-row
    diffusion steps incl. t = 0 and t = T-1, both loss terms."""
    base_model = dict(dim=4, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True,
                      input_dropout=0.0, dropout=0.2)
    variants = [
        ("plosses_a", dict(h=4), dict(lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")),
        ("plosses_b", dict(h=5), dict(additional_interpolation_steps=2, forward_conditioning="data", lambda_reconstruction=0.7,
                                       lambda_reconstruction2=1.0, loss_function="mse", time_encoding="normalized")),
        ("plosses_c", dict(h=4), dict(lambda_reconstruction2=0.0, loss_function="l1", time_encoding="discrete")),
    ]
    for name, meta, dk in variants:
        h = meta["h"]
        dkw = dict(enable_interpolator_dropout=False)
        dkw.update(dk)
        exp, ipol = ref_import.build_reference_dyffusion(system="spring-mesh", model="unet_simple",
                                                         model_kwargs=base_model, horizon=h, diffusion_kwargs=dkw)
        load_seeded(exp.model.model, seed=31)
        load_seeded(ipol.model, seed=32)
        dyn = exp.model.eval()
        T = dyn.num_timesteps
        g = torch.Generator().manual_seed(19)
        B = 6
        xt_last = torch.randn(B, 4, 10, 10, generator=g)
        cond = torch.randn(B, 4, 10, 10, generator=g)
        sc = torch.rand(B, 1, 10, 10, generator=g)
        t = torch.tensor([0, 1, T - 1, 2 % T, T - 2, 0])
        with torch.no_grad():
            out = dyn.p_losses(xt_last, cond, t, static_condition=sc)
        hp = dict(timesteps=h, num_timesteps=T, model=base_model, B=B,
                  **{k: dkw.get(k, d) for k, d in dict(
                      schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
                      interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none",
                      lambda_reconstruction=1.0, lambda_reconstruction2=0.0, loss_function="l1",
                      enable_interpolator_dropout=False).items()})
        arrs = {f"F::{k}": v.numpy() for k, v in dyn.model.state_dict().items()}
        arrs.update({f"I::{k}": v.numpy() for k, v in ipol.model.state_dict().items()})
        vals = {k.split("/")[-1]: float(v) for k, v in out.items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), xt_last=xt_last.numpy(), cond=cond.numpy(), sc=sc.numpy(),
                            t=t.numpy(), hp=json.dumps(hp), losses=json.dumps(vals), **arrs)
        print(name, vals)


def gen_plosses_train():
    """The TRAINING step of the imported reference: `DYffusion.p_losses` with the module in train mode (forecaster:
    batch-statistics BatchNorm, Dropout active; frozen interpolator in eval mode with its Dropout active), loss.backward().
    All nn.Dropout layers draw from DropoutSeeded(seed) in call order.  Stored: inputs, weights, the loss dict, the gradient
    of `loss` w.r.t. every forecaster parameter and the BatchNorm running statistics after the step."""
    base_model = dict(dim=4, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True,
                      input_dropout=0.0, dropout=0.2)
    variants = [
        ("plosses_train_a", dict(h=4, seed=71), dict(lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")),
        ("plosses_train_b", dict(h=5, seed=72), dict(additional_interpolation_steps=2, forward_conditioning="data",
                                                      lambda_reconstruction=0.7, lambda_reconstruction2=1.0, loss_function="mse",
                                                      time_encoding="normalized")),
    ]
    for name, meta, dk in variants:
        h = meta["h"]
        dkw = dict(enable_interpolator_dropout=True)
        dkw.update(dk)
        exp, ipol = ref_import.build_reference_dyffusion(system="spring-mesh", model="unet_simple",
                                                         model_kwargs=base_model, horizon=h, diffusion_kwargs=dkw)
        load_seeded(exp.model.model, seed=31)
        load_seeded(ipol.model, seed=32)
        dyn = exp.model
        dyn.train()
        # nn.Module.train() recurses into the frozen interpolator as well; `freeze_model` (dyffusion.py:468, utils.py:553-557)
        # put it in eval mode and Lightning (>= 2.2) restores every submodule's own mode when fitting: keep it frozen in eval
        dyn.interpolator.eval()
        for p in dyn.model.parameters():
            p.requires_grad_(True)
        assert not ipol.model.training and all(not p.requires_grad for p in ipol.model.parameters())
        T = dyn.num_timesteps
        g = torch.Generator().manual_seed(19)
        B = 6
        xt_last = torch.randn(B, 4, 10, 10, generator=g)
        cond = torch.randn(B, 4, 10, 10, generator=g)
        sc = torch.rand(B, 1, 10, 10, generator=g)
        t = torch.tensor([0, 1, T - 1, 2 % T, T - 2, 0])
        sd0 = {k: v.detach().clone() for k, v in dyn.model.state_dict().items()}
        with patched_dropout(DropoutSeeded(seed=meta["seed"])):
            out = dyn.p_losses(xt_last, cond, t, static_condition=sc)
            out["loss"].backward()
        hp = dict(timesteps=h, num_timesteps=T, model=base_model, B=B, dropout_seed=meta["seed"],
                  **{k: dkw.get(k, d) for k, d in dict(
                      schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
                      interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none",
                      lambda_reconstruction=1.0, lambda_reconstruction2=0.0, loss_function="l1",
                      enable_interpolator_dropout=True).items()})
        arrs = {f"F::{k}": v.numpy() for k, v in sd0.items()}
        arrs.update({f"I::{k}": v.numpy() for k, v in ipol.model.state_dict().items()})
        arrs.update({f"G::{k}": p.grad.numpy() for k, p in dyn.model.named_parameters()})
        arrs.update({f"B::{k}": v.detach().numpy() for k, v in dyn.model.state_dict().items()
                     if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")})
        vals = {k.split("/")[-1]: float(v) for k, v in out.items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), xt_last=xt_last.numpy(), cond=cond.numpy(), sc=sc.numpy(),
                            t=t.numpy(), hp=json.dumps(hp), losses=json.dumps(vals), **arrs)
        gn = float(torch.cat([p.grad.reshape(-1) for p in dyn.model.parameters()]).norm())
        print(name, vals, "grad norm", gn)


def gen_plosses_train_resnet():
    """The TRAINING step of the imported reference with the ResNet-UNet `src.models.unet.Unet` as the forecaster (OISST channel
    plumbing: C=1, forecaster conditioned on the initial condition, interpolator 2 -> 1): `DYffusion.p_losses` in train mode
    (GroupNorm has no batch statistics; every nn.Dropout -- block dropouts, LinearAttention input dropout, Attention
    probability dropout -- draws from DropoutSeeded(seed) in call order; the frozen interpolator in eval mode with its dropout
    active), loss.backward().  Stored: inputs, weights, losses, the gradient w.r.t. every forecaster parameter, and for variant
    b the normal draws of forward_conditioning="data+noise"."""
    variants = [
        ("plosses_train_resnet_a", dict(h=4, seed=81, dim=8, mults=(1, 2), hw=(12, 8)),
         dict(forward_conditioning="data", lambda_reconstruction=1.0, lambda_reconstruction2=0.5, loss_function="l1")),
        ("plosses_train_resnet_b", dict(h=5, seed=82, dim=8, mults=(1, 2, 4), hw=(12, 12)),
         dict(additional_interpolation_steps=2, forward_conditioning="data+noise", lambda_reconstruction=0.7,
              lambda_reconstruction2=1.0, loss_function="mse")),
    ]
    for name, meta, dk in variants:
        h = meta["h"]
        mk = dict(dim=meta["dim"], dim_mults=meta["mults"], with_time_emb=True, block_dropout=0.2, block_dropout1=0.1, attn_dropout=0.15)
        dkw = dict(enable_interpolator_dropout=True)
        dkw.update(dk)
        exp, ipol = ref_import.build_reference_dyffusion(system="oisst", model="unet_resnet", model_kwargs=mk, horizon=h,
                                                         diffusion_kwargs=dkw, box_size=meta["hw"][0])

        def seed_net(net, seed):
            shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
            st = oinit.seeded_state(shapes, seed, gain=1.0)
            for k in st:
                if k.endswith(".norm.g"):
                    st[k] = 1.0 + 0.1 * torch.randn(shapes[k], generator=torch.Generator().manual_seed(len(k)))
            net.load_state_dict(st, strict=True)

        seed_net(exp.model.model, 41)
        seed_net(ipol.model, 42)
        dyn = exp.model
        dyn.train()
        dyn.interpolator.eval()  # frozen (see gen_plosses_train)
        for p in dyn.model.parameters():
            p.requires_grad_(True)
        assert not ipol.model.training and all(not p.requires_grad for p in ipol.model.parameters())
        T = dyn.num_timesteps
        fch, ich = dyn.model, ipol.model
        g = torch.Generator().manual_seed(23)
        B, (Hh, Ww) = 5, meta["hw"]
        xt_last = torch.randn(B, 1, Hh, Ww, generator=g)
        cond = torch.randn(B, 1, Hh, Ww, generator=g)
        t = torch.tensor([0, 1, T - 1, 2 % T, T - 2])
        sd0 = {k: v.detach().clone() for k, v in dyn.model.state_dict().items()}
        with patched_dropout(DropoutSeeded(seed=meta["seed"])), patched_randn_like(500 + meta["seed"]) as draws:
            out = dyn.p_losses(xt_last, cond, t, static_condition=None)
            out["loss"].backward()
        hp = dict(timesteps=h, num_timesteps=T, model=dict(mk, dim_mults=list(mk["dim_mults"])), B=B, dropout_seed=meta["seed"],
                  noise_seed=500 + meta["seed"], n_noise_draws=len(draws),
                  forecaster_channels=dict(inputs=fch.num_input_channels, cond=fch.num_conditional_channels),
                  interpolator_channels=dict(inputs=ich.num_input_channels, cond=ich.num_conditional_channels),
                  **{k: dkw.get(k, d) for k, d in dict(
                      schedule="before_t1_only", additional_interpolation_steps=0, additional_interpolation_steps_factor=0,
                      interpolate_before_t1=True, time_encoding="dynamics", forward_conditioning="none",
                      lambda_reconstruction=1.0, lambda_reconstruction2=0.0, loss_function="l1",
                      enable_interpolator_dropout=True).items()})
        arrs = {f"F::{k}": v.numpy() for k, v in sd0.items()}
        arrs.update({f"I::{k}": v.numpy() for k, v in ipol.model.state_dict().items()})
        arrs.update({f"G::{k}": p.grad.numpy() for k, p in dyn.model.named_parameters()})
        vals = {k.split("/")[-1]: float(v) for k, v in out.items()}
        np.savez_compressed(os.path.join(HERE, name + ".npz"), xt_last=xt_last.numpy(), cond=cond.numpy(), t=t.numpy(),
                            hp=json.dumps(hp), losses=json.dumps(vals), **arrs)
        gn = float(torch.cat([p.grad.reshape(-1) for p in dyn.model.parameters()]).norm())
        print(name, vals, "grad norm", gn, "noise draws", len(draws), "channels", hp["forecaster_channels"], hp["interpolator_channels"])


def gen_interp_train():
    """Stage 1 of the reference's training: `InterpolationExperiment.get_loss(batch)` (interpolation.py:149-167 ->
    `BaseModel.get_loss`, _base_model.py:108-138) with the interpolator network in train mode (batch-statistics BatchNorm,
    Dropout active) and loss.backward().  The random interpolation times come from a patched torch.randint (fixed indices,
    stored); all nn.Dropout layers draw from DropoutSeeded(seed) in call order.  Stored: batch, weights, chosen times, loss,
    the gradient w.r.t. every parameter and the BatchNorm running statistics after the step."""
    base_model = dict(dim=4, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True,
                      input_dropout=0.0, dropout=0.2)
    for name, h, loss_fn, seed, idx in (("interp_train_a", 4, "mse", 81, [0, 2, 1, 2, 0]),
                                        ("interp_train_b", 6, "l1", 82, [4, 0, 3, 1, 2])):
        _, ipol = ref_import.build_reference_dyffusion(system="spring-mesh", model="unet_simple", model_kwargs=base_model,
                                                       horizon=h)
        load_seeded(ipol.model, seed=41)
        from src.utilities.utils import get_loss
        ipol.model.criterion = get_loss(loss_fn)
        ipol.train()
        for p_ in ipol.model.parameters():
            p_.requires_grad_(True)
        g = torch.Generator().manual_seed(23)
        B = len(idx)
        dynamics = torch.randn(B, 1 + h, 4, 10, 10, generator=g)
        cond = torch.rand(B, 1, 10, 10, generator=g)
        sd0 = {k: v.detach().clone() for k, v in ipol.model.state_dict().items()}
        orig_randint = torch.randint
        torch.randint = lambda *a, **k: torch.tensor(idx, dtype=torch.long)
        try:
            with patched_dropout(DropoutSeeded(seed=seed)):
                loss = ipol.get_loss(dict(dynamics=dynamics, condition=cond))
                loss.backward()
        finally:
            torch.randint = orig_randint
        times = torch.tensor(list(ipol.horizon_range))[torch.tensor(idx)]
        arrs = {f"P::{k}": v.numpy() for k, v in sd0.items()}
        arrs.update({f"G::{k}": p_.grad.numpy() for k, p_ in ipol.model.named_parameters()})
        arrs.update({f"B::{k}": v.detach().numpy() for k, v in ipol.model.state_dict().items()
                     if k.endswith("running_mean") or k.endswith("running_var") or k.endswith("num_batches_tracked")})
        hp = dict(horizon=h, window=1, model=base_model, loss_function=loss_fn, dropout_seed=seed, randint=idx,
                  horizon_range=[int(v) for v in ipol.horizon_range])
        np.savez_compressed(os.path.join(HERE, name + ".npz"), dynamics=dynamics.numpy(), cond=cond.numpy(), t=times.numpy(),
                            loss=np.float64(float(loss)), hp=json.dumps(hp), **arrs)
        print(name, "times", times.tolist(), "loss", float(loss), "grad norm",
              float(torch.cat([p_.grad.reshape(-1) for p_ in ipol.model.parameters()]).norm()))



# ------------------------------------------------------------------------------------------------ G6
def gen_fullsize():
    """NS 221x42 h=16 dim=64 (BASELINE config 2): checksums + probe points of the reference rollout, dropout OFF
    (deterministic) and a single interpolator/forecaster forward.  Parameters come from oracle.init.seeded_state."""
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[256, 256], with_time_emb=True, input_dropout=0.0,
              dropout=0.15)
    exp, ipol = ref_import.build_reference_dyffusion(system="navier-stokes", model="unet_simple", model_kwargs=mk,
                                                     horizon=16, diffusion_kwargs=dict(enable_interpolator_dropout=False))
    load_seeded(exp.model.model, seed=101)
    load_seeded(ipol.model, seed=102)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    probes = [(0, 0, 0, 0), (0, 1, 110, 21), (0, 2, 220, 41), (0, 0, 57, 3), (0, 1, 3, 40), (0, 2, 199, 17),
              (0, 0, 128, 30), (0, 1, 64, 8)]
    res = dict(seeds=dict(forecaster=101, interpolator=102, inputs=1), probes=probes, model=mk)
    with torch.no_grad():
        yF = exp.model.model(x0, time=torch.tensor([3.0]), condition=c)
        yI = ipol.model(torch.cat([x0, yF], 1), time=torch.tensor([5.0]), condition=c)
        res["forecaster_fwd"] = dict(mean=float(yF.mean()), std=float(yF.std()), probes=[float(yF[p]) for p in probes])
        res["interpolator_fwd"] = dict(mean=float(yI.mean()), std=float(yI.std()), probes=[float(yI[p]) for p in probes])
        out = exp.predict(x0, condition=c)
    res["rollout"] = {k: dict(mean=float(v.mean()), std=float(v.std()), probes=[float(v[p]) for p in probes])
                      for k, v in out.items()}
    np.savez_compressed(os.path.join(HERE, "fullsize_ns_fields.npz"), yF=yF.numpy().astype(np.float32),
                        yI=yI.numpy().astype(np.float32), t16=out["t16_preds"].numpy().astype(np.float32),
                        t1=out["t1_preds"].numpy().astype(np.float32), t8=out["t8_preds"].numpy().astype(np.float32))
    with open(os.path.join(HERE, "fullsize_checksums.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("fullsize:", {k: (round(v["mean"], 5), round(v["std"], 5)) for k, v in res["rollout"].items()})


def gen_fullsize_oisst():
    """G6 for BASELINE configs[2]: OISST 60x60, C=1, no static condition, `unet.Unet` dim 64 mults (1,2,4) for both networks,
    h=7 with k=25 extra interpolation steps (T=32: 32 forecaster + 61 interpolator forwards), forward_conditioning
    "data+noise" (torch.randn_like patched to a seeded generator), cold sampling, no refinement, dropout off, NB=1.
    Stored: all seven forecast fields; parameters come from oracle.init.seeded_state (seeds in the fixture)."""
    mk = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True, block_dropout=0.0, block_dropout1=0.0, attn_dropout=0.0)
    exp, ipol = ref_import.build_reference_dyffusion(
        system="oisst", model="unet_resnet", model_kwargs=mk, horizon=7,
        diffusion_kwargs=dict(additional_interpolation_steps=25, forward_conditioning="data+noise",
                              refine_intermediate_predictions=False, enable_interpolator_dropout=False))

    def seed_net(net, seed):
        shapes = {k: tuple(v.shape) for k, v in net.state_dict().items()}
        st = oinit.seeded_state(shapes, seed, gain=1.0)
        for k in st:
            if k.endswith(".norm.g"):
                st[k] = 1.0 + 0.1 * torch.randn(shapes[k], generator=torch.Generator().manual_seed(len(k)))
        net.load_state_dict(st, strict=True)

    seed_net(exp.model.model, 201)
    seed_net(ipol.model, 202)
    g = torch.Generator().manual_seed(11)
    x0 = torch.randn(1, 1, 60, 60, generator=g)
    with torch.no_grad(), patched_randn_like(301) as draws:
        out = exp.predict(x0)
    assert exp.model.num_timesteps == 32 and len(draws) == 32 and sorted(out) == [f"t{i}_preds" for i in range(1, 8)]
    np.savez_compressed(os.path.join(HERE, "fullsize_oisst_fields.npz"), x0=x0.numpy(),
                        meta=json.dumps(dict(seeds=dict(forecaster=201, interpolator=202, inputs=11, noise=301),
                                             model=dict(mk, dim_mults=list(mk["dim_mults"])), timesteps=7,
                                             additional_interpolation_steps=25, num_timesteps=32,
                                             forecaster_channels=dict(inputs=exp.model.model.num_input_channels,
                                                                      cond=exp.model.model.num_conditional_channels),
                                             interpolator_channels=dict(inputs=ipol.model.num_input_channels,
                                                                        cond=ipol.model.num_conditional_channels))),
                        **{k: v.numpy().astype(np.float32) for k, v in out.items()})
    print("fullsize OISST:", {k: (round(float(v.mean()), 4), round(float(v.std()), 4)) for k, v in out.items()})


def gen_metrics():
    """Ensemble metrics: outputs of the reference's own numpy functions (src/utilities/evaluation.py);
    its CRPS goes through xskillscore/properscoring, absent here, so only mse / spread-skill are reference-generated."""
    from src.utilities.evaluation import evaluate_ensemble_mse, evaluate_ensemble_spread_skill_ratio

    rng = np.random.default_rng(17)
    for name, (n, b, c, hh, ww) in {"a": (5, 3, 2, 7, 6), "b": (20, 2, 3, 13, 9), "c": (1, 4, 1, 5, 5)}.items():
        truth = rng.normal(size=(b, c, hh, ww)).astype(np.float32)
        preds = (truth[None] * 0.8 + rng.normal(scale=0.5 + 0.1 * n, size=(n, b, c, hh, ww))).astype(np.float32)
        mse = float(evaluate_ensemble_mse(preds, truth))
        ssr = float(evaluate_ensemble_spread_skill_ratio(preds, truth))
        np.savez(os.path.join(HERE, f"metrics_{name}.npz"), preds=preds, targets=truth, mse=mse, ssr=ssr)
        print(f"metrics_{name}: mse {mse:.6f} ssr {ssr:.6f}")


def gen_ckpt_keys():
    """Parameter/buffer names + shapes of the reference's Lightning modules (what a .ckpt's state_dict holds)."""
    mk = dict(dim=4, upsample_dims=[32, 32], with_time_emb=True, outer_sample_mode="bilinear", dropout=0.1)
    exp, ipol = ref_import.build_reference_dyffusion(model_kwargs=mk, horizon=4)
    out = {"model_kwargs": mk, "horizon": 4,
           "forecasting": {k: list(v.shape) for k, v in exp.state_dict().items()},
           "interpolation": {k: list(v.shape) for k, v in ipol.state_dict().items()}}
    with open(os.path.join(HERE, "ckpt_keys.json"), "w") as f:
        json.dump(out, f, indent=0)
    print("ckpt_keys:", len(out["forecasting"]), len(out["interpolation"]))


# ------------------------------------------------------------------------------------------------ boundary conditions
def gen_boundary():
    """`PhysicalSystemsBenchmarkDataModule.boundary_conditions` of the imported reference
    (src/datamodules/physical_systems_benchmark.py:245-297) on seeded inputs.  Stored: the metadata, the seed of the
    prediction tensor, and the (flat index, value) pairs of every element the reference changed -- all other elements must
    come back untouched.  Cases: NS (B,3,221,42) with a float time; NS ensemble stack (N,B,...) with per-batch times (the
    reference's first-dimension indexing); spring-mesh (B,4,10,10) and (N,B,4,10,10)."""
    from types import SimpleNamespace
    from src.datamodules.physical_systems_benchmark import PhysicalSystemsBenchmarkDataModule as DM

    def run(system, shape, B, time, seed):
        g = torch.Generator().manual_seed(seed)
        preds = torch.randn(*shape, generator=g)
        C, Hh, Ww = shape[-3:]
        meta = {"fixed_mask": torch.rand(B, C, Hh, Ww, generator=g) < 0.04}
        if system == "navier-stokes":
            meta["in_velocity"] = 0.5 + torch.rand(B, generator=g)
            vert = torch.zeros(B, 2, Hh, Ww)
            vert[:, 1] = (torch.linspace(0.0, 0.41, Ww)[None, None, :] + 0.001 * torch.rand(B, 1, Ww, generator=g)).expand(B, Hh, Ww)
            vert[:, 0] = torch.linspace(0.0, 2.2, Hh)[None, :, None]
            meta["vertices"] = vert
        else:
            meta["features"] = torch.randn(B, 5, 4, Hh, Ww, generator=g)
        targets = torch.zeros(B, C, Hh, Ww)
        fake = SimpleNamespace(hparams=SimpleNamespace(physical_system=system))
        before = preds.clone()
        out = DM.boundary_conditions(fake, preds, targets, meta, time=time)
        changed = (out != before).reshape(-1).nonzero().reshape(-1)
        arrs = dict(system=system, shape=np.array(shape), B=np.int64(B), seed=np.int64(seed),
                    fixed_mask=np.packbits(meta["fixed_mask"].numpy()), changed_idx=changed.numpy().astype(np.int64),
                    changed_val=out.reshape(-1)[changed].numpy(),
                    time=np.array(time if not torch.is_tensor(time) else time.numpy(), dtype=np.float64))
        if system == "navier-stokes":
            arrs.update(in_velocity=meta["in_velocity"].numpy(), vertex_y=meta["vertices"][:, 1, 0, :].numpy())
        else:
            arrs.update(base_q=meta["features"][:, 0, 2:].numpy())
        print(system, shape, "changed", int(changed.numel()), "of", out.numel())
        return arrs

    np.savez_compressed(os.path.join(HERE, "boundary_ns_b2.npz"), **run("navier-stokes", (2, 3, 221, 42), 2, 0.37, 51))
    np.savez_compressed(os.path.join(HERE, "boundary_ns_n3b2.npz"),
                        **run("navier-stokes", (3, 2, 3, 221, 42), 2, torch.tensor([0.2, 1.5]), 52))
    np.savez_compressed(os.path.join(HERE, "boundary_spring_b3.npz"), **run("spring-mesh", (3, 4, 10, 10), 3, 1.0, 53))
    np.savez_compressed(os.path.join(HERE, "boundary_spring_n2b3.npz"), **run("spring-mesh", (2, 3, 4, 10, 10), 3, 1.0, 54))


# ------------------------------------------------------------------------------------------------ caller contract (C1, 8f-1)
def gen_predict_step():
    """`predict_step` of the imported reference (`_base_experiment.py:700-703` -> `evaluation_step` :484-492 ->
    `_evaluation_step`, forecasting_multi_horizon.py:114-229): spring-mesh batch, N=3 ensemble members, horizon 4 with one
    autoregressive step (prediction horizon 8), the datamodule's boundary conditions applied to every field.
    Stored: the batch, both networks' weights and every array predict_step appends to `_predict_step_outputs`."""
    from types import SimpleNamespace
    from src.datamodules.physical_systems_benchmark import PhysicalSystemsBenchmarkDataModule as DM

    base_model = dict(dim=4, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True,
                      input_dropout=0.0, dropout=0.2)
    N, B, h = 3, 2, 4
    exp, ipol = ref_import.build_reference_dyffusion(system="spring-mesh", model="unet_simple", model_kwargs=base_model,
                                                     horizon=h, diffusion_kwargs=dict(enable_interpolator_dropout=False),
                                                     num_predictions=N)
    load_seeded(exp.model.model, seed=21)
    load_seeded(ipol.model, seed=22)
    exp.hparams.autoregressive_steps = 1
    assert exp.prediction_horizon == 2 * h and exp.num_autoregressive_steps == 1
    fake = SimpleNamespace(hparams=SimpleNamespace(physical_system="spring-mesh"))

    class FakeDM:  # the two datamodule methods evaluation_step touches (_base_experiment.py:486-488)
        def boundary_conditions(self, preds, targets, metadata, time=None):
            return DM.boundary_conditions(fake, preds, targets, metadata, time=time)

        def get_boundary_condition_kwargs(self, batch, batch_idx, split):
            return dict(t0=0.0, dt=1.0)

    exp._datamodule = FakeDM()
    g = torch.Generator().manual_seed(3)
    dyn = torch.randn(B, 1 + 2 * h, 4, 10, 10, generator=g)
    cond = torch.rand(B, 1, 10, 10, generator=g)
    fixed = torch.rand(B, 4, 10, 10, generator=g) < 0.1
    feats = torch.randn(B, 5, 4, 10, 10, generator=g)
    batch = {"dynamics": dyn.clone(), "condition": cond, "metadata": {"fixed_mask": fixed, "features": feats}}
    exp._predict_step_outputs = []
    with torch.no_grad():
        ret = exp.predict_step(batch, 0)
    assert ret is None and len(exp._predict_step_outputs) == 1
    res = exp._predict_step_outputs[0]
    arrs = {f"out::{k}": np.asarray(v) for k, v in res.items()}
    arrs.update({f"F::{k}": v.numpy() for k, v in exp.model.model.state_dict().items()})
    arrs.update({f"I::{k}": v.numpy() for k, v in ipol.model.state_dict().items()})
    hp = dict(timesteps=h, num_input_channels=4, num_predictions=N, B=B, model=base_model, prediction_horizon=2 * h,
              schedule="before_t1_only", interpolate_before_t1=True, sampling_type="cold", time_encoding="dynamics",
              refine_intermediate_predictions=True, forward_conditioning="none", enable_interpolator_dropout=False)
    np.savez_compressed(os.path.join(HERE, "predict_step_spring_ar2.npz"), dynamics=dyn.numpy(), condition=cond.numpy(),
                        fixed_mask=fixed.numpy(), base_q=feats[:, 0, 2:].numpy(), hp=json.dumps(hp), **arrs)
    print("predict_step_spring_ar2:", {k: v.shape for k, v in res.items() if k in ("t1_preds", "t8_preds", "t8_targets")},
          "dynamics scaled by", float(batch["dynamics"].abs().mean() / dyn.abs().mean()))


# ------------------------------------------------------------------------------------------------ stochastic statistics
def gen_ensemble_stats():
    """SURVEY 8c "stochastic mode: ensemble mean/variance within sampling error of the oracle over >= 256 members".
    The imported reference samples a 256-member ensemble (torch's own Bernoulli stream for the interpolator's MC dropout)
    of a dim-64 unet_simple pair on a 23x11 grid (resampled to 64^2), h=4, cold + refine; per-pixel ensemble mean and
    variance of every forecast field are stored.  (a) stats_ens256.npz.
    (b) the same statistics for BASELINE config 2 at full size (NS 221x42, h=16), 8 members: per-horizon scalars in
    fullsize_dropout_stats.json (mean / std over members and pixels, ensemble spread = sqrt(mean per-pixel variance))."""
    mk = dict(dim=64, outer_sample_mode="bilinear", upsample_dims=[64, 64], with_time_emb=True, input_dropout=0.0,
              dropout=0.15)
    exp, ipol = ref_import.build_reference_dyffusion(system="navier-stokes", model="unet_simple", model_kwargs=mk,
                                                     horizon=4, diffusion_kwargs=dict(enable_interpolator_dropout=True))
    load_seeded(exp.model.model, seed=101)
    load_seeded(ipol.model, seed=102)
    g = torch.Generator().manual_seed(4)
    x0 = torch.randn(1, 3, 23, 11, generator=g)
    c = torch.rand(1, 2, 23, 11, generator=g)
    N = 256
    torch.manual_seed(1234)
    outs = []
    with torch.no_grad():
        for _ in range(N // 64):
            outs.append(exp.predict(x0.repeat(64, 1, 1, 1), condition=c.repeat(64, 1, 1, 1)))
    arrs = dict(x0=x0.numpy(), c=c.numpy(), n_members=np.int64(N),
                hp=json.dumps(dict(timesteps=4, model=mk, seeds=dict(forecaster=101, interpolator=102, inputs=4))))
    for k in outs[0]:
        v = torch.cat([o[k] for o in outs], 0).double()
        arrs[f"mean::{k}"] = v.mean(0).float().numpy()
        arrs[f"var::{k}"] = v.var(0, unbiased=True).float().numpy()
        print(k, "ensemble mean std", float(v.mean(0).std()), "spread", float(v.var(0).mean().sqrt()))
    np.savez_compressed(os.path.join(HERE, "stats_ens256.npz"), **arrs)

    mk = dict(mk, upsample_dims=[256, 256])
    exp, ipol = ref_import.build_reference_dyffusion(system="navier-stokes", model="unet_simple", model_kwargs=mk,
                                                     horizon=16, diffusion_kwargs=dict(enable_interpolator_dropout=True))
    load_seeded(exp.model.model, seed=101)
    load_seeded(ipol.model, seed=102)
    g = torch.Generator().manual_seed(1)
    x0 = torch.randn(1, 3, 221, 42, generator=g)
    c = torch.rand(1, 2, 221, 42, generator=g)
    torch.manual_seed(4321)
    M = 8
    with torch.no_grad():
        out = exp.predict(x0.repeat(M, 1, 1, 1), condition=c.repeat(M, 1, 1, 1))
    res = dict(seeds=dict(forecaster=101, interpolator=102, inputs=1, torch=4321), n_members=M, model=mk, rollout={})
    for k, v in out.items():
        v = v.double()
        res["rollout"][k] = dict(mean=float(v.mean()), std=float(v.std()), spread=float(v.var(0, unbiased=True).mean().sqrt()),
                                 ens_mean_std=float(v.mean(0).std()))
    with open(os.path.join(HERE, "fullsize_dropout_stats.json"), "w") as f:
        json.dump(res, f, indent=1)
    print("fullsize dropout stats:", {k: (round(v["std"], 4), round(v["spread"], 4)) for k, v in res["rollout"].items()})


if __name__ == "__main__":
    which = sys.argv[1:] or ["schedules", "nets", "resnet", "samples", "fullsize", "metrics", "ckpt", "plosses", "stats", "boundary", "predict_step", "plosses_train", "fullsize_oisst", "interp_train", "plosses_train_resnet"]
    if "stats" in which:
        gen_ensemble_stats()
    if "boundary" in which:
        gen_boundary()
    if "predict_step" in which:
        gen_predict_step()
    if "plosses" in which:
        gen_plosses()
    if "plosses_train" in which:
        gen_plosses_train()
    if "plosses_train_resnet" in which:
        gen_plosses_train_resnet()
    if "interp_train" in which:
        gen_interp_train()
    if "metrics" in which:
        gen_metrics()
    if "ckpt" in which:
        gen_ckpt_keys()
    if "schedules" in which:
        gen_schedules()
    if "nets" in which:
        gen_nets()
    if "resnet" in which:
        gen_resnet_unets()
    if "samples" in which:
        gen_samples()
    if "fullsize" in which:
        gen_fullsize()
    if "fullsize_oisst" in which:
        gen_fullsize_oisst()
