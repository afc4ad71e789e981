"""Kernel time of the level-0 3x3 convs (conv_up_halo_kernel<5>, 64->64 @60x60) and the GroupNorm chain inside an OISST rollout.
usage: python tools/time_oisst_conv.py [rows]"""
import os
import sys

import torch

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
sys.path.insert(0, ROOT)
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import dyffusion_amd as D  # noqa: E402
from bench import _resnet_state  # noqa: E402

nb = int(sys.argv[1]) if len(sys.argv) > 1 else 300
kw = dict(dim=64, dim_mults=(1, 2, 4), with_time_emb=True)
F = D.Unet(num_input_channels=1, num_output_channels=1, num_conditional_channels=1, block_dropout=0.3, attn_dropout=0.1, **kw)
I = D.Unet(num_input_channels=2, num_output_channels=1, num_conditional_channels=0, block_dropout=0.6, block_dropout1=0.2, attn_dropout=0.6, **kw)
F.load_state_dict(_resnet_state(F, 0, 0.5))
I.load_state_dict(_resnet_state(I, 1, 0.5))
m = D.DYffusion(F, D.InterpolatorHandle(I, 7), timesteps=7, forward_conditioning="data+noise", interpolate_before_t1=True,
                additional_interpolation_steps=25, refine_intermediate_predictions=False, max_batch=nb, row_groups=1,
                dtype=os.environ.get("DYF_OISST_DTYPE") or None)
x0 = torch.randn(nb, 1, 60, 60).cuda()
m.sample(x0)
eng = m._engine
for rep in range(2):
    ms, launches, fl, by = eng.time_kernel_in_rollout(0, nb)
    print(f"rows={nb} conv3x3 level 0: {ms * 1e3:.1f} us avg over {launches} launches, {fl / ms / 1e9:.0f} TFLOP/s")
