"""ORACLE (test infrastructure, CPU) -- boundary conditions of the physical-systems benchmark, restated.

Restates /root/reference/src/datamodules/physical_systems_benchmark.py:245-297
(`PhysicalSystemsBenchmarkDataModule.boundary_conditions`), which `_evaluation_step` applies to every predicted field
before it is returned / fed back (src/experiment_types/forecasting_multi_horizon.py:175-182).  Indexing quirks of the
reference are kept on purpose -- the drop-in must give the reference's results:

  * navier-stokes: `preds[b_i, ..., mask] = 0` and `preds[b_i, ..., left] = inflow` index the FIRST dimension of `preds`
    with the batch index b_i.  For an ensemble stack (N, B, C, H, W) that is the ensemble member b_i (all its B batch items)
    -- members >= B are never touched, and B > N raises IndexError.
  * spring-mesh: (N, B, ...) stacks with shape[1] == batch size are handled per batch item (`preds[:, b_i]`); anything else
    indexes the first dimension like navier-stokes.

Parity: pinned against tests/golden/boundary_*.npz (outputs of the imported reference method) in
tests/test_oracle_boundary.py.  Only tests/ may import this package.
"""
import math
from typing import Dict, Union

import torch
from torch import Tensor


def navier_stokes_inflow(in_velocity: float, vertex_y: Tensor, t: float) -> Tensor:
    """physical_systems_benchmark.py:266-269, evaluated left to right in fp32 like the reference's tensor expression."""
    return in_velocity * 4 * vertex_y * (0.41 - vertex_y) / (0.41 * 0.41) * (1 - math.exp(-5 * t))


def boundary_conditions(physical_system: str, preds: Tensor, targets: Tensor, metadata: Dict[str, Tensor],
                        time: Union[float, Tensor] = None) -> Tensor:
    """In place on `preds`, returns it (as the reference does)."""
    batch_size = targets.shape[0]
    if physical_system == "navier-stokes":
        for b_i in range(batch_size):
            t_i = time if isinstance(time, float) else time[b_i].item()
            in_velocity = float(metadata["in_velocity"][b_i].item())
            fixed_mask = metadata["fixed_mask"][b_i, ...]
            assert fixed_mask.shape == preds.shape[-3:], f"fixed_mask={fixed_mask.shape}, predictions={preds.shape}"
            vertex_y = metadata["vertices"][b_i, 1, 0, :]
            left = torch.zeros(tuple(preds.shape[-3:]), dtype=torch.bool)
            left[0, 0, :] = True
            preds[b_i, ..., fixed_mask] = 0
            preds[b_i, ..., left] = navier_stokes_inflow(in_velocity, vertex_y, t_i).unsqueeze(0)
    elif physical_system == "spring-mesh":
        for b_i in range(batch_size):
            fixed_mask_pq = metadata["fixed_mask"][b_i]
            assert fixed_mask_pq.shape[0] == 4
            base_q = metadata["features"][b_i, 0, 2:]
            boundary = torch.cat([torch.zeros_like(base_q), base_q], dim=0)
            if preds.ndim == 5 and preds.shape[1] == batch_size:
                preds[:, b_i, ...] = torch.where(fixed_mask_pq, boundary, preds[:, b_i, ...])
            else:
                preds[b_i, ...] = torch.where(fixed_mask_pq, boundary, preds[b_i, ...])
    else:
        raise NotImplementedError(f"Boundary conditions for {physical_system} not implemented")
    return preds
