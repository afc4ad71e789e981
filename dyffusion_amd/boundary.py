"""Boundary conditions of the physical-systems benchmark on the GPU.

Host-side mirror of `PhysicalSystemsBenchmarkDataModule.boundary_conditions`
(`src/datamodules/physical_systems_benchmark.py:245-297`): same call signature `(preds, targets, metadata, time)`, same
result -- including the reference's indexing of ensemble stacks -- so an instance can be handed to
`MultiHorizonForecastingDYffusion.evaluation_step(boundary_conditions=...)` exactly where the reference passes
`self.datamodule.boundary_conditions` (`_base_experiment.py:486-488`).  The work itself is one masked-write kernel over the
whole `(rows, C, H, W)` tensor (`dyf_apply_boundary_conditions`), with the metadata resident in device memory, instead of a
Python loop over batch elements with boolean-mask `index_put_` calls.
"""
import ctypes as C
import math
from typing import Dict, Optional, Union

import torch
from torch import Tensor

from . import _lib as L
from .engine import HipEngine


class PhysicalSystemsBoundaryConditions:
    def __init__(self, physical_system: str, engine: HipEngine):
        if physical_system not in ("navier-stokes", "spring-mesh"):
            raise NotImplementedError(f"Boundary conditions for {physical_system} not implemented")  # as the reference
        self.physical_system = physical_system
        self.engine = engine
        self._meta_ref = None     # the metadata object itself: an id() alone is reused by CPython once the dict is freed
        self._meta_key = None
        self._meta_dev: Dict[str, Tensor] = {}

    # ------------------------------------------------------------------ metadata -> device tensors (cached per batch)
    def _source_key(self, metadata):
        """Identity of the tensors the device copies were made from: storage address and in-place modification counter.
        Tensors created under `torch.inference_mode()` (Lightning's evaluation loops move the batch to the device inside it)
        have no version counter -- reading `_version` raises -- and cannot be modified in place outside that mode, so the
        address and shape identify them."""
        names = ("fixed_mask", "in_velocity", "vertices") if self.physical_system == "navier-stokes" else ("fixed_mask", "features")
        return tuple((metadata[k].data_ptr(), None if metadata[k].is_inference() else metadata[k]._version,
                      tuple(metadata[k].shape)) for k in names)

    def _prepare(self, metadata, device) -> Dict[str, Tensor]:
        """The cache holds a REFERENCE to the metadata dict it was built from and is hit only by that very object with
        unchanged source tensors -- the evaluation loop applies the conditions to every horizon step of one batch (h calls per
        outer iteration); the next batch's dict is a different object even when CPython gives it the freed one's address."""
        key = (str(device), self._source_key(metadata))
        if metadata is self._meta_ref and key == self._meta_key:
            return self._meta_dev
        d: Dict[str, Tensor] = {"fixed_mask": metadata["fixed_mask"].to(device=device, dtype=torch.uint8).contiguous()}
        if self.physical_system == "navier-stokes":
            d["in_velocity"] = metadata["in_velocity"].to(device=device, dtype=torch.float32).reshape(-1).contiguous()
            d["vertex_y"] = metadata["vertices"][:, 1, 0, :].to(device=device, dtype=torch.float32).contiguous()
        else:
            base_q = metadata["features"][:, 0, 2:].to(device=device, dtype=torch.float32)
            d["boundary"] = torch.cat([torch.zeros_like(base_q), base_q], dim=1).contiguous()
        self._meta_ref, self._meta_key, self._meta_dev = metadata, key, d
        return d

    def _row_meta(self, preds: Tensor, batch_size: int) -> Tensor:
        """Which batch element's metadata the reference applies to every row of `preds` flattened to (rows, C, H, W)."""
        lead = preds.shape[:-3]
        if len(lead) == 1:
            if lead[0] < batch_size:
                raise IndexError(f"index {lead[0]} is out of bounds for dimension 0 with size {lead[0]}")
            rows = torch.full((lead[0],), -1, dtype=torch.int32)
            rows[:batch_size] = torch.arange(batch_size, dtype=torch.int32)
            return rows
        assert len(lead) == 2, f"predictions must be (B, C, H, W) or (N, B, C, H, W), got {tuple(preds.shape)}"
        n, b = lead
        if self.physical_system == "spring-mesh" and b == batch_size:  # preds[:, b_i] (physical_systems_benchmark.py:281-282)
            return torch.arange(b, dtype=torch.int32).repeat(n)
        # preds[b_i, ...]: the FIRST dimension (ensemble member b_i, all its batch items) -- kept as in the reference
        if n < batch_size:
            raise IndexError(f"index {n} is out of bounds for dimension 0 with size {n}")
        rows = torch.full((n, b), -1, dtype=torch.int32)
        rows[:batch_size] = torch.arange(batch_size, dtype=torch.int32)[:, None]
        return rows.reshape(-1)

    def __call__(self, preds: Tensor, targets: Optional[Tensor], metadata, time: Union[float, Tensor] = None) -> Tensor:
        """In place on `preds` (fp32, contiguous, on the GPU), returns it -- like the reference."""
        if not (preds.is_cuda and preds.dtype == torch.float32 and preds.is_contiguous()):
            raise ValueError("preds must be a contiguous fp32 tensor on the GPU")
        batch_size = targets.shape[0] if targets is not None else metadata["fixed_mask"].shape[0]
        d = self._prepare(metadata, preds.device)
        c, h, w = preds.shape[-3:]
        if tuple(d["fixed_mask"].shape[1:]) != (c, h, w):
            raise AssertionError(f"fixed_mask={tuple(d['fixed_mask'].shape[1:])}, predictions={tuple(preds.shape)}")
        if batch_size > d["fixed_mask"].shape[0]:  # the kernel indexes fixed_mask[b] / in_velocity[b] with the row's element
            raise IndexError(f"batch of {batch_size} elements but the metadata holds {d['fixed_mask'].shape[0]}")
        row_meta = self._row_meta(preds, batch_size).to(preds.device)
        a = L.BcArgs()
        a.kind = L.BC_NAVIER_STOKES if self.physical_system == "navier-stokes" else L.BC_SPRING_MESH
        a.n_fields, a.rows, a.channels, a.height, a.width = 1, row_meta.numel(), c, h, w
        a.n_meta = d["fixed_mask"].shape[0]
        a.row_meta_dev, a.fixed_mask_dev = row_meta.data_ptr(), d["fixed_mask"].data_ptr()
        keep = [row_meta]
        if self.physical_system == "navier-stokes":
            if isinstance(time, float):
                tf = torch.tensor([1 - math.exp(-5 * time)], dtype=torch.float32)
                a.times_per_meta = 0
            else:  # one time per batch element (t0 + dt * step, get_boundary_condition_kwargs)
                tf = torch.tensor([1 - math.exp(-5 * time[b].item()) for b in range(batch_size)], dtype=torch.float32)
                a.times_per_meta = 1
            tf = tf.to(preds.device)
            keep.append(tf)
            a.time_factor_dev, a.in_velocity_dev, a.vertex_y_dev = tf.data_ptr(), d["in_velocity"].data_ptr(), d["vertex_y"].data_ptr()
        else:
            a.boundary_dev = d["boundary"].data_ptr()
        eng = self.engine
        eng._check(eng._lib.dyf_apply_boundary_conditions(eng._h, C.byref(a), preds.data_ptr(), eng._stream()))
        return preds
