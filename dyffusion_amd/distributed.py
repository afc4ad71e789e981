"""Ensemble-sharded sampling across the GPUs of one node.

Ensemble members / batch items are independent rows of the NB batch dimension for the whole rollout
(`_base_experiment.py:503-538` tiles them; no op in the path mixes rows), so the path shards with NO collective
inside the rollout.  Each rank (one process per GPU) samples its contiguous block of rows with its own engine and
dropout sub-stream; ONE all-gather of the forecast stack (RCCL over xGMI with backend "nccl", gloo on CPU for the
tests) makes the full `(h, NB, C, H, W)` stack available on every rank.  The reference has no inference collective
(it only replicates the whole module under Lightning DDP, `src/configs/trainer/ddp.yaml`).
"""
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first `total_rows % world_size` ranks own one extra row (50 rows on 8 GPUs ->
    7,7,6,6,6,6,6,6)."""
    base, extra = divmod(total_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def all_gather_rows(local: Tensor, total_rows: int, group=None, row_dim: int = 1) -> Tensor:
    """All-gather tensors that are sharded along `row_dim` by `shard_rows` (uneven shards are padded to the largest
    one for the collective and trimmed afterwards)."""
    world = dist.get_world_size(group)
    counts = [shard_rows(total_rows, world, r)[1] - shard_rows(total_rows, world, r)[0] for r in range(world)]
    cmax = max(counts)
    x = local.movedim(row_dim, 0).contiguous()
    assert x.shape[0] == counts[dist.get_rank(group)], (x.shape, counts)
    if x.shape[0] < cmax:
        pad = torch.zeros((cmax - x.shape[0], *x.shape[1:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], 0)
    out = torch.empty((world * cmax, *x.shape[1:]), dtype=x.dtype, device=x.device)
    if dist.get_backend(group) == "gloo":
        parts = [torch.empty_like(x) for _ in range(world)]
        dist.all_gather(parts, x, group=group)
        out = torch.cat(parts, 0)
    else:
        dist.all_gather_into_tensor(out, x, group=group)
    pieces = [out[r * cmax: r * cmax + counts[r]] for r in range(world)]
    return torch.cat(pieces, 0).movedim(0, row_dim)


def sample_sharded(sample_fn: Callable[[Tensor, Optional[Tensor]], Dict[str, Tensor]], initial_condition: Tensor,
                   static_condition: Optional[Tensor] = None, group=None) -> Dict[str, Tensor]:
    """Every rank passes the FULL (NB, ...) inputs; it samples only its own rows with `sample_fn` (e.g.
    `lambda x, c: model.sample(x, static_condition=c)`) and receives the full `t{i}_preds` dict."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        return sample_fn(initial_condition, static_condition)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nb = initial_condition.shape[0]
    lo, hi = shard_rows(nb, world, rank)
    local = sample_fn(initial_condition[lo:hi], None if static_condition is None else static_condition[lo:hi])
    keys: List[str] = sorted(local, key=lambda k: int(k[1:].split("_")[0]))
    stack = torch.stack([local[k] for k in keys], 0)  # (h, rows_local, C, H, W)
    full = all_gather_rows(stack, nb, group=group, row_dim=1)
    return {k: full[i] for i, k in enumerate(keys)}
