"""Ensemble-sharded sampling across the GPUs of one node.

Ensemble members / batch items are independent rows of the NB batch dimension for the whole rollout
(`_base_experiment.py:503-538` tiles them, row = n*B + b; no op in the path mixes rows), so the path shards with NO
collective inside the rollout.  One process per GPU; rank r samples the contiguous block of rows `shard_rows` gives it
with its own engine, and ONE exchange -- an all-gather of the forecast stack (RCCL over xGMI with backend "nccl", gloo on
CPU for the tests) -- makes every `t{i}_preds` (NB, C, H, W) available on every rank.  The reference has no inference
collective (it only replicates the whole module under Lightning DDP, `src/configs/trainer/ddp.yaml`).

Results do not depend on the number of GPUs: every rank keeps the SAME seed and tells its engine the global index of its
first row (`set_row_offset`), and the engine's dropout / noise streams are keyed by the global row (csrc/common.h).

Layout of the exchange: every rank samples the same number of rows (`rows_per_rank` = ceil(NB / world); ranks that own
fewer rows -- 50 members on 8 GPUs are 7,7,6,6,6,6,6,6 -- repeat their last row, whose result is dropped), and the whole
local forecast stack `(h, rows_per_rank, C, H, W)` -- one contiguous tensor -- travels in ONE collective per predict call
(NS h=16, 80 rows per rank: one 143 MB message per rank instead of sixteen of 8.9 MB; a ring all-gather over xGMI is
per-link bound, so one large message).  Two routes:

* `exchange="engine"` (default on GPUs once `init_engine_comm` has run): `dyf_sample_gather` -- the ENGINE owns the RCCL
  communicator (SURVEY 8b "Ownership"), enqueues `ncclAllGather` on the rollout's stream right behind the captured graph and
  unpacks `[world][h][rows][...]` to `[h][NB][...]` in one kernel; no torch.distributed call in the predict path, and a host
  without torch can shard through the C ABI alone.
* `exchange="torch"`: one `all_gather_into_tensor` (`all_gather` under gloo, for the CPU tests) of the stack, then one
  transposing copy.
"""
from typing import Dict, List, Optional, Protocol, Tuple

import torch
import torch.distributed as dist
from torch import Tensor


def shard_rows(total_rows: int, world_size: int, rank: int) -> Tuple[int, int]:
    """Contiguous, balanced split: the first `total_rows % world_size` ranks own one extra row (50 rows on 8 GPUs ->
    7,7,6,6,6,6,6,6).  A rank may own no row at all when total_rows < world_size."""
    base, extra = divmod(total_rows, world_size)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def rows_per_rank(total_rows: int, world_size: int) -> int:
    return -(-total_rows // world_size)


def _gather_into(out_field: Tensor, local_field: Tensor, group) -> None:
    """out_field (world * r, ...) <- all-gather of local_field (r, ...), both contiguous."""
    if dist.get_backend(group) == "gloo":  # CPU tests: gloo has no all_gather_into_tensor
        world = dist.get_world_size(group)
        r = local_field.shape[0]
        dist.all_gather([out_field[k * r:(k + 1) * r] for k in range(world)], local_field, group=group)
    else:
        dist.all_gather_into_tensor(out_field, local_field, group=group)


def _valid_rows(total_rows: int, world: int, device) -> Optional[Tensor]:
    """Indices of the real rows inside the padded (world * rows_per_rank) gather layout, or None when nothing is padded."""
    rpr = rows_per_rank(total_rows, world)
    if total_rows == world * rpr:
        return None
    spans = [shard_rows(total_rows, world, r) for r in range(world)]
    return torch.cat([torch.arange(r * rpr, r * rpr + (b - a)) for r, (a, b) in enumerate(spans)]).to(device)


def all_gather_rows(local: Tensor, total_rows: int, group=None, row_dim: int = 1) -> Tensor:
    """All-gather a tensor sharded along `row_dim` (0 or 1) by `shard_rows`.  `local` holds this rank's rows, optionally
    already padded to `rows_per_rank` rows; the result has exactly `total_rows` rows in global order."""
    assert row_dim in (0, 1)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    rpr = rows_per_rank(total_rows, world)
    lo, hi = shard_rows(total_rows, world, rank)
    x = local.unsqueeze(0) if row_dim == 0 else local  # (fields, rows, ...)
    assert x.shape[1] in (hi - lo, rpr), (tuple(x.shape), hi - lo, rpr)
    if x.shape[1] < rpr:  # pad to the common row count (only uneven shards pay this copy)
        pad = torch.zeros((x.shape[0], rpr - x.shape[1], *x.shape[2:]), dtype=x.dtype, device=x.device)
        x = torch.cat([x, pad], 1)
    x = x.contiguous()
    out = torch.empty((x.shape[0], world * rpr, *x.shape[2:]), dtype=x.dtype, device=x.device)
    for i in range(x.shape[0]):
        _gather_into(out[i], x[i], group)
    keep = _valid_rows(total_rows, world, out.device)
    if keep is not None:
        out = out.index_select(1, keep)
    return out.squeeze(0) if row_dim == 0 else out


class _Sampler(Protocol):  # what sample_sharded needs of `DYffusion`
    def sample(self, initial_condition: Tensor, **kwargs) -> Dict[str, Tensor]: ...

    def set_row_offset(self, first_row: int) -> None: ...


def _all_ok(ok: bool, group, device) -> bool:
    """Every rank learns whether ALL ranks succeeded (all-reduce MIN of a flag: 4 bytes; under RCCL the flag lives on the GPU)."""
    on_gpu = dist.get_backend(group) == "nccl"
    flag = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device if on_gpu else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
    return bool(int(flag.item()))


def init_engine_comm(model, hw, total_rows: int, group=None) -> bool:
    """Create the engines' own RCCL communicator over the ranks of `group`: rank 0 draws the unique id (dyf_comm_unique_id),
    torch.distributed only carries those 128 bytes to the other ranks (any side channel would do), every rank calls dyf_comm_init.

    Failure-safe: every collective below is entered by EVERY rank whatever went wrong locally.  Rank 0 broadcasts
    `(ok, id | error text)` -- if it could not draw the id (e.g. no librccl to dlopen) the others see that instead of blocking
    in the broadcast -- and the per-rank `dyf_comm_init` status is agreed with one all-reduce(MIN): either all ranks return
    True and own a communicator, or all return False, none keeps one, and `sample_sharded` takes the torch.distributed route."""
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    from .engine import HipEngine
    box = [None]
    if rank == 0:
        try:
            box = [(True, HipEngine.comm_unique_id(model._engine_opts["dtype"]))]
        except Exception as ex:
            box = [(False, f"{type(ex).__name__}: {ex}")]
    dist.broadcast_object_list(box, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
    ok, payload = box[0]
    if not ok:  # the same verdict on every rank: nobody calls dyf_comm_init, no further collective
        model._comm_error = payload
        return False
    err = None
    try:
        model.comm_init(payload, rank, world, hw, rows_per_rank(total_rows, world))
    except Exception as ex:
        err = f"{type(ex).__name__}: {ex}"
    dev = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if _all_ok(err is None, group, dev):
        return True
    model._comm_error = err or "dyf_comm_init failed on another rank"
    model.comm_destroy()
    return False


def _unpack_stack(full: Tensor, total_rows: int, world: int) -> Tensor:
    """(world, h, rpr, ...) all-gather layout -> (h, total_rows, ...) in global row order (padding rows dropped)."""
    h, rpr = full.shape[1], full.shape[2]
    stack = full.transpose(0, 1).reshape(h, world * rpr, *full.shape[3:])  # one transposing copy
    keep = _valid_rows(total_rows, world, full.device)
    return stack if keep is None else stack.index_select(1, keep)


def sample_sharded(model: _Sampler, initial_condition: Tensor, static_condition: Optional[Tensor] = None,
                   group=None, exchange: Optional[str] = None) -> Dict[str, Tensor]:
    """Every rank passes the FULL (NB, ...) inputs and the same-seeded `model` (a `DYffusion`); it samples only its own
    rows and receives the full `t{i}_preds` dict.  Identical to `model.sample` on one GPU, bit for bit.  ONE collective per
    call; `exchange`: "engine" (dyf_sample_gather: needs `init_engine_comm`), "torch", or None = engine when available."""
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(group) == 1:
        model.set_row_offset(0)
        kw = {} if static_condition is None else {"static_condition": static_condition}
        return model.sample(initial_condition, **kw)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    nb = initial_condition.shape[0]
    rpr = rows_per_rank(nb, world)
    lo, hi = shard_rows(nb, world, rank)
    # every rank runs `rpr` rows (same batch size = same captured graph, uniform collective): ranks that own fewer rows
    # re-run their last row (or row 0 of the batch when they own none); those results are dropped after the gather
    idx = torch.arange(lo, lo + rpr).clamp_max(max(hi - 1, 0)).clamp_max(nb - 1).to(initial_condition.device)
    x = initial_condition.index_select(0, idx) if (hi - lo) < rpr else initial_condition[lo:hi]
    c = None if static_condition is None else \
        (static_condition.index_select(0, idx) if (hi - lo) < rpr else static_condition[lo:hi])
    model.set_row_offset(lo)
    if exchange is None:
        # decided from the LIVE engine: a communicator belongs to the engine it was created on, and `_ensure_engine` replaces
        # the engine when the grid or the batch grows (every rank launches the same rows on the same grid, so all ranks agree)
        live = getattr(model, "engine_comm_world", None)
        exchange = "engine" if live is not None and live() == world and initial_condition.is_cuda else "torch"
    if exchange == "engine":
        return model.sample_gathered(x, c, nb)
    if not hasattr(model, "sample_stack"):  # duck-typed samplers (tests): per-field route over the dict
        local = model.sample(x, **({} if c is None else {"static_condition": c}))
        keys: List[str] = sorted(local, key=lambda k: float(k[1:].split("_")[0]))
        stack, slot_keys = torch.stack([local[k] for k in keys], 0), dict(enumerate(keys))
    else:
        stack, slot_keys = model.sample_stack(x, c)
    stack = stack.contiguous()  # (h, rpr, C, H, W): the engine's forecast stack, already contiguous
    full = torch.empty((world, *stack.shape), dtype=stack.dtype, device=stack.device)
    if dist.get_backend(group) == "gloo":  # CPU tests / plumbing rehearsals: gloo has no all_gather_into_tensor
        if stack.is_cuda:
            # staged through the host HERE, not by gloo: after one gloo collective on device tensors (its own streams + host
            # callbacks) the next replay of a row-grouped rollout (three graphs on three streams) took 107 s instead of 0.65 s on an
            # MI355X shared by two ranks (gpurun_out/shard_oisst.log, round 4); ungrouped rollouts were unaffected
            host = stack.cpu()
            parts = [torch.empty_like(host) for _ in range(world)]
            dist.all_gather(parts, host, group=group)
            for k in range(world):
                full[k].copy_(parts[k])
        else:
            dist.all_gather([full[k] for k in range(world)], stack, group=group)
    else:
        dist.all_gather_into_tensor(full, stack, group=group)  # ONE collective
    out = _unpack_stack(full, nb, world)
    return {key: out[slot] for slot, key in slot_keys.items()}


def all_reduce_gradients(parameters, group=None, bucket_bytes: int = 64 << 20) -> int:
    """Data-parallel training (the reference: Lightning DDP, `src/configs/trainer/ddp.yaml`): average `param.grad` over the ranks.
    The engine's backward writes `param.grad` directly (`EngineLoss`), so torch's DDP hooks never fire; call this between
    `loss.backward()` and `optimizer.step()`.  Gradients are packed into flat buckets of up to `bucket_bytes` (one
    all-reduce each -- over xGMI a ring all-reduce is per-link bound, so few large messages: the 10.3 M parameters of a
    `unet_simple` are a single 41 MB bucket), summed, divided by the world size and unpacked in place.  Parameters without a
    gradient are skipped (they must be the same set on every rank).  Like DDP, BatchNorm statistics stay local to each rank.
    Returns the number of collectives issued."""
    if not (dist.is_available() and dist.is_initialized()):
        return 0
    world = dist.get_world_size(group)
    if world == 1:
        return 0
    grads = [p.grad for p in parameters if p.grad is not None]
    calls, i = 0, 0
    while i < len(grads):
        j, size = i, 0
        while j < len(grads) and (j == i or size + grads[j].numel() * grads[j].element_size() <= bucket_bytes) and \
                grads[j].dtype == grads[i].dtype and grads[j].device == grads[i].device:
            size += grads[j].numel() * grads[j].element_size()
            j += 1
        flat = torch.cat([g.reshape(-1) for g in grads[i:j]])
        dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
        flat.div_(world)
        off = 0
        for g in grads[i:j]:
            g.copy_(flat[off:off + g.numel()].view_as(g))
            off += g.numel()
        calls += 1
        i = j
    return calls
