"""N>1 path on CPU: world_size-2 gloo processes exercise the row sharding + forecast-stack all-gather used by the
ensemble-sharded sampler (dyffusion_amd/distributed.py).  The per-rank rollout itself is the single-GPU path."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dyffusion_amd.distributed import all_gather_rows, sample_sharded, shard_rows


def test_shard_rows_balanced_and_complete():
    for total, world in [(50, 8), (80, 8), (7, 2), (3, 4), (16, 1)]:
        spans = [shard_rows(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and spans[-1][1] == total
        assert all(a[1] == b[0] for a, b in zip(spans[:-1], spans[1:]))
        sizes = [b - a for a, b in spans]
        assert max(sizes) - min(sizes) <= 1
    assert [b - a for a, b in (shard_rows(50, 8, r) for r in range(8))] == [7, 7, 6, 6, 6, 6, 6, 6]


def _fake_rollout(x, c, row0=0):
    """Stands in for model.sample on CPU: every output row is a deterministic function of its own input row and of its
    GLOBAL row index (as the engine's row-keyed dropout streams are)."""
    rows = torch.arange(row0, row0 + x.shape[0], dtype=x.dtype).view(-1, 1, 1, 1)
    return {f"t{i}_preds": x * i + (0 if c is None else c.sum(1, keepdim=True)) + 0.125 * rows for i in range(1, 4)}


class _FakeModel:
    """Duck type of DYffusion for sample_sharded: `sample` + `set_row_offset`."""

    def __init__(self):
        self.row0 = 0

    def set_row_offset(self, first_row):
        self.row0 = int(first_row)

    def sample(self, x, static_condition=None):
        return _fake_rollout(x, static_condition, self.row0)


def _worker(rank, world, port, nb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(nb, 3, 5, 4, generator=g)
    c = torch.rand(nb, 2, 5, 4, generator=g)
    out = sample_sharded(_FakeModel(), x, c)
    want = _fake_rollout(x, c)
    ok = all(torch.equal(out[k], want[k]) for k in want) and sorted(out) == sorted(want)
    lo, hi = shard_rows(nb, world, rank)
    back = all_gather_rows(x[lo:hi], nb, row_dim=0)
    ok = ok and torch.equal(back, x)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nb", [6, 7, 1])  # even shards, uneven shards, a rank that owns no row
def test_sharded_sampling_world2_gloo(nb):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nb, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
