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


class _FakeStackModel(_FakeModel):
    """... with DYffusion.sample_stack: the forecast stack as one (h, rows, C, H, W) tensor + {slot: key}."""

    def sample_stack(self, x, static_condition=None):
        d = _fake_rollout(x, static_condition, self.row0)
        keys = sorted(d)
        return torch.stack([d[k] for k in keys], 0), dict(enumerate(keys))


def _worker(rank, world, port, nb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = torch.Generator().manual_seed(0)
    x = torch.randn(nb, 3, 5, 4, generator=g)
    c = torch.rand(nb, 2, 5, 4, generator=g)
    want = _fake_rollout(x, c)
    ok = True
    calls, orig = [], dist.all_gather
    dist.all_gather = lambda *a, **k: (calls.append(1), orig(*a, **k))[1]
    for model in (_FakeModel(), _FakeStackModel()):  # dict route and stack route
        del calls[:]
        out = sample_sharded(model, x, c)
        ok = ok and len(calls) == 1  # ONE collective per predict call, whatever the horizon
        ok = ok and all(torch.equal(out[k], want[k]) for k in want) and sorted(out) == sorted(want)
    dist.all_gather = orig
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


def _grad_worker(rank, world, port, q):
    from dyffusion_amd.distributed import all_reduce_gradients
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    params = [torch.nn.Parameter(torch.zeros(s)) for s in [(4, 3, 3, 3), (7,), (5, 2), (1,)]]
    frozen = torch.nn.Parameter(torch.zeros(3))  # no gradient: skipped
    per_rank = []
    for r in range(world):
        g = torch.Generator().manual_seed(100 + r)
        per_rank.append([torch.randn(p.shape, generator=g) for p in params])
    for p, gr in zip(params, per_rank[rank]):
        p.grad = gr.clone()
    calls_small = all_reduce_gradients(params + [frozen], bucket_bytes=64)  # several buckets
    ok = calls_small > 1 and frozen.grad is None
    for k, p in enumerate(params):
        want = sum(per_rank[r][k] for r in range(world)) / world
        ok = ok and torch.allclose(p.grad, want, atol=1e-6)
    for p, gr in zip(params, per_rank[rank]):
        p.grad = gr.clone()
    ok = ok and all_reduce_gradients(params) == 1  # one bucket
    for k, p in enumerate(params):
        ok = ok and torch.allclose(p.grad, sum(per_rank[r][k] for r in range(world)) / world, atol=1e-6)
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_gradient_all_reduce_world2_gloo():
    """Data-parallel training: `all_reduce_gradients` averages the `param.grad` the engine's backward wrote (bucketed)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]


class _CommModel(_FakeStackModel):
    """Duck type of the communicator surface of DYffusion (comm_init / comm_destroy / engine_comm_world)."""

    _engine_opts = {"dtype": "bf16"}

    def __init__(self, fail_init_on_rank=None):
        super().__init__()
        self.fail_init_on_rank, self.world, self.destroyed = fail_init_on_rank, 1, 0

    def comm_init(self, unique_id, rank, world, hw, rows):
        assert unique_id == b"\x07" * 128
        if rank == self.fail_init_on_rank:
            raise RuntimeError("ncclCommInitRank: unhandled system error (test)")
        self.world = world

    def comm_destroy(self):
        self.destroyed += 1
        self.world = 1

    def engine_comm_world(self):
        return self.world


def _comm_worker(rank, world, port, q):
    import dyffusion_amd.engine as E
    from dyffusion_amd.distributed import init_engine_comm

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ok = True
    # (1) rank 0 cannot draw the unique id: every rank returns False, nobody blocks in the broadcast, nobody calls comm_init
    def no_rccl(dtype="bf16"):
        raise NotImplementedError("librccl not found (test)")
    orig = E.HipEngine.comm_unique_id
    E.HipEngine.comm_unique_id = staticmethod(no_rccl)
    m = _CommModel()
    ok = ok and init_engine_comm(m, (5, 4), 6) is False and m.world == 1 and "librccl" in m._comm_error
    # (2) the id exists but ONE rank's dyf_comm_init fails: all ranks agree on False and drop their communicator
    E.HipEngine.comm_unique_id = staticmethod(lambda dtype="bf16": b"\x07" * 128)
    m = _CommModel(fail_init_on_rank=1)
    ok = ok and init_engine_comm(m, (5, 4), 6) is False and m.world == 1 and m.destroyed == 1
    # ... and sample_sharded then takes the torch route by itself (engine_comm_world() == 1), still one collective
    g = torch.Generator().manual_seed(0)
    x, c = torch.randn(6, 3, 5, 4, generator=g), torch.rand(6, 2, 5, 4, generator=g)
    out = sample_sharded(m, x, c)
    want = _fake_rollout(x, c)
    ok = ok and all(torch.equal(out[k], want[k]) for k in want)
    # (3) success on every rank
    m = _CommModel()
    ok = ok and init_engine_comm(m, (5, 4), 6) is True and m.world == world and m.destroyed == 0
    E.HipEngine.comm_unique_id = orig
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


def test_init_engine_comm_is_failure_safe_world2_gloo():
    """a rank that cannot create / join the engine-owned communicator must not leave the others in a collective."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_comm_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(results) == [(0, True), (1, True)]
