"""-m gpu: the N>1 path end to end on ONE GPU: two ranks (gloo rendezvous on 127.0.0.1, both engines on cuda:0) run
`sample_sharded` with MC dropout on -- row sharding, global-row dropout streams, padded uneven shards, the all-gather of the
forecast stack -- and must reproduce, BIT FOR BIT, what a single process samples for the whole batch (results
are invariant to the number of GPUs).  RCCL itself needs one device per rank; the driver's multi-GPU bench covers it."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

HP = dict(timesteps=4, forward_conditioning="none", interpolate_before_t1=True, schedule="before_t1_only",
          sampling_type="cold", refine_intermediate_predictions=True, enable_interpolator_dropout=True)
MK = dict(dim=64, upsample_dims=[64, 64], outer_sample_mode="bilinear", with_time_emb=True, dropout=0.15)


def _inputs(nb):
    g = torch.Generator().manual_seed(77)
    return torch.randn(nb, 3, 23, 11, generator=g), torch.rand(nb, 2, 23, 11, generator=g)


def _model(max_batch):
    from tests.gpu_common import build_dyffusion, seeded_pair
    PF, PI = seeded_pair(64, 3, 2)
    m = build_dyffusion(PF, PI, MK, 3, 2, HP, max_batch=max_batch, batch_invariant=True)
    m.seed(31337)
    return m


def _worker(rank, world, port, nb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dyffusion_amd.distributed import sample_sharded
    torch.cuda.set_device(0)
    x0, c = _inputs(nb)
    out = sample_sharded(_model(nb), x0.cuda(), c.cuda())
    if rank == 0:
        q.put({k: v.cpu() for k, v in out.items()})
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("nb", [6, 5])  # even shards (3 + 3), uneven shards (3 + 2: rank 1 repeats a row, dropped after the gather)
def test_two_ranks_reproduce_the_single_process_fields_bitwise(nb):
    x0, c = _inputs(nb)
    want = {k: v.cpu() for k, v in _model(nb).sample(x0.cuda(), static_condition=c.cuda()).items()}
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, nb, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = q.get(timeout=300)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert sorted(got) == sorted(want)
    for k in want:
        assert got[k].shape == want[k].shape == (nb, 3, 23, 11)
        assert torch.equal(got[k], want[k]), (k, float((got[k] - want[k]).abs().max()))
    # MC dropout is on: the rows really are different members
    assert not torch.equal(want["t4_preds"][0], want["t4_preds"][1])


def _rccl_worker(port, q):
    """World size 1 over backend "nccl" (= RCCL): the collective entry points the N > 1 path uses, on the one GPU of this box."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    from dyffusion_amd.distributed import _gather_into, all_gather_rows, all_reduce_gradients
    g = torch.Generator().manual_seed(5)
    field = torch.randn(6, 3, 23, 11, generator=g).cuda()
    full = torch.empty_like(field)
    _gather_into(full, field, None)              # dist.all_gather_into_tensor on RCCL
    stack = torch.randn(4, 6, 3, 5, 4, generator=g).cuda()
    back = all_gather_rows(stack, 6, row_dim=1)
    p = torch.nn.Parameter(torch.zeros(5, device="cuda"))
    p.grad = torch.arange(5.0, device="cuda")
    calls = all_reduce_gradients([p])            # world 1: no collective
    t = torch.ones(4, device="cuda")
    dist.all_reduce(t)                           # the bench's barrier / max-over-ranks timing reduction
    dist.barrier()
    torch.cuda.synchronize()
    q.put(bool(torch.equal(full, field) and torch.equal(back, stack) and calls == 0 and float(t.sum()) == 4.0))
    dist.destroy_process_group()


def test_rccl_collectives_single_rank():
    import queue
    ctx = mp.get_context("spawn")
    result = None
    for attempt in range(2):  # (one box in a dozen runs sat in RCCL's one-rank bootstrap for 300 s; a fresh process gets a second try)
        with socket.socket() as s:
            s.bind(("127.0.0.1", 0))
            port = s.getsockname()[1]
        q = ctx.Queue()
        p = ctx.Process(target=_rccl_worker, args=(port, q))
        p.start()
        try:
            result = q.get(timeout=240)
        except queue.Empty:
            p.kill()  # exactly the process started here
            p.join(timeout=60)
            continue
        p.join(timeout=120)
        assert p.exitcode == 0
        break
    assert result is True


def _engine_comm_worker(q):
    """The ENGINE-owned exchange through the C ABI alone (no torch.distributed): dyf_comm_unique_id -> dyf_comm_init (rank 0 of 1)
    -> dyf_sample_gather = rollout + ncclAllGather on the rollout's stream + unpack.  World size 1 (RCCL needs one device per
    rank): the gathered stack must equal dyf_sample's, bit for bit, first call (graph capture) and replay."""
    torch.cuda.set_device(0)
    from dyffusion_amd.engine import HipEngine
    nb = 5
    x0, c = _inputs(nb)
    m = _model(nb)
    want = [{k: v.clone() for k, v in m.sample(x0.cuda(), static_condition=c.cuda()).items()} for _ in range(2)]
    m.seed(31337)  # restart the dropout stream: the gathered calls must draw the same masks
    uid = HipEngine.comm_unique_id()
    m.comm_init(uid, 0, 1, (23, 11), nb)
    got = [{k: v.clone() for k, v in m.sample_gathered(x0.cuda(), c.cuda(), nb).items()} for _ in range(2)]
    torch.cuda.synchronize()
    ok = all(sorted(g) == sorted(w) and all(torch.equal(g[k], w[k]) for k in w) for g, w in zip(got, want))
    ok = ok and not torch.equal(got[0]["t4_preds"], got[1]["t4_preds"])  # MC dropout: the replay drew fresh masks
    try:
        m._engine.comm_init(uid, 0, 1)  # a second communicator on the same engine is refused
        ok = False
    except RuntimeError:
        pass
    m._engine.comm_destroy()
    q.put(bool(ok))


def test_engine_owned_rccl_exchange_single_rank():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_engine_comm_worker, args=(q,))
    p.start()
    assert q.get(timeout=300) is True
    p.join(timeout=120)
    assert p.exitcode == 0


def test_gather_unpack_matches_the_row_split():
    """The unpack kernel behind dyf_sample_gather against the host's shard arithmetic: world 1 exercises the identity; the index
    map for world > 1 (uneven shards, padding rows dropped) is the torch route's `_unpack_stack`, checked here on the GPU."""
    from dyffusion_amd.distributed import _unpack_stack, rows_per_rank, shard_rows
    g = torch.Generator().manual_seed(3)
    for world, total in [(2, 5), (8, 50), (4, 3), (3, 9)]:
        rpr = rows_per_rank(total, world)
        full = torch.randn(world, 4, rpr, 3, 6, 5, generator=g).cuda()
        out = _unpack_stack(full, total, world)
        assert out.shape == (4, total, 3, 6, 5)
        for r in range(world):
            lo, hi = shard_rows(total, world, r)
            assert torch.equal(out[:, lo:hi], full[r, :, : hi - lo])


def _grouped_worker(rank, world, port, q):
    """OISST-shaped ResNet-UNet pair, 240 rows sharded over two ranks on ONE GPU: 120 rows per rank = three row groups (three graphs on
    three streams per rank); MC dropout on."""
    import time
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from dyffusion_amd.distributed import sample_sharded
    from tests.test_gpu_row_groups import sharded_group_model
    torch.cuda.set_device(0)
    m, x0 = sharded_group_model(120)
    x0 = x0.cuda()
    times, first = [], None
    for call in range(3):  # first call captures the graphs, the later ones replay them behind an exchange
        torch.cuda.synchronize()
        dist.barrier()
        t0 = time.perf_counter()
        out = sample_sharded(m, x0, None)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
        if call == 0:  # (later calls continue the dropout / noise streams: other draws)
            first = {k: v[[0, 119, 120, 239]].cpu() for k, v in out.items()}
    if rank == 0:
        q.put((m._engine.row_groups, times, first))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_with_row_groups_replay_at_speed_and_match_one_process():
    """Regression (round 4): after ONE gloo collective on device tensors the next replay of a row-grouped rollout took 107 s instead of
    0.65 s on a GPU shared by two ranks; `sample_sharded` now stages a gloo exchange through host tensors.  Rows of both ranks must
    be those of the single-process rollout of the whole batch up to 16-bit rounding (global-row dropout and noise streams: a row
    that drew another row's would differ by O(1); the kernel forms follow the rows per launch, so not bit for bit) and a replay must
    not crawl."""
    from tests.test_gpu_row_groups import sharded_group_model
    m, x0 = sharded_group_model(240)
    want = {k: v[[0, 119, 120, 239]].cpu() for k, v in m.sample(x0.cuda()).items()}
    m._engine.close()
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grouped_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    groups, times, got = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    assert groups == 3
    assert max(times[1:]) < 10.0, times  # ~0.5 s when healthy
    from tests.helpers import rel_rms
    worst = max(rel_rms(got[k][r], want[k][r]) for k in want for r in range(4))
    print(f"two ranks x 120 rows on three groups each vs one process: worst row rel-RMS {worst:.3e}; call times {times}")
    assert worst <= 1e-2
