"""2 ranks (gloo) on ONE GPU: the OISST sharded step of bench.py, per-call times. env DYF_GN_FUSED, MODE=sharded|plain|stack"""
import os, sys, time
sys.path.insert(0, os.getcwd())
from tools._forms import forward_env_forms  # noqa: E402

forward_env_forms()  # DYF_* switches of this run -> dyf_debug_set_form
import torch, torch.distributed as dist
import bench
from dyffusion_amd.distributed import sample_sharded, shard_rows, rows_per_rank
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
mode = os.environ.get("MODE", "sharded")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("gloo", rank=rank, world_size=world)
rows = 300
rpr = rows_per_rank(rows, world)
m = bench.oisst_model(rpr)[0]
xf = torch.randn(rows, 1, 60, 60, generator=torch.Generator().manual_seed(3)).to(dev)
m._ensure_engine((60, 60), rpr)
lo, hi = shard_rows(rows, world, rank)
for c in range(int(os.environ.get("CALLS", "3"))):
    torch.cuda.synchronize(); dist.barrier()
    t0 = time.perf_counter()
    if mode == "sharded":
        out = sample_sharded(m, xf, None, exchange="torch")
    elif mode == "stack":
        m.set_row_offset(lo); out = m.sample_stack(xf[lo:hi], None)
    else:
        m.set_row_offset(lo); out = m.sample(xf[lo:hi])
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"[{mode} fused={os.environ.get('DYF_GN_FUSED','1')} rank {rank}] call {c}: issue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms", flush=True)
dist.destroy_process_group()
