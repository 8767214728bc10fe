"""World-size-2 checks of the N > 1 path of bench.py on CPU (gloo): rank-sharded independent clips, the start-up
weight broadcast over every packed tensor, max-over-ranks timing.  The data path itself has no collective."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    from posetraj_amd.ops import Packed

    class Layer:                                   # stands in for a packed block (module path checked by packed_tensors)
        pass
    Layer.__module__ = "posetraj_amd.blocks"
    g = torch.Generator().manual_seed(100 + rank)  # every rank starts from different weights, like bench.py
    m = Layer()
    m.conv = Packed(w=torch.randn(128, 64, generator=g), bias=torch.randn(128, generator=g), N=100, K=60)
    m.norm = (torch.randn(64, generator=g), torch.randn(64, generator=g))
    m.children = [Layer()]
    m.children[0].lin = Packed(w=torch.randn(256, 128, generator=g), bias=None, N=256, K=128)
    m.alias = m.children[0].lin.w                  # the same tensor reachable twice must be sent once
    tensors = bench.packed_tensors(m, cuda_only=False)
    assert len(tensors) == 5
    gb, secs, n_coll = bench.broadcast_packed(tensors, src=0, bucket_bytes=64 * 1024)
    assert n_coll == 2 and abs(gb - sum(t.numel() * 4 for t in tensors) / 1e9) < 1e-12       # 160 KiB in 64 KiB buckets
    # every rank now holds rank 0's weights
    ref = torch.Generator().manual_seed(100)
    assert torch.equal(m.conv.w, torch.randn(128, 64, generator=ref))
    # clips are sharded rank::world and every clip is independent (own seed)
    clips = list(range(7))
    mine = bench.shard(clips, rank, world)
    assert mine == clips[rank::world]
    seeds = torch.tensor([1234 + c for c in mine] + [0] * (4 - len(mine)))
    gathered = [torch.zeros(4, dtype=torch.long) for _ in range(world)]
    dist.all_gather(gathered, seeds)
    allseeds = sorted(int(v) for t in gathered for v in t if v)
    assert allseeds == [1234 + c for c in clips]
    # the reported time is the max over ranks
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    assert float(t) == float(world)
    out.put((rank, True))
    dist.destroy_process_group()


def test_world_size_2_gloo():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    got = sorted(q.get(timeout=5) for _ in range(world))
    assert got == [(0, True), (1, True)]


def test_bench_starts_its_own_ranks_when_no_launcher_is_present():
    """`python bench.py --gpus 2` with no WORLD_SIZE in the environment must start two ranks itself:
    the launch + rendezvous + clip-sharding path, on the CPU over gloo."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only"], env=env,
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line == {"world": 2, "clip_seeds": [1234, 1235], "launcher": "self"}
    # a failing rank makes the launcher exit non-zero
    bad = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--rendezvous-only", "--workload", "nope"],
                         env=env, capture_output=True, text=True, timeout=240)
    assert bad.returncode != 0
