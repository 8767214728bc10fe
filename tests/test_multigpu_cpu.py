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


def _grad_worker(rank, world, port, out):
    """Two ranks, a flat 'gradient' buffer each, parameters marked ready in reverse order (as the reverse pass reaches them)."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from posetraj_amd.grad_sync import GradientBuckets, broadcast_parameters
    names = [f"p{i}" for i in range(12)]
    sizes = [100, 3000, 8, 520, 4096, 1, 777, 2048, 64, 1500, 32, 900]
    spans, off = {}, 0
    for n, s in zip(names, sizes):
        spans[n] = (off, s)
        off += (s + 7) // 8 * 8
    dead = {"p2", "p6"}                                        # parameters that never receive a gradient (attn2.to_q ...)
    params = torch.full((off,), float(rank))
    broadcast_parameters(params)
    assert float(params.abs().max()) == 0.0                    # everyone holds rank 0's
    flat = torch.zeros(off)
    gb = GradientBuckets(flat, spans, bucket_bytes=4096 * 4)   # 4096 floats per bucket: parameters straddle bucket borders
    assert gb.world == world and len(gb.bounds) == (off + 4095) // 4096
    early = []
    for step in range(3):
        g = torch.Generator().manual_seed(1000 * step + rank)
        flat.zero_()
        gb.begin()
        for n in reversed(names):
            if n in dead:
                continue
            a, s = spans[n]
            flat[a:a + s] = torch.randn(s, generator=g)        # "the reverse pass writes this parameter's gradient"
            gb.mark_ready(n)
        gb.finish()
        want = torch.zeros(off)
        for r in range(world):
            gr = torch.Generator().manual_seed(1000 * step + r)
            for n in reversed(names):
                if n in dead:
                    continue
                a, s = spans[n]
                want[a:a + s] += torch.randn(s, generator=gr)
        assert torch.allclose(flat, want, atol=1e-6), step
        early.append(gb.launched_early)
    # the first synchronised step learns which parameters produce gradients and sends everything at the end; later steps
    # send every bucket as soon as its last gradient is there
    assert early[0] == 0 and early[1] == len(gb.bounds) and early[2] == len(gb.bounds), early

    def one_step(step, dead_, order, **begin_kw):
        g = torch.Generator().manual_seed(1000 * step + rank)
        flat.zero_()
        gb.begin(**begin_kw)
        for n in order:
            if n not in dead_:
                a, s = spans[n]
                flat[a:a + s] = torch.randn(s, generator=g)
                gb.mark_ready(n)
        gb.finish()
        want = torch.zeros(off)
        for r in range(world):
            gr = torch.Generator().manual_seed(1000 * step + r)
            for n in order:
                if n not in dead_:
                    a, s = spans[n]
                    want[a:a + s] += torch.randn(s, generator=gr)
        assert torch.allclose(flat, want, atol=1e-6), step
        return gb.launched_early

    # a parameter OUTSIDE the learned set starts to produce a gradient (a camera=True ControlNet whose first steps had
    # camera_cond=None).  Reached before its bucket has left, the bucket is held back to finish() - sums exact - and learned:
    nb = len(gb.bounds)
    assert one_step(3, {"p6"}, list(reversed(names))) == nb - 1            # p2 is new: its bucket waits
    assert one_step(4, {"p6"}, list(reversed(names))) == nb                # ... and is part of the set from then on
    # reached AFTER its bucket has left (written under / behind the in-place all-reduce): loud, on every rank, never silent
    late = [n for n in reversed(names) if n != "p6"] + ["p6"]
    try:
        one_step(5, set(), late)
        raised = False
    except RuntimeError as e:
        raised = "after its bucket had been all-reduced" in str(e)
    assert raised
    gb.finish()                                                            # (the collectives both ranks enqueued are collected)
    # the trainer names what decides the step's graph; a change drops the learned set and the step sends at the end
    assert one_step(6, {"p6"}, list(reversed(names)), signature=(True, True)) == 0
    assert one_step(7, {"p6"}, list(reversed(names)), signature=(True, True)) == nb
    gb.reset()
    assert one_step(8, {"p6"}, list(reversed(names)), signature=(True, True)) == 0

    # the ranks DISAGREE about which parameters produce a gradient (rank 0 reaches p9, rank 1 does not) and about the
    # signature.  Buckets leave in one order on every rank, so the sums stay exact whatever each rank decides locally, and what the
    # step learned is agreed on in finish(): the sets differ, every rank forgets its own, the next step sends at the end.
    def lopsided(step, sig):
        flat.zero_()
        gb.begin(signature=sig)
        for n in reversed(names):
            if n == "p6" or (n == "p9" and rank == 1):
                continue
            a, s_ = spans[n]
            flat[a:a + s_] = float(rank + 1) * (step + 1)
            gb.mark_ready(n)
        gb.finish()
        want = torch.zeros(off)
        for n in names:
            a, s_ = spans[n]
            want[a:a + s_] = float(step + 1) * (0.0 if n == "p6" else 1.0 if n == "p9" else 3.0)
        assert torch.equal(flat, want), step
        return gb.launched_early
    assert one_step(9, {"p6"}, list(reversed(names)), signature=(True, True)) == nb       # (learned, in agreement)
    early = lopsided(10, (True, True))                         # rank 1 never completes p9's bucket (sent in finish), rank 0 sends early
    assert early == nb if rank == 0 else early < nb, early     # ... different moments, the same order: exact
    assert gb._expected is None                                # ... and nobody trusts what it saw
    assert lopsided(11, (True, True)) == 0
    assert lopsided(12, (True, rank == 0)) == 0                # signatures differ between the ranks: exact, nothing learned
    assert gb._expected is None
    out.put((rank, True))
    dist.destroy_process_group()


def test_gradient_buckets_world_size_2_gloo():
    """The training step's only collective (data-parallel gradient averaging, posetraj_amd/grad_sync.py) over gloo."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_grad_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert sorted(q.get(timeout=5) for _ in range(world)) == [(0, True), (1, True)]


def test_gradient_buckets_are_inert_without_a_process_group():
    from posetraj_amd.grad_sync import GradientBuckets
    flat = torch.arange(10.0)
    gb = GradientBuckets(flat, {"a": (0, 10)})
    gb.begin(); gb.mark_ready("a"); gb.finish()
    assert gb.world == 1 and torch.equal(flat, torch.arange(10.0))
