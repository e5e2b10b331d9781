"""world_size-2 gloo test of the N>1 path's rank logic (replicas / prompt sharding / max-over-ranks timing)."""
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


def _worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from videomv_amd.dist import rank_seed, shard_prompts, max_over_ranks, gather_counts
    prompts = [f"p{i}" for i in range(5)]
    mine = shard_prompts(prompts, rank, world)
    full = shard_prompts(prompts, rank, world, replicate=True)
    t = max_over_ranks(1.0 + rank)               # slowest rank defines the job time
    counts = gather_counts(len(mine))
    # per-rank noise streams differ (seed + rank), as in the reference
    g = torch.Generator().manual_seed(rank_seed(11, rank))
    noise = torch.randn(4, generator=g)
    gathered = [torch.zeros(4) for _ in range(world)]
    dist.all_gather(gathered, noise)
    q.put((rank, mine, len(full), t, counts, bool(torch.equal(gathered[0], gathered[1]))))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_replica_logic():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, m0, f0, t0, c0, same0), (r1, m1, f1, t1, c1, same1) = res
    assert m0 == ["p0", "p2", "p4"] and m1 == ["p1", "p3"] and f0 == f1 == 5
    assert t0 == t1 == 2.0
    assert c0 == c1 == [3, 2] and sum(c0) == 5
    assert not same0 and not same1
