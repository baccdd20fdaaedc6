"""World-size-2 test (gloo, CPU) of the N>1 path: contiguous sharding of superclusters across ranks
and the single tally all-reduce.  Per-rank results come from the oracle here (no GPU in this test);
on GPUs bench.py runs the same shard/all-reduce helpers over RCCL."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from vcfdist_amd import api, shard
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    batch = api.Synth(n_sc=301, len_a=8, len_b=200, len_max=200, seed=77).batch()
    beg, end = shard.shard_range(batch.n_sc, rank, world)
    mine = batch.subset(np.arange(beg, end))
    res = O.run(mine)
    local = shard.tally_from_results(res, mine.var_off)
    total = shard.allreduce_tally(local)
    q.put((rank, beg, end, local.tolist(), total.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharding_and_tally_allreduce():
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from vcfdist_amd import api, shard
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # shards are disjoint and cover everything
    assert got[0][1] == 0 and got[0][2] == got[1][1] and got[1][2] == 301
    # every rank sees the same reduced tally, equal to the single-process tally of the whole batch
    batch = api.Synth(n_sc=301, len_a=8, len_b=200, len_max=200, seed=77).batch()
    whole = shard.tally_from_results(O.run(batch), batch.var_off)
    assert got[0][4] == got[1][4] == whole.tolist()
    assert (np.array(got[0][3]) + np.array(got[1][3])).tolist() == whole.tolist()


def test_shard_ranges_are_balanced():
    from vcfdist_amd import shard
    for n in (0, 1, 7, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1
