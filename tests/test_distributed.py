"""World-size-2 tests (gloo, CPU) of the N > 1 path: superclusters dealt by the reference's size estimate,
one all-reduce of the precision/recall counters, the all-gather of the per-supercluster phasing with the redundant per-rank
Viterbi, and the gather of the per-variant records for the writers.  Per-rank results come from the oracle here (no GPU in
this test); on GPUs bench.py and the command line run the same helpers over RCCL."""
import os
import socket
import sys

import numpy as np
import pytest
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SYN = dict(n_sc=301, len_a=8, len_b=200, len_max=200, seed=77)


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import torch.distributed as dist
    import oracle_lib as O
    from vcfdist_amd import api, shard, summary as S
    dist.init_process_group("gloo", init_method=f"tcp://127.0.0.1:{port}", rank=rank, world_size=world)
    syn = api.Synth(**SYN)
    batch, v = syn.batch(), syn.variants()
    cells = shard.estimate_cells(batch)
    parts = shard.deal(cells, world)
    idx = parts[rank]
    mine = batch.subset(idx)
    res = O.run(mine)
    local = shard.tally_from_results(res, mine.var_off)
    total = shard.allreduce_tally(local)
    # item 1: every rank gets every supercluster's phasing and runs the contig's Viterbi itself
    sc_phase, orig, swp = shard.allgather_phase(res, idx, batch.n_sc)
    phase_set = (np.arange(batch.n_sc) // 40).astype(np.int32)
    pb, sw, fl = S.phase(sc_phase, phase_set, L=O.lib(), prefix="vso")
    # item 2: counters of this rank's superclusters under the global phasing, summed over ranks
    cls = [S.var_class(v.var_type[s], v.var_ref_len[s], v.var_alt_len[s]) for s in range(4)]
    cls_mine = [np.concatenate([c[batch.var_off[s][k]:batch.var_off[s][k + 1]] for k in idx]) if len(idx) else c[:0] for s, c in enumerate(cls)]
    counts = O.oracle_pr_counts(O.lib(), mine.var_off, res, cls_mine, pb[idx])
    counts_all = shard.allreduce_tally(counts)
    # item 3: the per-variant records in supercluster order
    whole = shard.gather_results(res, idx, batch.var_off)
    q.put((rank, idx.tolist(), int(cells[idx].sum()), local.tolist(), total.tolist(), sc_phase.tolist(), pb.tolist(),
           counts_all.tolist(), [whole.errtype[h][w].tolist() for h in range(4) for w in range(2)],
           [whole.credit[h][w].view(np.uint32).tolist() for h in range(4) for w in range(2)], whole.aln_dist.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_deal_allreduce_allgather_gather():
    sys.path.insert(0, ROOT)
    import oracle_lib as O
    from vcfdist_amd import api, shard, summary as S
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
    syn = api.Synth(**SYN)
    batch, v = syn.batch(), syn.variants()
    res = O.run(batch)
    # the deal: disjoint, complete, balanced in estimated cells
    i0, i1 = got[0][1], got[1][1]
    assert sorted(i0 + i1) == list(range(batch.n_sc)) and abs(len(i0) - len(i1)) <= 1
    c0, c1 = got[0][2], got[1][2]
    assert abs(c0 - c1) <= 0.02 * (c0 + c1)
    # every rank sees the same reduced tally, equal to the single-process tally of the whole batch
    whole = shard.tally_from_results(res, batch.var_off)
    assert got[0][4] == got[1][4] == whole.tolist()
    assert (np.array(got[0][3]) + np.array(got[1][3])).tolist() == whole.tolist()
    # all-gathered phasing == single-process phasing on both ranks, and so is the Viterbi run on it
    assert got[0][5] == got[1][5] == res.sc_phase.tolist()
    phase_set = (np.arange(batch.n_sc) // 40).astype(np.int32)
    pb, sw, fl = S.phase(res.sc_phase, phase_set, L=O.lib(), prefix="vso")
    assert got[0][6] == got[1][6] == pb.tolist()
    # summed counters == counters of the undivided batch
    cls = [S.var_class(v.var_type[s], v.var_ref_len[s], v.var_alt_len[s]) for s in range(4)]
    want = O.oracle_pr_counts(O.lib(), batch.var_off, res, cls, pb)
    assert got[0][7] == got[1][7] == want.tolist()
    # gathered per-variant records == single-process results
    flat_e = [res.errtype[h][w].tolist() for h in range(4) for w in range(2)]
    flat_c = [res.credit[h][w].view(np.uint32).tolist() for h in range(4) for w in range(2)]
    assert got[0][8] == got[1][8] == flat_e and got[0][9] == got[1][9] == flat_c
    assert got[0][10] == got[1][10] == res.aln_dist.tolist()


def test_shard_ranges_are_balanced():
    from vcfdist_amd import shard
    for n in (0, 1, 7, 1000003):
        for w in (1, 2, 4, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(r[k][1] == r[k + 1][0] for k in range(w - 1))
            sizes = [e - b for b, e in r]
            assert max(sizes) - min(sizes) <= 1


def test_deal_balances_estimated_cells():
    sys.path.insert(0, ROOT)
    from vcfdist_amd import shard
    rng = np.random.RandomState(4)
    cells = (np.exp(rng.normal(6, 2.0, size=100000))).astype(np.int64) + 1      # heavy tail, like supercluster sizes
    for w in (2, 4, 8):
        parts = shard.deal(cells, w)
        assert sorted(np.concatenate(parts).tolist()) == list(range(len(cells)))
        load = np.array([cells[p].sum() for p in parts], dtype=np.float64)
        assert load.max() - load.min() <= cells.max()                             # as even as the largest item allows
        big = np.argsort(-cells)[:w * 10]
        assert all(abs(np.isin(big, p).sum() - 10) <= 4 for p in parts)           # the huge ones are spread over all ranks


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus 2` started the way the driver starts `--gpus 1` (plain python, no torchrun, WORLD_SIZE unset)
    must become a two-rank job: bench.py re-executes itself under torch.distributed.run.  --plumbing-check stops behind the
    rendezvous and one all-reduce (gloo), so this runs without a GPU; tests/test_gpu_configs.py runs the real thing on the
    GPU box with both ranks on its one GPU."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.launch_command(8, ["--gpus", "8", "--steps", "3"], port=1234)
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and cmd[cmd.index("--nproc-per-node") + 1] == "8"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "8", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--plumbing-check"], env=env, cwd=root,
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line == {"plumbing_check": True, "n_gpus": 2, "rank_sum": 3}


@pytest.mark.gpu
def test_native_collectives_on_a_one_rank_communicator():
    """vpr_allreduce_counts / vpr_allgather_phase (the C ABI's RCCL entry points) on a communicator of one rank: the plumbing --
    run-time resolution of librccl, data types, packing -- gives exactly vpr_pr_counts and the local phasing columns"""
    import torch
    from vcfdist_amd import api, rccl, summary
    if not rccl.available():
        pytest.skip("no RCCL library in this process")
    torch.cuda.set_device(0)
    syn = api.Synth(n_sc=500, len_a=10, len_b=300, len_max=300, seed=77, var_per_base=0.02)
    pr = api.PrecisionRecall()
    res = pr.run(syn.batch())
    cls = syn.var_class()
    comm = rccl.Comm(1, 0, rccl.unique_id())
    try:
        want = summary.pr_counts(pr, cls, None)
        got = rccl.allreduce_counts(pr, comm, cls, None)
        assert want.sum() > 0 and np.array_equal(got, want)
        perm = np.random.RandomState(3).permutation(500).astype(np.int32)      # global indices of the local superclusters
        ph, og, sw = rccl.allgather_phase(pr, comm, perm, 500)
        assert np.array_equal(ph[perm], res.sc_phase) and np.array_equal(og[perm], res.orig_phase_dist) and np.array_equal(sw[perm], res.swap_phase_dist)
        # ONE RCCL in the process, and the C side bound that very copy (PyTorch's own, loaded with local visibility: opening
        # "librccl.so.1" by name beside it would be a second copy, and a communicator made by one used through the other a crash)
        bound, mapped = rccl.library_paths()
        assert len(mapped) == 1, mapped
        assert bound in ("", None) or os.path.realpath(bound) == os.path.realpath(mapped[0]), (bound, mapped)
    finally:
        comm.destroy()


@pytest.mark.gpu
def test_one_rccl_copy_after_a_bench_step_with_the_native_collective():
    """the bench's default run (N = 1: vpr_allreduce_counts on a one-rank communicator) in a fresh process: afterwards exactly one
    librccl is mapped, the one the library bound, and the line says it used the native collective"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, json, io, contextlib; sys.argv = ['bench.py', '--n-sc', '20000', '--steps', '2', '--warmup', '2', '--no-cpu-baseline', "
            "'--no-secondary', '--one-pass-batches', '0']; import bench; bench.main(); "
            "from vcfdist_amd import rccl; print('RCCL_PATHS ' + json.dumps(rccl.library_paths()))")
    out = subprocess.run([sys.executable, "-c", code], cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["collective"].startswith("vpr_allreduce_counts"), line["collective"]
    bound, mapped = json.loads([l for l in out.stdout.splitlines() if l.startswith("RCCL_PATHS ")][-1][len("RCCL_PATHS "):])
    assert len(mapped) == 1 and (not bound or os.path.realpath(bound) == os.path.realpath(mapped[0])), (bound, mapped)
