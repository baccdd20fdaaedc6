"""Multi-GPU sharding of the path: superclusters are independent work units
(reference: src/dist.cpp:1738-1903 touches only its own variant ranges), so each rank
owns a disjoint set and the only collective is the final tally all-reduce."""
import numpy as np


def shard_range(n_sc: int, rank: int, world: int):
    """Contiguous, balanced slice [beg, end) of supercluster indices owned by `rank`."""
    return (n_sc * rank) // world, (n_sc * (rank + 1)) // world


def rank_seed(seed: int, rank: int) -> int:
    """Seed of the rank-local synthetic shard (weak scaling: every rank generates its own superclusters)."""
    return seed + 7919 * rank


def tally_from_results(res, var_off) -> np.ndarray:
    """int64[2 callsets][TP,FP,FN] of the phasing each supercluster's distances select
    (sc_phase SWAP -> swap slot 1, else slot 0).  Host mirror of the device kernel k_phase_tally."""
    out = np.zeros((2, 3), np.int64)
    n_sc = len(res.sc_phase)
    for h in range(4):
        sc_of_var = np.repeat(np.arange(n_sc), np.diff(var_off[h]))
        use_swap = res.sc_phase[sc_of_var] == 1
        et = np.where(use_swap, res.errtype[h][1], res.errtype[h][0])
        for e in range(3):
            out[h >> 1, e] += int((et == e).sum())
    return out


def allreduce_tally(tally: np.ndarray, device=None) -> np.ndarray:
    """Sum the per-rank tallies over the default process group (RCCL on GPUs, gloo on CPU)."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.array(tally, dtype=np.int64, copy=True))   # all_reduce is in place: never alias the input
    if device is not None:
        t = t.to(device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t)
    return t.cpu().numpy()


# ----------------------------------------------------------------------------------------------------------------------
# dealing by estimated cells, the all-gather of the per-supercluster phasing, the gather of the per-variant
# records.  Collectives go through torch.distributed's default group (RCCL on GPUs, gloo on CPU); tensors live on `device`.
# ----------------------------------------------------------------------------------------------------------------------
def estimate_cells(batch) -> np.ndarray:
    """The reference's own size estimate of a supercluster, max query hap length x max truth hap length
    (sort_superclusters, cluster.cpp:56-101; its `mem` is this x 20 bytes)."""
    lq = np.maximum(np.diff(batch.hap_off[0]), np.diff(batch.hap_off[1])).astype(np.int64)
    lt = np.maximum(np.diff(batch.hap_off[2]), np.diff(batch.hap_off[3])).astype(np.int64)
    return lq * lt


def deal(cells: np.ndarray, world: int):
    """Supercluster indices of every rank: sorted by estimated cells (largest first, index as tie-break); the largest
    few thousand go greedily to the least loaded rank (LPT), the long tail of small ones is dealt in a snake
    (0..w-1, w-1..0, ...) starting from the least loaded rank.  Every rank gets the same share of the sum and of the huge
    ones; each rank's indices are returned in ascending order.  Deterministic: every rank computes the same partition."""
    cells = np.asarray(cells, dtype=np.int64)
    n = len(cells)
    order = np.lexsort((np.arange(n), -cells))
    owner = np.zeros(n, np.int64)
    head = min(n, 4096)
    load = [0] * world
    for k in range(head):                       # LPT on the head of the distribution
        r = min(range(world), key=lambda j: (load[j], j))
        owner[k] = r
        load[r] += int(cells[order[k]])
    rank_by_load = sorted(range(world), key=lambda j: (load[j], j))
    k = np.arange(n - head)
    lap, pos = k // world, k % world
    snake = np.where(lap % 2 == 0, pos, world - 1 - pos)
    owner[head:] = np.asarray(rank_by_load, np.int64)[snake]
    return [np.sort(order[owner == r]) for r in range(world)]


def _allgather_varlen(x: np.ndarray, device=None):
    """all-gather of one 1-D array per rank (lengths differ) -> list of arrays, one per rank"""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return [np.asarray(x)]
    x = np.ascontiguousarray(x)
    dt_in = x.dtype
    if dt_in in (np.uint32, np.uint16, np.uint64):       # the collectives know signed integers only
        x = x.view({4: np.int32, 2: np.int16, 8: np.int64}[dt_in.itemsize])
    t = torch.from_numpy(x)
    dev = device if device is not None else "cpu"
    n = torch.tensor([t.numel()], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(sizes, n)
    sizes = [int(s.item()) for s in sizes]
    m = max(max(sizes), 1)
    pad = torch.zeros(m, dtype=t.dtype, device=dev)
    pad[:t.numel()] = t.to(dev)
    outs = [torch.zeros(m, dtype=t.dtype, device=dev) for _ in range(world)]
    dist.all_gather(outs, pad)
    return [o[:s].cpu().numpy().view(dt_in) for o, s in zip(outs, sizes)]


def allgather_phase(res_local, idx_local, n_total: int, device=None):
    """(sc_phase, orig_phase_dist, swap_phase_dist) of ALL superclusters on every rank (3 x int32 x n_sc):
    the per-contig phasing (phaseblockData::phase, phase.cpp:285-355) needs every supercluster of the contig, and each
    rank then runs it redundantly (summary.phase)."""
    mine = np.stack([np.asarray(idx_local, np.int32), res_local.sc_phase.astype(np.int32),
                     res_local.orig_phase_dist.astype(np.int32), res_local.swap_phase_dist.astype(np.int32)], axis=1).ravel()
    out = [np.zeros(n_total, np.int32) for _ in range(3)]
    for part in _allgather_varlen(mine, device):
        p = part.reshape(-1, 4)
        for k in range(3):
            out[k][p[:, 0]] = p[:, 1 + k]
    return out


def gather_results(res_local, idx_local, var_off_global, device=None):
    """The per-variant result records and the per-alignment scalars of every rank assembled in supercluster order
    (what the TSV / VCF writers need).  Every rank gets the assembled Results (the collectives are
    all-gathers; a caller that writes on rank 0 ignores the others' copy).  var_off_global: [4] x (n_sc + 1) offsets of
    the undivided batch."""
    from . import _abi as A
    n_sc = len(var_off_global[0]) - 1
    out = A.Results(n_sc, [int(v[-1]) for v in var_off_global])
    idxs = _allgather_varlen(np.asarray(idx_local, np.int64), device)
    for name in ("aln_dist", "aln_end_plane", "aln_beg_plane", "aln_status"):
        for idx, part in zip(idxs, _allgather_varlen(getattr(res_local, name), device)):
            getattr(out, name).reshape(-1, 4)[idx] = part.reshape(-1, 4)
    for name in ("sc_phase", "orig_phase_dist", "swap_phase_dist"):
        for idx, part in zip(idxs, _allgather_varlen(getattr(res_local, name), device)):
            getattr(out, name)[idx] = part
    for h in range(4):
        off = np.asarray(var_off_global[h], np.int64)
        for w in range(2):
            for name, dt in A.Results.PER_VAR:
                parts = _allgather_varlen(getattr(res_local, name)[h][w].view(np.uint8 if dt == np.uint8 else np.int32), device)
                dst = getattr(out, name)[h][w].view(np.uint8 if dt == np.uint8 else np.int32)
                for idx, part in zip(idxs, parts):
                    cnt = off[idx + 1] - off[idx]
                    if cnt.sum():
                        pos = np.repeat(off[idx] - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(int(cnt.sum()))
                        dst[pos] = part
    return out


def subset_per_variant(arr, var_off, idx):
    """rows of a per-variant array (one hap slot) that belong to the superclusters `idx`, in that order"""
    off = np.asarray(var_off, np.int64)
    idx = np.asarray(idx, np.int64)
    cnt = off[idx + 1] - off[idx]
    if int(cnt.sum()) == 0:
        return np.asarray(arr)[:0]
    pos = np.repeat(off[idx] - np.concatenate(([0], np.cumsum(cnt)[:-1])), cnt) + np.arange(int(cnt.sum()))
    return np.asarray(arr)[pos]


def deal_contigs(weights, world: int):
    """contig indices of every rank (command line: a contig is evaluated by one rank): heaviest first to the least loaded"""
    order = sorted(range(len(weights)), key=lambda k: (-int(weights[k]), k))
    load = [0] * world
    out = [[] for _ in range(world)]
    for k in order:
        r = min(range(world), key=lambda j: (load[j], j))
        out[r].append(k)
        load[r] += int(weights[k])
    return [sorted(o) for o in out]


def any_rank(flag: bool, device=None) -> bool:
    """True on every rank iff `flag` is set on at least one (one all-reduce, MAX); no process group: the flag itself"""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int32, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return bool(int(t.item()))
