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
