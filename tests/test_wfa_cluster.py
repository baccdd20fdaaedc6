"""biWFA dependency clustering: the HIP implementation behind vcl_wfa_cluster against the CPU
restatement of wf_swg_cluster / wf_swg_align / wf_swg_max_reach (oracle/wfa_oracle.cpp).  The oracle's own sanity
checks run on the CPU; the parity tests need a GPU."""
import numpy as np
import pytest

import oracle_lib as O
from vcfdist_amd import api, cluster as K


def rand_contig(rng, n, p_repeat=0.3):
    out = []
    while sum(len(x) for x in out) < n:
        if rng.rand() < p_repeat:
            unit = "".join(rng.choice(list("ACGT"), size=rng.randint(1, 5)))
            out.append(unit * rng.randint(3, 15))
        else:
            out.append("".join(rng.choice(list("ACGT"), size=rng.randint(10, 60))))
    return "".join(out)[:n]


def rand_hap(rng, ctg, n_var, max_indel=8):
    """non-overlapping SUB / INS / DEL on the contig, away from its ends"""
    pos, typ, refs, alts = [], [], [], []
    p = 30
    for _ in range(n_var):
        p += int(rng.choice([1, 2, 3, 5, 8, 20, 60, 200]))
        if p > len(ctg) - 60:
            break
        t = int(rng.choice([1, 2, 3], p=[0.6, 0.2, 0.2]))
        if t == 1:
            r = ctg[p]; a = rng.choice([c for c in "ACGT" if c != r])
        elif t == 2:
            k = rng.randint(1, max_indel + 1)
            r = ""; a = ctg[p:p + k] if rng.rand() < 0.5 else "".join(rng.choice(list("ACGT"), size=k))
        else:
            k = rng.randint(1, max_indel + 1)
            r = ctg[p:p + k]; a = ""
        pos.append(p); typ.append(t); refs.append(r); alts.append(a)
        p += len(r) + 1
    return K.HapSeq(pos, typ, refs, alts)


def test_oracle_isolated_snp_stays_alone():
    rng = np.random.RandomState(1)
    ctg = "".join(rng.choice(list("ACGT"), size=400))
    p1, p2 = 100, 300
    hap = K.HapSeq([p1, p2], [1, 1], [ctg[p1], ctg[p2]], ["A" if ctg[p1] != "A" else "C", "A" if ctg[p2] != "A" else "C"])
    c, st = K.wfa_cluster(hap, ctg, L=O.lib(), prefix="vco")
    assert c.n == 2 and c.var_beg.tolist() == [0, 1, 2]
    assert c.left_reach[0] <= p1 and c.right_reach[0] >= p1 + 1 and c.right_reach[0] + 10 < c.left_reach[1]
    assert st["iterations"] >= 1 and st["align_calls"] >= 2


def test_oracle_repeat_indels_merge():
    # two 2-base deletions in one long dinucleotide repeat can be slid onto each other: one cluster
    ctg = "ACGTTGCA" * 5 + "AC" * 40 + "TGCATTGA" * 5
    s = 40
    hap = K.HapSeq([s + 10, s + 50], [3, 3], ["AC", "AC"], ["", ""])
    c, _ = K.wfa_cluster(hap, ctg, L=O.lib(), prefix="vco")
    assert c.n == 1 and c.var_beg.tolist() == [0, 2]


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_matches_oracle(seed):
    rng = np.random.RandomState(100 + seed)
    ctg = rand_contig(rng, int(rng.choice([800, 3000, 12000])))
    hap = rand_hap(rng, ctg, int(rng.choice([1, 5, 40, 300])), max_indel=int(rng.choice([3, 8, 25])))
    kw = dict(sub=5, open=6, extend=2, max_cluster_itrs=int(rng.choice([1, 4])), reach_min_gap=int(rng.choice([0, 10])))
    if seed == 7:
        kw.update(sub=3, open=2, extend=1)        # the evaluation penalties (globals.h:53-55)
    want, so = K.wfa_cluster(hap, ctg, L=O.lib(), prefix="vco", **kw)
    got, sg = K.wfa_cluster(hap, ctg, **kw)
    assert got == want, (seed, got.var_beg.tolist()[:10], want.var_beg.tolist()[:10])
    assert (sg["iterations"], sg["align_calls"], sg["reach_calls"]) == (so["iterations"], so["align_calls"], so["reach_calls"])
    print(f"seed {seed}: {len(hap.pos)} variants -> {got.n} clusters, {sg['iterations']} iterations, "
          f"{sg['reach_calls']} reach calls, {sg['ms_device']:.2f} ms on the device")


@pytest.mark.gpu
def test_gpu_contig_edges():
    # variants close to both contig ends: the doubling window hits the edge (the `beg_pos == 0` / `end_pos == len` exits)
    rng = np.random.RandomState(5)
    ctg = "AC" * 30 + "".join(rng.choice(list("ACGT"), size=80)) + "GT" * 30
    hap = K.HapSeq([6, 150, 190], [3, 1, 2], ["AC", ctg[150], ""], ["", "A" if ctg[150] != "A" else "C", "GT"])
    want, _ = K.wfa_cluster(hap, ctg, L=O.lib(), prefix="vco")
    got, _ = K.wfa_cluster(hap, ctg)
    assert got == want
