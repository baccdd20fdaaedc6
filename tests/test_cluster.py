"""Distance clustering + superclustering: the host implementation behind
include/vcfdist_cluster.h against the CPU restatement of cluster.cpp (oracle/cluster_oracle.cpp), plus
hand-worked cases.  No GPU needed: the reference's code for this step is host code too."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from vcfdist_amd import api, cluster as K

INF = K.SENTINEL


def olib():
    return O.lib()


def test_exports():
    L = C.CDLL(api.build())
    for name in K.EXPORTED:
        assert hasattr(L, name), name


def test_gap_clustering_hand_case():
    # variants at 100, 120 (rlen 1), 300 (rlen 5), 1000; gap 50: reaches [50,171] merge of the first two
    # (120-50 <= 151), [250,355], [950,1051]
    h = K.Hap(pos=[100, 120, 300, 1000], rlen=[1, 1, 5, 1])
    for lib, prefix in ((None, "vcl"), (olib(), "vco")):
        c = K.simple_cluster(h, 0, 50, 0, L=lib, prefix=prefix)
        assert c.var_beg.tolist() == [0, 2, 3, 4]
        assert c.left_reach.tolist() == [50, 250, 950, INF]
        assert c.right_reach.tolist() == [171, 355, 1051, INF]


def test_size_clustering_uses_variant_size():
    # a 200-base deletion at 500 reaches 200 left and right, swallowing the SNP at 350
    h = K.Hap(pos=[350, 500], rlen=[1, 200], type=[1, 3], ref_len=[1, 200], alt_len=[1, 0])
    c_gap = K.simple_cluster(h, 0, 50, 0)
    c_size = K.simple_cluster(h, 1, 50, 0)
    assert c_gap.n == 2 and c_size.n == 1
    assert c_size == K.simple_cluster(h, 1, 50, 0, L=olib(), prefix="vco")
    assert c_size.left_reach[0] == 300 and c_size.right_reach[0] == 900


def test_empty_hap_has_no_table():
    h = K.Hap(pos=[], rlen=[])
    for lib, prefix in ((None, "vcl"), (olib(), "vco")):
        assert K.simple_cluster(h, 0, 50, 0, L=lib, prefix=prefix).n == 0


def test_supercluster_hand_case():
    # Q1 clusters [50,171] [950,1051]; T1 cluster [160,400] chains the first with T2's [390,600]
    q1 = K.Hap(pos=[100, 120, 1000], rlen=[1, 1, 1])
    t1 = K.Hap(pos=[210], rlen=[140])
    t2 = K.Hap(pos=[440], rlen=[110])
    e = K.Hap(pos=[], rlen=[])
    haps = [q1, e, t1, t2]
    cl = [K.simple_cluster(h, 0, 50, 0) for h in haps]
    for lib, prefix in ((None, "vcl"), (olib(), "vco")):
        s = K.supercluster(haps, cl, 10000, L=lib, prefix=prefix)
        assert s.n == 2
        assert [b.tolist() for b in s.brk] == [[0, 1, 2], [0, 0, 0], [0, 1, 1], [0, 1, 1]]
        assert s.beg.tolist() == [99, 999] and s.end.tolist() == [551, 1002]
    assert s.var_off(0).tolist() == [0, 2, 3]


def random_haps(rng, n, span, p_big=0.02):
    haps = []
    for _ in range(4):
        m = rng.randint(0, n + 1)
        pos = np.sort(rng.randint(10, span, size=m))
        typ = rng.choice([1, 2, 3], size=m, p=[0.7, 0.15, 0.15]).astype(np.uint8)
        size = np.where(rng.rand(m) < p_big, rng.randint(20, 400, size=m), rng.randint(1, 6, size=m))
        ref_len = np.where(typ == 1, 1, np.where(typ == 3, size, 0))
        alt_len = np.where(typ == 1, 1, np.where(typ == 2, size, 0))
        haps.append(K.Hap(pos, ref_len, typ, ref_len, alt_len))
    return haps


@pytest.mark.parametrize("seed", range(12))
def test_random_against_oracle(seed):
    rng = np.random.RandomState(seed)
    haps = random_haps(rng, n=rng.choice([0, 3, 40, 400]), span=rng.choice([500, 5000, 60000]))
    size_mode = seed & 1
    gap, rgap = int(rng.choice([5, 50, 200])), int(rng.choice([0, 0, 10]))
    cl = [K.simple_cluster(h, size_mode, gap, rgap) for h in haps]
    ocl = [K.simple_cluster(h, size_mode, gap, rgap, L=olib(), prefix="vco") for h in haps]
    assert all(a == b for a, b in zip(cl, ocl))
    for max_size in (10000, 600, 150):     # the small limits force (repeated) splitting
        s = K.supercluster(haps, cl, max_size)
        o = K.supercluster(haps, cl, max_size, L=olib(), prefix="vco")
        assert s == o, (seed, max_size)
        # properties: superclusters partition the variants in order; pieces within the limit unless unsplittable
        for i in range(4):
            v = s.var_off(i)
            assert v[0] == 0 and v[-1] == len(haps[i].pos) and np.all(np.diff(v) >= 0)
        if s.n_unsplittable == 0 and s.n:
            assert (s.end - s.beg).max() <= max(max_size, 0) or s.n_oversize == 0
        if s.n > 1:
            assert np.all(s.beg[1:] >= s.beg[:-1])


def test_oversize_split_counts():
    # 30 SNPs 40 apart chain into one 1200-base supercluster at gap 50; limit 500 forces splits
    pos = np.arange(30) * 40 + 100
    haps = [K.Hap(pos, np.ones(30)), K.Hap([], []), K.Hap(pos + 7, np.ones(30)), K.Hap([], [])]
    cl = [K.simple_cluster(h, 0, 50, 0) for h in haps]
    s = K.supercluster(haps, cl, 500)
    o = K.supercluster(haps, cl, 500, L=olib(), prefix="vco")
    assert s == o and s.n_oversize == 1 and s.n >= 3
    assert (s.end - s.beg).max() <= 500
    assert s.clusters[0].n > cl[0].n            # clusters were split in place
    assert np.all(s.cells > 0)


def test_invalid_input():
    """A variant type other than SUB/INS/DEL is the reference's "Variant type ... unexpected" ERROR, raised where the
    size of a variant is computed (cluster.cpp:873), i.e. with `-c size` only: product and oracle return VCL_ERR_TYPE there
    and, like the reference, do not look at the type with `-c gap`.  Unsorted positions violate a precondition the
    reference relies on its VCF reader for; the product checks it (VCL_ERR_ARG), the oracle mirrors the reference."""
    badtype = K.Hap([5, 10], [1, 1], [1, 4], [1, 1], [1, 1])
    for L, pre in ((None, "vcl"), (O.lib(), "vco")):
        with pytest.raises(ValueError, match="-2"):
            K.simple_cluster(badtype, 1, 50, 10, L=L, prefix=pre)
        assert K.simple_cluster(badtype, 0, 50, 10, L=L, prefix=pre).n == 1
    with pytest.raises(ValueError, match="-1"):
        K.simple_cluster(K.Hap([10, 5], [1, 1], [1, 1], [1, 1], [1, 1]), 0, 50, 10)
