"""Phasing of the superclusters and the precision/recall summary: host code and device
counters of include/vcfdist_pr.h against the restatement of phase.cpp:271-355 / print.cpp:324-566."""
import numpy as np
import pytest

import oracle_lib as O
from vcfdist_amd import _abi as A
from vcfdist_amd import api, summary as S


def test_phase_hand_cases():
    # all ORIG: nothing to do; one SWAP in the middle of ORIGs inside one phase set: a flip (cheaper than two switches)
    for lib, pre in ((None, "vpr"), (O.lib(), "vso")):
        pb, sw, fl = S.phase([0, 0, 0], [1, 1, 1], L=lib, prefix=pre)
        assert pb.tolist() == [0, 0, 0] and len(sw) == 0 and len(fl) == 0
        pb, sw, fl = S.phase([0, 0, 1, 0, 0], [1, 1, 1, 1, 1], L=lib, prefix=pre)
        assert pb.tolist() == [0, 0, 0, 0, 0] and fl.tolist() == [2] and len(sw) == 0
        # a run of SWAPs: one switch in, one switch out; NONE superclusters follow their block
        pb, sw, fl = S.phase([0, 2, 1, 1, 1, 2, 0, 0], [1] * 8, L=lib, prefix=pre)
        assert pb.tolist()[2:5] == [1, 1, 1] and len(sw) == 2 and len(fl) == 0
        # the same change of phase across a phase-set border costs nothing and is not an error
        pb, sw, fl = S.phase([0, 0, 1, 1], [1, 1, 2, 2], L=lib, prefix=pre)
        assert pb.tolist() == [0, 0, 1, 1] and len(sw) == 0 and len(fl) == 0


@pytest.mark.parametrize("seed", range(10))
def test_phase_random_against_oracle(seed):
    rng = np.random.RandomState(seed)
    n = int(rng.choice([0, 1, 2, 17, 400]))
    sc = rng.choice([0, 1, 2], size=n, p=[0.5, 0.3, 0.2])
    ps = np.cumsum(rng.rand(n) < 0.1)
    a = S.phase(sc, ps)
    b = S.phase(sc, ps, L=O.lib(), prefix="vso")
    assert all(np.array_equal(x, y) for x, y in zip(a, b))


def test_summary_rows_against_oracle():
    rng = np.random.RandomState(3)
    for trial in range(20):
        nq = 61
        counts = np.zeros((2, 4, 3, nq), np.int64)
        # monotone counters like the real ones
        for cs in range(2):
            for t in range(3):
                for e in range(3):
                    counts[cs, t, e] = np.sort(rng.randint(0, 5000 if trial else 1, size=nq))[::-1]
            counts[cs, 3] = counts[cs, :3].sum(axis=0)
        a = S.pr_summary(counts)
        b = S.pr_summary(counts, L=O.lib(), prefix="vso")
        assert [r.key() for r in a] == [r.key() for r in b]


@pytest.mark.gpu
def test_device_counts_and_summary_against_oracle():
    syn = api.Synth(n_sc=4000, len_mode=1, len_a=25.0, len_b=1.0, len_min=4, len_max=2000, seed=31, p_keep=0.8, p_drop=0.1)
    v = syn.variants()
    batch = syn.batch()
    pr = api.PrecisionRecall()
    res = pr.run(batch)
    cls = [S.var_class(v.var_type[s], v.var_ref_len[s], v.var_alt_len[s], sv_threshold=6) for s in range(4)]
    pb, sw, fl = S.phase(res.sc_phase, np.ones(batch.n_sc, np.int32))
    assert all(np.array_equal(x, y) for x, y in zip((pb, sw, fl), S.phase(res.sc_phase, np.ones(batch.n_sc), L=O.lib(), prefix="vso")))
    got = S.pr_counts(pr, cls, pb)
    want = O.oracle_pr_counts(O.lib(), batch.var_off, res, cls, pb)
    assert np.array_equal(got, want)
    assert got[:, 3].sum() > 0 and got[1, 1, 2].max() > 0            # INDEL false negatives exist in this workload
    rows = S.pr_summary(got)
    assert [r.key() for r in rows] == [r.key() for r in S.pr_summary(want, L=O.lib(), prefix="vso")]
    # thresholds NONE: every variant counts once
    tally = pr.tally()
    none_all = rows[6]
    print("ALL NONE: truth_tp %d query_tp %d truth_fn %d query_fp %d  prec %.6f recall %.6f f1 %.6f (%d switches, %d flips)"
          % (none_all.truth_tp, none_all.query_tp, none_all.truth_fn, none_all.query_fp, none_all.precision, none_all.recall,
             none_all.f1_score, len(sw), len(fl)))
    assert none_all.query_tp + none_all.query_fp > 0
    del tally


def test_phase_rejects_an_unknown_phase_value():
    """a supercluster phase other than ORIG / SWAP / NONE is the reference's "Unexpected phase" ERROR (phase.cpp:305)"""
    import pytest
    import oracle_lib as O
    for L, pre in ((None, "vpr"), (O.lib(), "vso")):
        with pytest.raises(ValueError):
            S.phase([0, 3, 1], [0, 0, 0], L=L, prefix=pre)
    pb, sw, fl = S.phase([], [])
    assert len(pb) == 0 and len(sw) == 0 and len(fl) == 0


def test_var_class_helper_matches_the_numpy_rule():
    """vpr_var_class (host helper of the C ABI) against summary.var_class, print.cpp:362-372"""
    import ctypes as C
    from vcfdist_amd import api
    rng = np.random.RandomState(5)
    n = 5000
    t = rng.randint(1, 4, size=n).astype(np.uint8)
    rl = rng.choice([0, 1, 2, 49, 50, 51, 400], size=n).astype(np.int32)
    al = rng.choice([0, 1, 2, 49, 50, 51, 400], size=n).astype(np.int32)
    L = api.lib()
    L.vpr_var_class.restype = None
    L.vpr_var_class.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]
    for thr in (50, 2, 1000):
        out = np.zeros(n, np.uint8)
        L.vpr_var_class(t.ctypes.data, rl.ctypes.data, al.ctypes.data, n, thr, out.ctypes.data)
        assert np.array_equal(out, S.var_class(t, rl, al, thr))
