"""CPU tests of the oracle (oracle/pr_oracle.cpp): pinned against the reference-produced
toy vector recorded in SURVEY.md Appendix A.1, textbook Levenshtein, an independent dense
dynamic programme, and hand-checked credit assignments."""
import numpy as np
import pytest

import dense_model as M
import oracle_lib as O
from vcfdist_amd import _abi as A

S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL


def toy_variants():
    ref = "ACGTACGTTTTTGGCA"
    qv = [(4, S, "A", "G", 30.0), (8, I, "", "TT", 30.0), (12, D, "GG", "", 30.0)]
    tv = [(4, S, "A", "G", 30.0)]
    return A.Variants.from_sites([ref], [dict(ctg=0, beg=3, end=15, vars=[qv, qv, tv, tv])])


def test_generate_matches_reference_toy_vector():
    """output of the unmodified reference's generate_ptrs_strs on
    ref ACGTACGTTTTTGGCA, region [3,15], SUB@4 A->G, INS@8 TT, DEL@12 GG."""
    b = O.generate(toy_variants())
    assert bytes(b.hap_seq[0]).decode() == "TGCGTTTTTTTCA"
    assert b.hap_ptr[0].tolist() == [0, 1, 2, 3, 4, 4, 4, 5, 6, 7, 8, 11, 12]
    assert b.hap_flag[0].tolist() == [0, 7, 0, 0, 0, 11, 5, 0, 0, 0, 0, 0, 0]
    assert b.ref_ptr[0].tolist() == [0, 1, 2, 3, 4, 7, 8, 9, 10, 10, 10, 11, 12]
    assert b.ref_flag[0].tolist() == [0, 7, 0, 0, 0, 0, 0, 0, 0, 3, 5, 0, 0]
    assert bytes(b.ref_seq).decode() == "TACGTTTTTGGCA"


def test_forward_matches_reference_toy_vector():
    """Same appendix: against a truth carrying only the SUB, s = 0 on all four alignments and
    the end cell is on the QUERY plane (the false INS and DEL are bypassed through REF)."""
    b = O.generate(toy_variants())
    ex = O.Extra(b, want=(0, 0))
    r = O.run(b, extra=ex)
    assert r.aln_dist.tolist() == [0, 0, 0, 0]
    assert r.aln_end_plane.tolist() == [0, 0, 0, 0]
    # the walk leaves the QUERY plane around both false variants
    plane, qri, ti, sync, edit = ex.path_arrays()
    assert plane.tolist() == [0, 0, 0, 0, 0, 1, 0, 0, 0, 1, 1, 0, 0]
    assert edit.sum() == 0
    # credit: SUB is TP with credit 1 (ref_ed 1 -> query_ed 0); INS and DEL are FP in groups of their own
    for h in (0, 1):
        assert r.errtype[h][0].tolist() == [A.ERRTYPE_TP, A.ERRTYPE_FP, A.ERRTYPE_FP]
        assert r.credit[h][0].tolist() == [1.0, 0.0, 0.0]
        assert r.ref_ed[h][0].tolist() == [1, 0, 0]
    for h in (2, 3):
        assert r.errtype[h][0].tolist() == [A.ERRTYPE_TP]
        assert r.callq[h][0].tolist() == [30.0]
    assert r.sc_phase.tolist() == [A.PHASE_NONE]


def lev(a, b):
    prev = list(range(len(b) + 1))
    for i, ca in enumerate(a, 1):
        cur = [i]
        for j, cb in enumerate(b, 1):
            cur.append(min(prev[j] + 1, cur[j - 1] + 1, prev[j - 1] + (ca != cb)))
        prev = cur
    return prev[-1]


def test_edit_distance_is_levenshtein():
    rng = np.random.RandomState(5)
    for _ in range(1500):
        k = rng.randint(2, 5)
        a = bytes(rng.randint(65, 65 + k, rng.randint(0, 14)).astype(np.uint8))
        b = bytes(rng.randint(65, 65 + k, rng.randint(0, 14)).astype(np.uint8))
        assert O.edit_distance(a, b) == lev(a, b), (a, b)
    assert O.edit_distance(b"", b"ACG") == 3 and O.edit_distance(b"AC", b"") == 2


def test_store_phase():
    # dist.cpp:456-469: equal -> NONE; zero protects division; float ratio vs 0.6
    assert O.store_phase([0, 0, 0, 0]) == (A.PHASE_NONE, 0, 0)
    assert O.store_phase([0, 3, 4, 0]) == (A.PHASE_ORIG, 0, 7)
    assert O.store_phase([2, 0, 0, 1]) == (A.PHASE_SWAP, 3, 0)
    assert O.store_phase([5, 1, 1, 5])[0] == A.PHASE_SWAP      # 1 - 2/10 = 0.8 > 0.6
    assert O.store_phase([2, 1, 1, 2])[0] == A.PHASE_NONE      # 1 - 2/4 = 0.5
    assert O.store_phase([1, 3, 3, 1])[0] == A.PHASE_ORIG      # 1 - 2/6 = 0.667


@pytest.mark.parametrize("seed,lmax,rate", [(1, 40, 0.08), (2, 30, 0.2), (4, 14, 0.3)])
def test_dense_model_equals_sparse_oracle(seed, lmax, rate):
    """The dense row-sweep formulation the kernels use gives the oracle's distance, end plane,
    flag bytes on every cell with D <= s, and (outside order-defined swap ties) its backward
    scores / path pointers / begin plane."""
    from vcfdist_amd import api
    syn = api.Synth(n_sc=60, len_a=6, len_b=lmax, len_min=5, len_max=lmax, var_per_base=rate, p_repeat=0.5,
                    p_snp=0.5, seed=seed)
    B = syn.batch()
    for sc in range(B.n_sc):
        one = B.subset([sc])
        for aln in range(4):
            ex = O.Extra(one, want=(0, aln), dump_matrices=True)
            r = O.run(one, extra=ex)
            qs, ts = aln >> 1, 2 + (aln & 1)
            Q, T, R = bytes(one.hap_seq[qs]), bytes(one.hap_seq[ts]), bytes(one.ref_seq)
            Dm, FL, TIE, CH = M.forward(Q, R, T, one.hap_ptr[qs], one.hap_flag[qs], one.ref_ptr[qs],
                                        one.ref_flag[qs], one.hap_flag[ts])
            dq, dr = Dm[0][len(Q) - 1, len(T) - 1], Dm[1][len(R) - 1, len(T) - 1]
            s = min(dq, dr)
            endp = 0 if dq == s else 1
            assert s == r.aln_dist[aln] and endp == r.aln_end_plane[aln]
            for pl in range(2):
                mask = Dm[pl] <= s
                assert (FL[pl][mask] == ex.flags[pl][mask]).all()
                assert not ex.flags[pl][~mask].any()
            if ex.swap_used_conflict_nonmax[aln] == 0:
                SC, PP = M.backward(Q, R, T, one.hap_ptr[qs], one.hap_flag[qs], one.ref_ptr[qs],
                                    one.ref_flag[qs], FL, CH, endp)
                for pl in range(2):
                    assert (SC[pl] == ex.pscore[pl]).all() and (PP[pl] == ex.pptr[pl]).all()
                assert (0 if SC[0][0, 0] >= 0 else 1) == r.aln_beg_plane[aln]


def _forward_of(one, aln):
    ex = O.Extra(one, want=(0, aln), dump_matrices=True)
    r = O.run(one, extra=ex)
    qs, ts = aln >> 1, 2 + (aln & 1)
    Q, T, R = bytes(one.hap_seq[qs]), bytes(one.hap_seq[ts]), bytes(one.ref_seq)
    Dm, FL, _, _ = M.forward(Q, R, T, one.hap_ptr[qs], one.hap_flag[qs], one.ref_ptr[qs], one.ref_flag[qs], one.hap_flag[ts])
    n_src = max(len(x) for x in M.swap_sources(one.ref_ptr[qs], one.ref_flag[qs], len(Q)) +
                M.swap_sources(one.hap_ptr[qs], one.hap_flag[qs], len(R)))
    return ex, r, Dm, FL, (len(Q), len(R), len(T)), n_src


def test_dense_model_equals_sparse_oracle_with_many_swap_sources():
    """runs of up to seven directly adjacent deletion records: up to eight allowed swap sources on the position behind
    the run (what the library's two candidate lists hold); the dense formulation still gives the
    oracle's distance, end plane and flag bytes"""
    from vcfdist_amd import api
    import indel_runs
    B = api.batch_from_variants(indel_runs.indel_run_superclusters(21, n_sc=30))
    most = 0
    for sc in range(B.n_sc):
        one = B.subset([sc])
        for aln in range(4):
            ex, r, Dm, FL, (lq, lr, lt), n_src = _forward_of(one, aln)
            s = min(Dm[0][lq - 1, lt - 1], Dm[1][lr - 1, lt - 1])
            assert s == r.aln_dist[aln]
            for pl in range(2):
                mask = Dm[pl] <= s
                assert (FL[pl][mask] == ex.flags[pl][mask]).all()
            most = max(most, n_src)
    assert most == 8


def test_forward_flags_never_hold_substitution_and_swap():
    """A swap needs equal bases at its target cell (dist.cpp:335-350), and where the bases are equal the diagonal move is a
    match: PTR_SUB and PTR_SWP_MAT never meet in one flag byte of the reference's forward pass.  (The kernels may
    therefore reuse one of the two bits when the other is set.)"""
    from vcfdist_amd import api
    import indel_runs
    batches = [api.Synth(n_sc=40, len_a=6, len_b=40, len_min=5, len_max=40, var_per_base=0.2, p_repeat=0.5, p_snp=0.5, seed=9).batch(),
               api.batch_from_variants(indel_runs.indel_run_superclusters(22, n_sc=20))]
    cells = 0
    for B in batches:
        for sc in range(B.n_sc):
            one = B.subset([sc])
            for aln in range(4):
                ex = O.Extra(one, want=(0, aln), dump_matrices=True)
                O.run(one, extra=ex)
                for pl in range(2):
                    f = ex.flags[pl]
                    assert not (((f & M.F_SUB) != 0) & ((f & M.F_SWP) != 0)).any()
                    cells += int((f != 0).sum())
    assert cells > 10000


def test_identical_haps_are_all_tp():
    """truth == query: every alignment has distance 0 (ORIG and SWAP both 0 for homozygous sets),
    every variant TP with credit 1."""
    from vcfdist_amd import api
    syn = api.Synth(n_sc=40, len_a=20, len_b=200, len_max=200, p_keep=1.0, p_drop=0.0, p_hom=1.0, seed=11)
    b = syn.batch()
    r = O.run(b)
    assert (r.aln_dist == 0).all()
    for h in range(4):
        for w in range(2):
            assert (r.errtype[h][w] == A.ERRTYPE_TP).all()
            assert (r.credit[h][w] == 1.0).all()


def test_dropped_truth_variant_is_fp_and_missing_query_is_fn():
    ref = "ACGTACGTACGTACGTACGT"
    q = [(5, S, "C", "T", 20.0), (12, S, "A", "G", 33.0)]
    t = [(5, S, "C", "T", 50.0)]
    v = A.Variants.from_sites([ref], [dict(ctg=0, beg=4, end=14, vars=[q, q, t, t])])
    r = O.run(O.generate(v))
    assert r.errtype[0][0].tolist() == [A.ERRTYPE_TP, A.ERRTYPE_FP]
    assert r.callq[0][0].tolist() == [20.0, 33.0]
    assert r.errtype[2][0].tolist() == [A.ERRTYPE_TP] and r.callq[2][0].tolist() == [20.0]
    # swap roles: the extra variant is now in the truth -> FN with callq = max_qual
    v2 = A.Variants.from_sites([ref], [dict(ctg=0, beg=4, end=14, vars=[t, t, q, q])])
    r2 = O.run(O.generate(v2))
    assert r2.errtype[2][0].tolist() == [A.ERRTYPE_TP, A.ERRTYPE_FN]
    assert r2.callq[2][0].tolist() == [50.0, 60.0]
    assert r2.aln_dist.tolist() == [1, 1, 1, 1]


HAND_REF = "GATTACAGCCTGCAAAAAAGTAGCATCGGATCTTGACCA"      # positions 13..18: AAAAAA
# (name, supercluster [beg, end], query variants, truth variants, expectations derived BY HAND from the reference's rules):
# per slot (query 0 / truth 2): errtype, credit, ref_ed, query_ed, sync_group, callq; the four distances; the status bits
HAND_CASES = [
    # dist.cpp:1293-1296: credit = 1 - query_ed / ref_ed.  The truth inserts GG, the query G: one edit is left of two ->
    # credit 0.5 < credit_threshold 0.7 -> FP / FN; an FN's quality is max_qual (dist.cpp:1333-1346)
    ("half of an insertion", 8, 30, [(22, I, "", "G", 40.0)], [(22, I, "", "GG", 50.0)],
     dict(q=([A.ERRTYPE_FP], [0.5], [2], [1], [0], [40.0]), t=([A.ERRTYPE_FN], [0.5], [2], [1], [0], [60.0]), dist=[1] * 4, status=0)),
    # three of four deleted bases: 1 - 1/4 = 0.75 >= 0.7 -> TP on both sides, with the query's quality
    ("three quarters of a deletion", 8, 34, [(22, D, "GCA", "", 40.0)], [(22, D, "GCAT", "", 50.0)],
     dict(q=([A.ERRTYPE_TP], [0.75], [4], [1], [0], [40.0]), t=([A.ERRTYPE_TP], [0.75], [4], [1], [0], [40.0]), dist=[1] * 4, status=0)),
    # dist.cpp:949-968: no sync point at an insertion's location, so a SNP and the insertion right behind it share one sync
    # section: ref_ed = 1 + 2, one sync group, the smaller of the two qualities for both (dist.cpp:1284-1288)
    ("SNP and adjacent insertion: one section", 8, 34, [(21, S, "A", "C", 40.0), (22, I, "", "GG", 30.0)],
     [(21, S, "A", "C", 50.0), (22, I, "", "GG", 50.0)],
     dict(q=([A.ERRTYPE_TP] * 2, [1.0] * 2, [3] * 2, [0] * 2, [0, 0], [30.0] * 2), t=([A.ERRTYPE_TP] * 2, [1.0] * 2, [3] * 2, [0] * 2, [0, 0], [30.0] * 2),
          dist=[0] * 4, status=0)),
    # ... while two SNPs with matching bases between them are two sections (groups count from the end of the supercluster)
    ("two SNPs: two sections", 8, 34, [(21, S, "A", "C", 40.0), (24, S, "A", "T", 30.0)], [(21, S, "A", "C", 50.0), (24, S, "A", "T", 50.0)],
     dict(q=([A.ERRTYPE_TP] * 2, [1.0] * 2, [1] * 2, [0] * 2, [1, 0], [40.0, 30.0]), t=([A.ERRTYPE_TP] * 2, [1.0] * 2, [1] * 2, [0] * 2, [1, 0], [40.0, 30.0]),
          dist=[0] * 4, status=0)),
    # dist.cpp:1219-1223: truth variants that cancel (an inserted and a deleted A of one homopolymer: the haplotype IS the
    # reference, and between them truth and reference are a base apart, so there is no sync point): ref_ed = 0 with truth
    # variants -> the WARN, ref_ed = 1 "to prevent divide-by-zero", credit 1 - 0/1 -> TP; no query variant: quality max_qual
    ("truth variants that cancel", 8, 26, [], [(13, I, "", "A", 50.0), (18, D, "A", "", 45.0)],
     dict(q=([], [], [], [], [], []), t=([A.ERRTYPE_TP] * 2, [1.0] * 2, [1] * 2, [0] * 2, [0, 0], [60.0] * 2), dist=[0] * 4, status=A.ST_WARN_ZERO_ED)),
    # the wrong allele: nothing of the one edit is explained -> credit 0
    ("wrong SNP allele", 8, 30, [(21, S, "A", "C", 40.0)], [(21, S, "A", "G", 50.0)],
     dict(q=([A.ERRTYPE_FP], [0.0], [1], [1], [0], [40.0]), t=([A.ERRTYPE_FN], [0.0], [1], [1], [0], [60.0]), dist=[1] * 4, status=0)),
]


def hand_case_variants(case):
    name, beg, end, q, t, want = case
    return A.Variants.from_sites([HAND_REF], [dict(ctg=0, beg=beg, end=end, vars=[q, q, t, t])])


def check_hand_case(case, r):
    name, want = case[0], case[5]
    assert r.aln_dist.tolist() == want["dist"], name
    assert [int(x) & A.ST_WARN_MASK for x in r.aln_status.tolist()] == [want["status"]] * 4, name
    for slot, key in ((0, "q"), (1, "q"), (2, "t"), (3, "t")):
        err, credit, ref_ed, query_ed, sg, callq = want[key]
        for w in range(2):      # homozygous on both sides: ORIG and SWAP phasing agree
            assert r.errtype[slot][w].tolist() == err, (name, slot, w)
            assert r.credit[slot][w].tolist() == credit and r.callq[slot][w].tolist() == callq, (name, slot, w)
            assert r.ref_ed[slot][w].tolist() == ref_ed and r.query_ed[slot][w].tolist() == query_ed, (name, slot, w)
            assert r.sync_group[slot][w].tolist() == sg, (name, slot, w)


@pytest.mark.parametrize("case", HAND_CASES, ids=[c[0] for c in HAND_CASES])
def test_hand_derived_credit_and_sync_rules(case):
    """rules of calc_prec_recall / get_prec_recall_path_sync the demo never reaches (partial credit on either side of the
    threshold, the no-sync-at-an-insertion rule, cancelling truth variants), each with the answer worked out by hand from
    dist.cpp:949-968, :1219-1223, :1284-1346"""
    check_hand_case(case, O.run(O.generate(hand_case_variants(case))))


def test_golden_toy_vector_file():
    """tests/golden/toy_a1.json (reference-produced) against the oracle."""
    import json
    import os
    g = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "toy_a1.json")))
    T = {"SUB": S, "INS": I, "DEL": D}
    qv = [(p, T[t], r, a, 30.0) for p, t, r, a in g["query_variants"]]
    tv = [(p, T[t], r, a, 30.0) for p, t, r, a in g["truth_variants"]]
    v = A.Variants.from_sites([g["ref"]], [dict(ctg=0, beg=g["region"][0], end=g["region"][1], vars=[qv, qv, tv, tv])])
    b = O.generate(v)
    assert bytes(b.hap_seq[0]).decode() == g["query_str"]
    assert b.hap_ptr[0].tolist() == g["q2r_ptrs"] and b.hap_flag[0].tolist() == g["q2r_flags"]
    assert b.ref_ptr[0].tolist() == g["r2q_ptrs"] and b.ref_flag[0].tolist() == g["r2q_flags"]
    r = O.run(b)
    assert r.aln_dist.tolist() == g["s"]
    assert [("QUERY", "REF")[e] for e in r.aln_end_plane] == g["end_plane"]


@pytest.mark.parametrize("fixture", ["regression_seed7.npz", "regression_joint61.npz"])
def test_oracle_regression_fixture(fixture):
    """The oracle still produces tests/golden/regression_seed7.npz / regression_joint61.npz (guards the checker itself)."""
    import os
    sys_path = os.path.join(os.path.dirname(__file__), "golden")
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_regression", os.path.join(sys_path, "make_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    from vcfdist_amd import api
    g = np.load(os.path.join(sys_path, fixture))
    r = O.run(api.Synth(**mod.FIXTURES[fixture]).batch())
    assert np.array_equal(r.aln_dist, g["aln_dist"]) and np.array_equal(r.sc_phase, g["sc_phase"])
    for h in range(4):
        for w in range(2):
            for name, dt in r.PER_VAR:
                x, y = getattr(r, name)[h][w], g[f"{name}_{h}_{w}"]
                if dt == np.float32:
                    x, y = x.view(np.uint32), y.view(np.uint32)
                assert np.array_equal(x, y), (name, h, w)
