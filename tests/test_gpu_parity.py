"""GPU parity tests: the HIP path behind the C ABI vs the CPU oracle, bit-exact on every
integer, byte and float32 bit pattern -- nothing is masked.  Where the reference's result depends on
its container order (several optimal swap predecessors, dist.cpp:347,376: the last writer wins) the
oracle keeps the reference's containers and the library replays their order on the device
(pr_tie.hip); both sides raise VPR_ST_SWAP_TIE for such alignments and must agree on everything."""
import numpy as np
import pytest

import oracle_lib as O
from vcfdist_amd import _abi as A
from vcfdist_amd import api

pytestmark = pytest.mark.gpu


def compare(batch, cfg=None, variants_struct=None):
    """every result array of the library against the oracle, bit for bit; returns the number of alignments in which
    the oracle's containers kept a swap predecessor other than the highest index (the ones only the replay gets right).
    variants_struct: upload the batch's variant tables instead (generate_ptrs_strs on the device, vpr_upload_variants)."""
    ex = O.Extra(batch)
    want = O.run(batch, extra=ex)
    pr = api.PrecisionRecall(cfg) if cfg is not None else api.PrecisionRecall()
    if variants_struct is None:
        got = pr.run(batch)
    else:
        pr.upload_variants(variants_struct, batch)
        pr.execute()
        got = pr.download()
    for f in ("aln_dist", "aln_end_plane", "aln_beg_plane", "aln_status", "sc_phase", "orig_phase_dist", "swap_phase_dist"):
        a, b = getattr(got, f), getattr(want, f)
        assert np.array_equal(a, b), (f, np.flatnonzero(a != b)[:8])
    for h in range(4):
        for w in range(2):
            for name, dt in A.Results.PER_VAR:
                x, y = getattr(got, name)[h][w], getattr(want, name)[h][w]
                if dt == np.float32:
                    x, y = x.view(np.uint32), y.view(np.uint32)
                assert np.array_equal(x, y), (name, h, w, np.flatnonzero(x != y)[:8])
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    assert pr.timing().n_tie_replays >= n_tie
    return got, want, int((ex.swap_used_conflict_nonmax > 0).sum()), pr


@pytest.mark.parametrize("band_mode", [1, 3, 2, 0])
@pytest.mark.parametrize("name,kw", [
    ("tiny_repeats", dict(n_sc=400, len_a=6, len_b=60, len_min=5, len_max=60, seed=1, var_per_base=0.08, p_snp=0.5, p_repeat=0.5)),
    ("c64x4", dict(n_sc=200, len_a=65, len_b=250, len_min=65, len_max=250, seed=2, var_per_base=0.03)),
    ("c256x4", dict(n_sc=60, len_a=260, len_b=1000, len_min=260, len_max=1000, seed=3, var_per_base=0.01)),
    ("c256x8", dict(n_sc=24, len_a=1030, len_b=2000, len_min=1030, len_max=2000, seed=4)),
    ("c1024x8", dict(n_sc=8, len_a=2100, len_b=5000, len_min=2100, len_max=5000, seed=5)),
    ("mixed", dict(n_sc=300, len_a=8, len_b=2500, len_min=5, len_max=2500, seed=6)),
    ("wgs_like", dict(n_sc=3000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=3000, seed=8)),
])
def test_parity_by_kernel_class(name, kw, band_mode):
    batch = api.Synth(**kw).batch()
    got, want, ntie, pr = compare(batch, A.default_config(band_mode=band_mode))
    t = pr.timing()
    print(f"{name} band_mode={band_mode}: {batch.n_sc} sc, {batch.dense_cells():.3e} dense cells, "
          f"{t.cells_touched:.3e} touched, {t.n_band_retries} retries, kernels {t.ms_total:.2f} ms, "
          f"{ntie} alignments decided by the container order")


@pytest.mark.parametrize("band_mode", [1, 0])
def test_big_class_1024x16_single(band_mode):
    batch = api.Synth(n_sc=1, len_mode=2, len_a=9000.0, len_min=9000, len_max=9000, seed=9).batch()
    compare(batch, A.default_config(band_mode=band_mode))


def test_band_retries_reach_wider_windows_and_dense():
    """Dropping most truth variants of indel-rich haps makes s large, so 64-cell windows fail the exit
    test and the alignment is re-run with wider windows / the dense kernels; results stay exact."""
    batch = api.Synth(n_sc=40, len_a=400, len_b=3000, len_min=400, len_max=3000, seed=17, var_per_base=0.05,
                      p_snp=0.2, indel_mean=12.0, p_keep=0.3, p_drop=0.6).batch()
    got, want, ntie, pr = compare(batch)
    assert pr.timing().n_band_retries > 0


def test_minimum_sizes_and_empty_haps():
    ref = "ACGTTGCAACGT"
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    scs = [
        dict(ctg=0, beg=2, end=4, vars=[[(3, S, "T", "A", 9.0)], [], [], []]),            # one SNP, only query hap 1
        dict(ctg=0, beg=2, end=4, vars=[[], [], [(3, S, "T", "A", 9.0)], []]),            # only truth hap 1
        dict(ctg=0, beg=4, end=6, vars=[[(5, I, "", "GG", 5.0)], [(5, I, "", "GG", 5.0)],
                                         [(5, I, "", "GG", 7.0)], [(5, I, "", "G", 7.0)]]),   # INS, partial match
        dict(ctg=0, beg=5, end=9, vars=[[(6, D, "CA", "", 5.0)], [], [(6, D, "CA", "", 7.0)], [(6, D, "C", "", 7.0)]]),
        dict(ctg=0, beg=1, end=10, vars=[[], [], [], []]),                                # no variants at all
    ]
    v = A.Variants.from_sites([ref], scs)
    batch = api.batch_from_variants(v)
    compare(batch)


def test_superclusters_at_the_contig_end():
    """a variant on one of the contig's last two bases: the region (end = pos + rlen + 1, cluster.cpp:595) is cut at the
    last base (include/vcfdist_pr.h, vpr_batch_from_variants); library and oracle agree on every array, also where no base
    is left behind the variant (one-row alignments)"""
    ref = "ACGTTGCAACGTACGGTCAT"        # 20 bases
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    scs = [
        dict(ctg=0, beg=16, end=20, vars=[[(17, D, "C", "", 9.0)], [], [(17, D, "C", "", 9.0)], []]),     # end = contig length
        dict(ctg=0, beg=17, end=21, vars=[[(18, S, "A", "G", 9.0)], [], [(18, S, "A", "G", 9.0)], [(18, S, "A", "C", 5.0)]]),
        dict(ctg=0, beg=15, end=20, vars=[[(16, D, "TC", "", 9.0)], [(17, I, "", "GG", 4.0)], [(16, D, "TC", "", 9.0)], []]),
        dict(ctg=0, beg=18, end=21, vars=[[(19, D, "T", "", 9.0)], [], [(19, D, "T", "", 9.0)], []]),     # nothing behind the variant
        dict(ctg=0, beg=18, end=21, vars=[[(19, S, "T", "A", 9.0)], [], [], [(19, S, "T", "A", 9.0)]]),
        dict(ctg=0, beg=18, end=20, vars=[[(19, I, "", "AC", 9.0)], [(19, I, "", "AC", 9.0)], [(19, I, "", "AC", 9.0)], []]),
    ]
    v = A.Variants.from_sites([ref], scs)
    batch = api.batch_from_variants(v)
    assert batch.lens(0)[4] == 4 and batch.lens(3)[4] == 2
    got, want, _, _ = compare(batch)
    assert want.aln_dist[0] == 0 and want.aln_dist[12] == 0 and batch.lens(3)[2] == 1


def test_sv_sized_sections_use_deferred_edit_distance():
    """Large indels make sync sections whose ref/truth segments both exceed the inline limit,
    so K4 (k_ed) computes wf_ed for them."""
    rng = np.random.RandomState(3)
    L = 1500
    ref = "".join(rng.choice(list("ACGT"), L))
    ins = "".join(rng.choice(list("ACGT"), 300))
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    # truth: a 200-base deletion with a 300-base insertion right behind it (ONE sync section whose reference and truth segments
    # are both longer than the inline limit); query: a slightly different 190-base deletion + 280 bases of the insertion
    t = [(400, D, ref[400:600], "", 40.0), (600, I, "", ins, 40.0)]
    q = [(405, D, ref[405:595], "", 30.0), (600, I, "", ins[:280], 30.0)]
    v = A.Variants.from_sites([ref], [dict(ctg=0, beg=300, end=1100, vars=[q, q, t, t])])
    batch = api.batch_from_variants(v)
    got, want, ntie, pr = compare(batch)
    names = {s_.kernel.decode() for s_ in pr.launch_stats()}
    assert any(n.startswith("k_ed") for n in names), names       # the deferred wf_ed kernel ran
    assert (got.ref_ed[2][0] > 32).any()


def test_deferred_edit_distance_bit_parallel_groups_and_other_bytes(monkeypatch):
    """k_ed_bits (pr_ed.hip): a sync section whose reference and truth segments are both longer than 4 096 bases -- the pattern
    takes two groups of 64 blocks, the carries between them cross LDS -- with bytes other than the four bases on both sides
    (N, a lower-case base: masks built on the spot).  Against the oracle, and equal to the wavefront kernel (VPR_ED_WF)."""
    rng = np.random.RandomState(11)
    L = 11500
    ref = list(rng.choice(list("ACGT"), L))
    for i in rng.choice(L, 40, replace=False):
        ref[i] = "N"
    ref = "".join(ref)
    ins = list(rng.choice(list("ACGT"), 4600))
    for i in rng.choice(4600, 25, replace=False):
        ins[i] = "N"
    ins[100] = "a"
    ins = "".join(ins)
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    t = [(400, D, ref[400:4900], "", 40.0), (4900, I, "", ins, 40.0)]
    q = [(405, D, ref[405:4895], "", 30.0), (4900, I, "", ins[:4580], 30.0)]
    v = A.Variants.from_sites([ref], [dict(ctg=0, beg=300, end=5400, vars=[q, q, t, t])])
    batch = api.batch_from_variants(v)
    got, want, _, pr = compare(batch)
    names = {s_.kernel.decode() for s_ in pr.launch_stats()}
    assert "k_ed_bits" in names and int(got.ref_ed[2][0].max()) > 2000
    monkeypatch.setenv("VPR_ED_WF", "1")
    pr2 = api.PrecisionRecall()
    other = pr2.run(batch)
    monkeypatch.delenv("VPR_ED_WF")
    assert not got.diff(other) and "k_ed_wf" in {s_.kernel.decode() for s_ in pr2.launch_stats()}


def test_workspace_chunking_gives_identical_results():
    batch = api.Synth(n_sc=120, len_a=50, len_b=800, len_max=800, seed=12).batch()
    full = api.PrecisionRecall().run(batch)
    small = api.PrecisionRecall(A.default_config(workspace_bytes=8 << 20)).run(batch)
    assert not full.diff(small)
    dense = api.PrecisionRecall(A.default_config(band_mode=0, workspace_bytes=8 << 20)).run(batch)
    assert not full.diff(dense)


def test_results_independent_of_batch_order():
    batch = api.Synth(n_sc=200, len_a=10, len_b=400, len_max=400, seed=13).batch()
    perm = np.random.RandomState(1).permutation(batch.n_sc)
    a = api.PrecisionRecall().run(batch)
    b = api.PrecisionRecall().run(batch.subset(perm))
    assert np.array_equal(a.aln_dist.reshape(-1, 4)[perm].ravel(), b.aln_dist)
    assert np.array_equal(a.sc_phase[perm], b.sc_phase)


def test_repeat_execute_is_idempotent():
    batch = api.Synth(n_sc=100, len_a=10, len_b=300, len_max=300, seed=14).batch()
    pr = api.PrecisionRecall()
    pr.upload(batch)
    pr.execute()
    a = pr.download()
    pr.execute()
    b = pr.download()
    assert not a.diff(b)


def test_walk_matches_oracle_path():
    batch = api.Synth(n_sc=1, len_mode=2, len_a=120.0, len_min=120, len_max=120, seed=21, var_per_base=0.05).batch()
    pr = api.PrecisionRecall(A.default_config(flags=A.CFG_KEEP_PATHS))    # (the zero-distance lane kernel keeps 16-byte path entries on request)
    pr.run(batch)
    for aln in range(4):
        ex = O.Extra(batch, want=(0, aln))
        O.run(batch, extra=ex)
        pl, q, t, sy, ed = pr.path(0, aln)
        opl, oq, ot, osy, oed = ex.path_arrays()
        assert np.array_equal(pl, opl) and np.array_equal(q, oq) and np.array_equal(t, ot)
        assert np.array_equal(sy, osy[:len(sy)]) and np.array_equal(ed, oed[:len(ed)])


def test_full_size_properties_identical_haps():
    """Size-independent property at a size the oracle would take minutes for: truth == query,
    all homozygous => every distance 0 and every variant TP with credit exactly 1.0."""
    syn = api.Synth(n_sc=20000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=6000, seed=31,
                    p_keep=1.0, p_drop=0.0, p_hom=1.0)
    batch = syn.batch(copy=False)
    r = api.PrecisionRecall().run(batch)
    assert (r.aln_dist == 0).all() and (r.sc_phase == A.PHASE_NONE).all()
    for h in range(4):
        for w in range(2):
            assert (r.errtype[h][w] == A.ERRTYPE_TP).all() and (r.credit[h][w] == 1.0).all()
            assert (r.query_ed[h][w] == 0).all() and (r.ref_ed[h][w] >= 1).all()


def test_full_size_property_swapped_haps_swap_phase():
    """Swapping the two query haps swaps (orig, swap) phase distances and exchanges the swap slots."""
    syn = api.Synth(n_sc=5000, len_mode=1, len_a=30.0, len_b=1.0, len_min=4, len_max=2000, seed=32, p_hom=0.3)
    b = syn.batch()
    swapped = A.Batch(b.n_sc, [b.hap_off[1], b.hap_off[0], b.hap_off[2], b.hap_off[3]],
                      [b.hap_seq[1], b.hap_seq[0], b.hap_seq[2], b.hap_seq[3]],
                      [b.hap_ptr[1], b.hap_ptr[0], b.hap_ptr[2], b.hap_ptr[3]],
                      [b.hap_flag[1], b.hap_flag[0], b.hap_flag[2], b.hap_flag[3]],
                      b.ref_off, b.ref_seq, [b.ref_ptr[1], b.ref_ptr[0]], [b.ref_flag[1], b.ref_flag[0]],
                      [b.var_off[1], b.var_off[0], b.var_off[2], b.var_off[3]],
                      [b.var_pos[1], b.var_pos[0], b.var_pos[2], b.var_pos[3]],
                      [b.var_qual[1], b.var_qual[0], b.var_qual[2], b.var_qual[3]])
    r1 = api.PrecisionRecall().run(b)
    r2 = api.PrecisionRecall().run(swapped)
    assert np.array_equal(r1.orig_phase_dist, r2.swap_phase_dist)
    assert np.array_equal(r1.swap_phase_dist, r2.orig_phase_dist)
    d1 = r1.aln_dist.reshape(-1, 4)
    d2 = r2.aln_dist.reshape(-1, 4)
    assert np.array_equal(d1[:, [2, 3, 0, 1]], d2)
    # truth hap 1 results under swap slot w of run 1 == under slot 1-w of run 2
    for w in range(2):
        assert np.array_equal(r1.errtype[2][w], r2.errtype[2][1 - w])


@pytest.mark.parametrize("fixture", ["regression_seed7.npz", "regression_joint61.npz"])
def test_gpu_matches_committed_regression_fixture(fixture):
    """GPU results against tests/golden/regression_seed7.npz / regression_joint61.npz (oracle-generated, committed; the second is the
    joint SNP + INDEL + SV shape of BASELINE configs[3])."""
    import importlib.util
    import os
    gd = os.path.join(os.path.dirname(__file__), "golden")
    spec = importlib.util.spec_from_file_location("make_regression", os.path.join(gd, "make_regression.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    g = np.load(os.path.join(gd, fixture))
    batch = api.Synth(**mod.FIXTURES[fixture]).batch()
    r = api.PrecisionRecall().run(batch)
    assert np.array_equal(r.aln_dist, g["aln_dist"]) and np.array_equal(r.aln_end_plane, g["aln_end_plane"])
    assert np.array_equal(r.sc_phase, g["sc_phase"])
    for h in range(4):
        for w in range(2):
            for name, dt in r.PER_VAR:
                x, y = getattr(r, name)[h][w], g[f"{name}_{h}_{w}"]
                if dt == np.float32:
                    x, y = x.view(np.uint32), y.view(np.uint32)
                assert np.array_equal(x, y), (name, h, w)


def test_recluster_supercluster_end_to_end():
    """rows (f1) -> (a): variants -> gap clustering -> superclustering (include/vcfdist_cluster.h) -> Level B ->
    host marshalling -> HIP path, against the oracle on the same re-derived superclusters.  The synthetic spans
    abut, so a 40-base gap chains many of them and the 400-base limit forces supercluster splits."""
    from vcfdist_amd import cluster as K
    v = api.Synth(n_sc=500, len_mode=1, len_a=30.0, len_b=0.8, len_min=8, len_max=400, seed=11).variants()
    assert len(v.ctg_off) == 2
    haps = [K.Hap(v.var_pos[i], v.var_ref_len[i], v.var_type[i], v.var_ref_len[i], v.var_alt_len[i]) for i in range(4)]
    cl = [K.simple_cluster(h, 0, 40, 0) for h in haps]
    s = K.supercluster(haps, cl, 400)
    o = K.supercluster(haps, cl, 400, L=O.lib(), prefix="vco")
    assert s == o and s.n > 20 and s.n_oversize > 0
    v2 = A.Variants(v.ctg_off, v.ctg_seq, np.zeros(s.n, np.int32), s.beg, s.end, [s.var_off(i) for i in range(4)],
                    v.var_pos, v.var_type, v.var_qual, v.var_ref_off, v.var_ref_len, v.var_alt_off, v.var_alt_len,
                    v.allele_pool)
    batch = api.batch_from_variants(v2)
    assert batch.n_sc == s.n
    got, want, ntie, pr = compare(batch)
    print(f"{s.n} superclusters from 500 spans ({s.n_oversize} oversize), max span {int((s.end - s.beg).max())}, "
          f"{ntie} alignments decided by the container order")


def test_zero_distance_level_mostly_rejects():
    """A discordant callset (truth keeps only 40 % of the query's sites): most alignments have s > 0, so the
    zero-distance sweep rejects more than the in-place round's cap (a quarter of the part) and the overflow goes to
    the retry ladder; everything still has to come out bit-exact."""
    batch = api.Synth(n_sc=6000, len_mode=1, len_a=25.0, len_b=0.9, len_min=4, len_max=300, seed=21,
                      p_keep=0.4, p_drop=0.3, var_per_base=0.05).batch()
    got, want, ntie, pr = compare(batch)
    t = pr.timing()
    frac_zero = float((want.aln_dist == 0).mean())
    print(f"{batch.n_sc} sc, s=0 for {frac_zero:.2f} of the alignments, {t.n_band_retries} retries, {ntie} alignments decided by the container order")
    assert frac_zero < 0.75 and t.n_band_retries > 6000      # > 25 % rejected: the in-place cap overflows


def test_biwfa_cluster_supercluster_end_to_end():
    """rows (f2) -> (f1) -> (a): variants -> biWFA clustering on the GPU -> superclustering -> Level B -> marshalling ->
    HIP path, every stage against its oracle."""
    from vcfdist_amd import cluster as K
    v = api.Synth(n_sc=300, len_mode=1, len_a=30.0, len_b=0.8, len_min=8, len_max=300, seed=13, p_repeat=0.5).variants()
    ctg = bytes(v.ctg_seq)
    haps, cl = [], []
    for i in range(4):
        pool = v.allele_pool[i]
        refs = [bytes(pool[o:o + n]) for o, n in zip(v.var_ref_off[i], v.var_ref_len[i])]
        alts = [bytes(pool[o:o + n]) for o, n in zip(v.var_alt_off[i], v.var_alt_len[i])]
        h = K.HapSeq(v.var_pos[i], v.var_type[i], refs, alts)
        got, sg = K.wfa_cluster(h, ctg)
        want, so = K.wfa_cluster(h, ctg, L=O.lib(), prefix="vco")
        assert got == want, i
        haps.append(h); cl.append(got)
    s = K.supercluster(haps, cl, 2000)
    assert s == K.supercluster(haps, cl, 2000, L=O.lib(), prefix="vco") and s.n > 5
    v2 = A.Variants(v.ctg_off, v.ctg_seq, np.zeros(s.n, np.int32), s.beg, s.end, [s.var_off(i) for i in range(4)],
                    v.var_pos, v.var_type, v.var_qual, v.var_ref_off, v.var_ref_len, v.var_alt_off, v.var_alt_len,
                    v.allele_pool)
    batch = api.batch_from_variants(v2)
    got, want, ntie, pr = compare(batch)
    print(f"{sum(c.n for c in cl)} biWFA clusters -> {s.n} superclusters (largest {int((s.end - s.beg).max())}), {ntie} alignments decided by the container order")


def test_fuzz_smoke():
    """a few seconds of tests/fuzz_parity.py (random workload shapes, fixed seeds) inside the suite; the long runs are
    done by hand: `python tests/fuzz_parity.py 900`"""
    import fuzz_parity
    runs, scs, _ = fuzz_parity.fuzz(8.0, 777000, verbose=False, max_batches=12)
    assert runs >= 3 and scs > 1000


def _adjacent_deletions(k):
    """one supercluster: k directly adjacent one-base deletions (separate records) on QUERY hap 1, nothing elsewhere"""
    rng = np.random.default_rng(5)
    ctg = rng.choice(np.frombuffer(b"ACGT", np.uint8), size=60).astype(np.uint8)
    pos = np.arange(20, 20 + k, dtype=np.int32)
    z32, z64, zu8, zf = np.zeros(0, np.int32), np.zeros(0, np.int64), np.zeros(0, np.uint8), np.zeros(0, np.float32)
    return A.Variants(np.array([0, 60], np.int64), ctg, np.zeros(1, np.int32), np.array([19], np.int32),
                      np.array([20 + k + 1], np.int32), [np.array([0, k], np.int64)] + [np.array([0, 0], np.int64)] * 3,
                      [pos, z32, z32, z32], [np.full(k, 3, np.uint8), zu8, zu8, zu8], [np.full(k, 30.0, np.float32), zf, zf, zf],
                      [np.arange(k, dtype=np.int64), z64, z64, z64], [np.ones(k, np.int32), z32, z32, z32],
                      [np.full(k, k, np.int64), z64, z64, z64], [np.zeros(k, np.int32), z32, z32, z32],
                      [ctg[20:20 + k].copy()] + [np.zeros(1, np.uint8)] * 3)


def test_adjacent_deletion_records_up_to_the_swap_source_limit():
    """Directly adjacent deletion records all point at the hap base in front of the first one; each contributes one
    allowed swap source to the position behind them.  The library keeps eight sources per position (two int4 lists, a
    3-bit rank in the flag byte): up to seven adjacent records match the oracle, eight are refused by vpr_upload with a
    message -- a documented limit, not a wrong result."""
    for k in (3, 4, 5, 7):
        compare(api.batch_from_variants(_adjacent_deletions(k)))
    r = api.PrecisionRecall().run(api.batch_from_variants(_adjacent_deletions(8)))
    assert ((r.aln_status & A.ST_ERR_LIMIT) != 0).all() and (r.errtype[0][0] == A.ERRTYPE_UN).all()


def test_a_supercluster_beyond_the_swap_source_limit_does_not_touch_the_rest_of_the_batch():
    """nine directly adjacent deletion records in ONE supercluster of a batch (the reference has no limit on swap sources per
    position, dist.cpp:335-350; the library keeps eight): that supercluster's four alignments come back with VPR_ST_ERR_LIMIT
    and unevaluated variants, every other supercluster equals the oracle -- vpr_upload used to refuse the whole batch"""
    rng = np.random.RandomState(23)
    ref = "".join(rng.choice(list("ACGT"), 400))
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    other = lambda c: "ACGT"[("ACGT".index(c) + 1) % 4]
    run9 = [(200 + k, D, ref[200 + k], "", 30.0) for k in range(9)]
    scs = [dict(ctg=0, beg=40, end=70, vars=[[(50, S, ref[50], other(ref[50]), 20.0)], [], [(50, S, ref[50], other(ref[50]), 40.0)], [(60, I, "", "GT", 9.0)]]),
           dict(ctg=0, beg=190, end=220, vars=[run9, [], [(200, D, ref[200:209], "", 50.0)], []]),
           dict(ctg=0, beg=300, end=340, vars=[[(310, D, ref[310:313], "", 20.0)], [(320, S, ref[320], other(ref[320]), 5.0)],
                                               [(310, D, ref[310:313], "", 40.0)], [(320, S, ref[320], other(ref[320]), 7.0)]])]
    v = A.Variants.from_sites([ref], scs)
    batch = api.batch_from_variants(v)
    got = api.PrecisionRecall().run(batch)
    want = O.run(batch)
    want = want[0] if isinstance(want, tuple) else want
    st = got.aln_status.reshape(-1, 4)
    assert ((st[1] & A.ST_ERR_LIMIT) != 0).all() and not (st[[0, 2]] & A.ST_ERR_LIMIT).any()
    for name in ("aln_dist", "aln_end_plane", "aln_beg_plane", "aln_status"):
        assert np.array_equal(getattr(got, name).reshape(-1, 4)[[0, 2]], getattr(want, name).reshape(-1, 4)[[0, 2]]), name
    for h in range(4):
        keep = np.ones(int(batch.var_off[h][-1]), bool)
        keep[batch.var_off[h][1]:batch.var_off[h][2]] = False
        for w in range(2):
            assert (got.errtype[h][w][~keep] == A.ERRTYPE_UN).all()
            for name, dt in A.Results.PER_VAR:
                x, y = getattr(got, name)[h][w][keep], getattr(want, name)[h][w][keep]
                assert np.array_equal(x.view(np.uint32) if dt == np.float32 else x, y.view(np.uint32) if dt == np.float32 else y), (name, h, w)


@pytest.mark.parametrize("band_mode", [1, 3, 2, 0])
def test_runs_of_adjacent_deletion_records(band_mode):
    """five to eight swap sources on one position (second candidate list, third rank bit in F_SUB's place), with and
    without the same records on the truth side, at every window mode"""
    import indel_runs
    for seed in (11, 12):
        compare(api.batch_from_variants(indel_runs.indel_run_superclusters(seed)), A.default_config(band_mode=band_mode))


@pytest.mark.parametrize("kw", [
    dict(n_sc=500, len_a=6, len_b=120, len_min=5, len_max=120, seed=71, var_per_base=0.08, p_snp=0.4, p_repeat=0.5),
    dict(n_sc=40, len_a=300, len_b=3000, len_min=300, len_max=3000, seed=72, var_per_base=0.02, p_snp=0.5, indel_mean=20.0),
    dict(n_sc=3000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=3000, seed=73),
])
def test_device_generate_ptrs_strs_equals_host_marshalling(kw):
    """vpr_upload_variants: the device writes the haplotype strings, the reference string and the pointer / flag arrays
    (generate_ptrs_strs, dist.cpp:145-242; pr_gen.hip) from the variant tables.  Every array must equal the host
    marshalling's (itself compared with the oracle's restatement in tests/test_host.py), and the path's results the
    oracle's."""
    syn = api.Synth(**kw)
    host = syn.batch()
    pr = api.PrecisionRecall()
    pr.upload_variants(syn.struct, host)
    dev = pr.download_level_a(host)
    for f in ("hap_off", "hap_seq", "hap_ptr", "hap_flag"):
        for h in range(4):
            assert np.array_equal(getattr(dev, f)[h], getattr(host, f)[h]), (f, h)
    assert np.array_equal(dev.ref_off, host.ref_off) and np.array_equal(dev.ref_seq, host.ref_seq)
    for q in range(2):
        assert np.array_equal(dev.ref_ptr[q], host.ref_ptr[q]) and np.array_equal(dev.ref_flag[q], host.ref_flag[q])
    pr.execute()
    got = pr.download()
    want = O.run(host)
    want = want[0] if isinstance(want, tuple) else want
    assert not got.diff(want)


def test_device_generate_contig_end_and_bad_input():
    """the regions at a contig's end (cut at the last base) through the device generator, and input the sizing pass refuses"""
    ref = "ACGTTGCAACGTACGGTCAT"
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    scs = [dict(ctg=0, beg=16, end=20, vars=[[(17, D, "C", "", 9.0)], [], [(17, D, "C", "", 9.0)], []]),
           dict(ctg=0, beg=15, end=20, vars=[[(16, D, "TC", "", 9.0)], [(17, I, "", "GG", 4.0)], [(16, D, "TC", "", 9.0)], []]),
           dict(ctg=0, beg=18, end=20, vars=[[(19, I, "", "AC", 9.0)], [(19, I, "", "AC", 9.0)], [(19, I, "", "AC", 9.0)], []])]
    v = A.Variants.from_sites([ref], scs)
    host = api.batch_from_variants(v)
    pr = api.PrecisionRecall()
    pr.upload_variants(v.as_struct(), v)
    dev = pr.download_level_a(host)
    for h in range(4):
        assert np.array_equal(dev.hap_seq[h], host.hap_seq[h]) and np.array_equal(dev.hap_ptr[h], host.hap_ptr[h]) and np.array_equal(dev.hap_flag[h], host.hap_flag[h])
    pr.execute()
    want = O.run(host)
    want = want[0] if isinstance(want, tuple) else want
    assert not pr.download().diff(want)
    bad = A.Variants.from_sites([ref], [dict(ctg=0, beg=4, end=9, vars=[[(6, D, "CA", "", 5.0), (7, S, "A", "T", 5.0)], [], [], []])])
    with pytest.raises(api.VprError):
        api.PrecisionRecall().upload_variants(bad.as_struct(), bad)


@pytest.mark.gpu
def test_hand_derived_credit_and_sync_rules_through_the_hip_path():
    """the hand-derived cases of tests/test_oracle.py (partial credit, sync rules, cancelling truth variants) through the C ABI:
    the HIP result equals the hand-derived answer directly, not only the oracle's"""
    import test_oracle as TO
    for case in TO.HAND_CASES:
        batch = api.batch_from_variants(TO.hand_case_variants(case))
        TO.check_hand_case(case, api.PrecisionRecall().run(batch))



def test_distance_one_lane_level_finishes_most_rejects_and_equals_the_oracle(monkeypatch):
    """pr_d1.hip: alignments the zero-distance lane kernel rejects with a complete wave 0 go through the distance-1 lane kernel;
    on whole-genome-like input most of them have s = 1 and are finished there (every array still equal to the oracle's), and
    with the level switched off (VPR_NO_D1) the 16-cell kernels give the same arrays"""
    batch = api.Synth(n_sc=20000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=1000, seed=11).batch()
    monkeypatch.setenv("VPR_D1_MAX_ROWS", "1024")      # (default 256 rows: longer rejects stay with the 16-cell kernels; here all of them)
    got, want, ntie, pr = compare(batch)
    t = pr.timing()
    n1 = int((got.aln_dist == 1).sum())
    print(f"{t.n_lane1_seen} rejects with a complete wave 0, {t.n_lane1_finished} finished at the lane level, {n1} alignments with s = 1, "
          f"{t.n_lane1_waves_dropped} waves dropped")
    assert t.n_lane1_seen > 5000 and t.n_lane1_finished > 0.8 * t.n_lane1_seen and t.n_lane1_finished <= n1
    assert t.n_lane1_finished > 0.9 * n1 and t.n_lane1_waves_dropped == 0
    monkeypatch.delenv("VPR_D1_MAX_ROWS")
    pr1 = api.PrecisionRecall()
    assert not got.diff(pr1.run(batch)) and 0 < pr1.timing().n_lane1_seen < t.n_lane1_seen
    monkeypatch.setenv("VPR_NO_D1", "1")
    pr2 = api.PrecisionRecall()
    assert not got.diff(pr2.run(batch)) and pr2.timing().n_lane1_seen == 0


@pytest.mark.parametrize("kw", [
    dict(n_sc=600, len_a=20, len_b=300, len_min=20, len_max=300, seed=41, var_per_base=0.02, p_snp=0.3, indel_mean=3.0),      # INS / DEL steps
    dict(n_sc=600, len_a=20, len_b=300, len_min=20, len_max=300, seed=42, var_per_base=0.02, p_snp=0.9),                     # SUB steps
    dict(n_sc=600, len_a=20, len_b=200, len_min=20, len_max=200, seed=43, var_per_base=0.03, p_snp=0.4, p_repeat=0.7),       # repeats: several diagonals
])
def test_distance_one_walks_equal_the_oracle_paths(kw):
    """the walks of the distance-1 lane level, step for step (plane, position, truth row, sync and edit flags), incl. the extra
    step of an INS move"""
    batch = api.Synth(**kw).batch()
    pr = api.PrecisionRecall(A.default_config(flags=A.CFG_KEEP_PATHS))
    got = pr.run(batch)
    t = pr.timing()
    assert t.n_lane1_finished > 50
    n_checked = n_edit = 0
    for a in np.flatnonzero(got.aln_dist == 1)[:200]:
        sc, aln = int(a) // 4, int(a) % 4
        one = batch.subset(np.array([sc]))
        ex = O.Extra(one, want=(0, aln))
        O.run(one, extra=ex)
        pl, q, tt, sy, ed = pr.path(sc, aln)
        opl, oq, ot, osy, oed = ex.path_arrays()
        assert np.array_equal(pl, opl) and np.array_equal(q, oq) and np.array_equal(tt, ot), (sc, aln)
        assert np.array_equal(sy, osy[:len(sy)]) and np.array_equal(ed, oed[:len(ed)]), (sc, aln)
        n_checked += 1
        n_edit += int(ed.sum())
    assert n_checked >= 100 and n_edit == n_checked
