"""GPU parity tests beyond the WGS-like case: the container-order replay (pr_tie.hip) under every window mode and in its
second attempt, the walk (path / sync / edits of get_prec_recall_path_sync, dist.cpp:842-999) at every window level, and
the workloads of BASELINE.json configs 3 (`-l 10000 -s 10002`: SV-sized indels, long-sequence DP, wf_ed segments of
thousands of bases; globals.cpp:478-481, dist.cpp:1406-1506) and 5 (hap lengths log-uniform 32-16384), against the CPU
oracle where it finishes in seconds and through size-independent properties beyond that."""
import json
import os

import numpy as np
import pytest

import oracle_lib as O
from test_gpu_parity import compare
from vcfdist_amd import _abi as A
from vcfdist_amd import api

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


# ----------------------------------------------------------------------------------------------------------------------
# ties: the reference keeps the last writer of swap_pred (dist.cpp:347,376); repeat-rich, indel-heavy superclusters tie often
# ----------------------------------------------------------------------------------------------------------------------
TIE_RICH = dict(n_sc=1500, len_a=6, len_b=90, len_min=5, len_max=90, seed=41, var_per_base=0.12, p_snp=0.3, p_repeat=1.0,
                p_keep=0.6, p_drop=0.2)


@pytest.mark.parametrize("band_mode", [1, 3, 2, 0])
def test_container_order_decides_ties_in_every_window_mode(band_mode):
    batch = api.Synth(**TIE_RICH).batch()
    got, want, n_nonmax, pr = compare(batch, A.default_config(band_mode=band_mode))
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    print(f"band_mode {band_mode}: {n_tie} alignments with consulted ties, {n_nonmax} where the container order is not the "
          f"highest index, {pr.timing().n_tie_replays} replays, {pr.timing().ms_tie:.2f} ms")
    assert n_nonmax > 20          # the workload really depends on the replay


def test_tie_replay_second_attempt_with_worst_case_logs():
    """32-entry FIFO logs overflow on almost every replay; the alignments stay marked and the final pass repeats them with
    logs sized for the worst case"""
    batch = api.Synth(**dict(TIE_RICH, n_sc=600, seed=42)).batch()
    got, want, n_nonmax, pr = compare(batch, A.default_config(flags=A.CFG_TIE_SMALL_LOGS))
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    assert n_nonmax > 5 and pr.timing().n_tie_replays > n_tie     # (> : second attempts are counted)


def test_many_ties_met_by_a_retry_round():
    """dense variants in repeats, short haps: most alignments are retried at 64 cells and a third of those meet a tie there,
    i.e. one retry round hands hundreds of alignments to a tie round (tests/fuzz_chain.py seed 9362: the retry ladders'
    regions of the tie list buffer overlapped its end, and whatever was allocated behind it lost its contents)"""
    for seed in (9362, 9363):
        syn = api.Synth(n_sc=320, len_mode=1, len_a=200.0, len_b=0.4, len_min=8, len_max=583, seed=seed, p_repeat=0.9,
                        var_per_base=0.05, p_snp=0.30, indel_mean=4.1)
        got, want, n_nonmax, pr = compare(syn.batch())
        t = pr.timing()
        print(f"seed {seed}: {t.n_band_retries} retries, {t.n_tie_replays} tie replays, {n_nonmax} decided other than by the largest source")
        assert t.n_tie_replays > 250 and t.n_band_retries > 1500


def test_ties_in_every_chunk_of_a_many_chunk_plan():
    """a tie-rich batch through an 8 MB workspace (the smallest the library takes): the round-0 plan is cut into dozens of chunks and each of them hands
    alignments to early tie replays, every one of which takes a decision list -- the lists are recycled per chunk (they
    used to be handed out once per vpr_execute: 64 of them, then VPR_ERR_STATE "out of decision lists")"""
    batch = api.Synth(**dict(TIE_RICH, n_sc=18000, seed=43)).batch()
    got, want, n_nonmax, pr = compare(batch, A.default_config(workspace_bytes=8 << 20))
    n_main = sum(1 for s_ in pr.launch_stats() if s_.kernel.decode() == "k_zero_lane" and s_.n_units > 100)
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    print(f"{n_main} chunks, {n_tie} tied alignments, {pr.timing().n_tie_replays} replays")
    assert n_main > 20 and n_tie > 200


def test_guarded_allocations_no_access_behind_an_array():
    """every device array in an allocation of its own (VPR_CFG_GUARD_ALLOC): a write or read far behind an array is a GPU
    fault here, not a silent hit on whatever the pooled allocator placed there.  Tie-rich retries, long alignments, a
    WGS-like mix."""
    cfg = A.default_config(flags=A.CFG_GUARD_ALLOC)
    for kw in (dict(n_sc=320, len_mode=1, len_a=200.0, len_b=0.4, len_min=8, len_max=583, seed=9364, p_repeat=0.9, var_per_base=0.05,
                    p_snp=0.30, indel_mean=4.1),
               dict(n_sc=60, len_a=520, len_b=1400, len_min=520, len_max=1400, seed=44, var_per_base=0.03, p_snp=0.3, p_repeat=1.0,
                    indel_mean=6.0, p_keep=0.7, p_drop=0.15),
               dict(n_sc=3000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000, seed=11)):
        batch = api.Synth(**kw).batch()
        os.environ["VPR_DEBUG"] = "1"       # (the rounds of the run on stderr: shown when the comparison fails)
        try:
            compare(batch, cfg)
        except AssertionError:
            # what a mismatch here depends on: the same handle again, fresh handles with and without the guard
            want = O.run(batch)
            for flags in (A.CFG_GUARD_ALLOC, 0):
                for attempt in range(3):
                    pr = api.PrecisionRecall(A.default_config(flags=flags))
                    d1 = pr.run(batch).diff(want)
                    pr.execute()
                    d2 = pr.download().diff(want)
                    print(f"diagnosis: flags {flags} fresh handle {attempt}: {d1[:3]} / executed again: {d2[:3]}")
            raise
        finally:
            del os.environ["VPR_DEBUG"]


def test_ties_in_long_alignments_and_retried_ones():
    """long alignments (one wave each, speculative replays behind the forward sweep) and alignments the retry ladders
    accepted at a wider window (their ties are collected behind the ladder round's backward sweep)"""
    batch = api.Synth(n_sc=160, len_a=520, len_b=1400, len_min=520, len_max=1400, seed=43, var_per_base=0.03, p_snp=0.3,
                      p_repeat=1.0, indel_mean=6.0, p_keep=0.7, p_drop=0.15).batch()
    got, want, n_nonmax, pr = compare(batch)
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    print(f"{n_tie} alignments with consulted ties, {n_nonmax} decided by the container order, {pr.timing().n_band_retries} retries")
    assert n_tie > 10 and pr.timing().n_band_retries > 0


def test_tie_round_of_the_long_part_copies_the_saved_forward_flags():
    """alignments of 2 048+ rows start at the 64-cell level with one wavefront each; their forward sweep keeps a second copy
    of the flags (k_fwd_stripe_save) and the tie round of the marked ones copies it back (k_restore_stripe) instead of
    repeating the sweep.  Same arrays as the oracle, and as the library with the copy switched off."""
    batch = api.Synth(n_sc=16, len_a=2100, len_b=3200, len_min=2100, len_max=3200, seed=47, var_per_base=0.03, p_snp=0.3,
                      p_repeat=1.0, indel_mean=6.0, p_keep=0.7, p_drop=0.15).batch()
    got, want, n_nonmax, pr = compare(batch)
    names = [s_.kernel.decode() for s_ in pr.launch_stats()]
    n_tie = int(((want.aln_status & A.ST_SWAP_TIE) != 0).sum())
    print(f"{n_tie} alignments with consulted ties, {n_nonmax} decided by the container order; kernels: {sorted(set(names))}")
    assert n_tie > 0 and "k_restore_stripe" in names
    os.environ["VPR_NO_FLAG_SAVE"] = "1"
    try:
        pr2 = api.PrecisionRecall()
        got2 = pr2.run(batch)
    finally:
        del os.environ["VPR_NO_FLAG_SAVE"]
    assert "k_restore_stripe" not in [s_.kernel.decode() for s_ in pr2.launch_stats()]
    assert not got2.diff(got)


# ----------------------------------------------------------------------------------------------------------------------
# the walk at every window level
# ----------------------------------------------------------------------------------------------------------------------
def _check_paths(batch, cfg, pairs):
    cfg.flags |= A.CFG_KEEP_PATHS       # (the zero-distance lane kernel writes 16-byte path entries on request only)
    pr = api.PrecisionRecall(cfg)
    pr.run(batch)
    n_checked = 0
    for sc, aln in pairs:
        try:
            pl, q, t, sy, ed = pr.path(sc, aln)
        except api.VprError:      # the retry ladder reused the workspace of the round that accepted this alignment
            continue
        ex = O.Extra(batch, want=(sc, aln))
        O.run(batch, extra=ex)
        opl, oq, ot, osy, oed = ex.path_arrays()
        assert np.array_equal(pl, opl) and np.array_equal(q, oq) and np.array_equal(t, ot), (sc, aln)
        assert np.array_equal(sy, osy[:len(sy)]) and np.array_equal(ed, oed[:len(ed)]), (sc, aln)
        n_checked += 1
    assert n_checked >= max(1, len(pairs) // 2)
    return pr


@pytest.mark.parametrize("band_mode", [1, 3, 2, 0])
def test_walk_matches_oracle_path_at_every_start_level(band_mode):
    """zero-distance / 16-cell (k_walk_q16), 64-cell (k_walk_rows, k_walk<lane>) and dense (k_walk<lane>) walks"""
    batch = api.Synth(n_sc=3, len_mode=2, len_a=150.0, len_min=150, len_max=150, seed=51, var_per_base=0.05, p_keep=0.7).batch()
    _check_paths(batch, A.default_config(band_mode=band_mode), [(s, a) for s in range(3) for a in range(4)])


@pytest.mark.parametrize("seed,indel_mean", [(52, 3.0), (53, 14.0), (54, 60.0), (55, 200.0), (56, 600.0)])
def test_walk_matches_oracle_path_long_and_wide_levels(seed, indel_mean):
    """long alignments (row-sweep walk of the 64-cell layout) and alignments that only the 256 / 1024-cell windows or the
    dense level accept (k_walk<wave>, k_walk<lane> on window rows): one supercluster per case, truth missing most of the
    query's indels, whose size decides the window that is needed.  (The walks of the round that finished last in its
    ladder are checked; an earlier round's workspace has been reused.)"""
    batch = api.Synth(n_sc=1, len_mode=2, len_a=1800.0, len_min=1800, len_max=1800, seed=seed, var_per_base=0.01, p_snp=0.3,
                      indel_mean=indel_mean, p_keep=0.3, p_drop=0.6, p_hom=1.0).batch()
    pr = _check_paths(batch, A.default_config(), [(0, a) for a in range(4)])
    print(f"indel mean {indel_mean}: s = {pr.download().aln_dist.tolist()}, {pr.timing().n_band_retries} retries")


# ----------------------------------------------------------------------------------------------------------------------
# the reference-produced toy vector of SURVEY.md A.1 through the HIP path
# ----------------------------------------------------------------------------------------------------------------------
def test_reference_toy_vector_through_the_hip_path():
    g = json.load(open(os.path.join(HERE, "golden", "toy_a1.json")))
    T = {"SUB": A.TYPE_SUB, "INS": A.TYPE_INS, "DEL": A.TYPE_DEL}
    qv = [(p, T[t], r, a, 30.0) for p, t, r, a in g["query_variants"]]
    tv = [(p, T[t], r, a, 30.0) for p, t, r, a in g["truth_variants"]]
    v = A.Variants.from_sites([g["ref"]], [dict(ctg=0, beg=g["region"][0], end=g["region"][1], vars=[qv, qv, tv, tv])])
    b = api.batch_from_variants(v)          # the library's own generate_ptrs_strs (host C++)
    assert bytes(b.hap_seq[0]).decode() == g["query_str"]
    assert b.hap_ptr[0].tolist() == g["q2r_ptrs"] and b.hap_flag[0].tolist() == g["q2r_flags"]
    assert b.ref_ptr[0].tolist() == g["r2q_ptrs"] and b.ref_flag[0].tolist() == g["r2q_flags"]
    for band_mode in (1, 0):
        r = api.PrecisionRecall(A.default_config(band_mode=band_mode)).run(b)
        assert r.aln_dist.tolist() == g["s"]
        assert [("QUERY", "REF")[e] for e in r.aln_end_plane] == g["end_plane"]


# ----------------------------------------------------------------------------------------------------------------------
# config 3: SV-sized variants, long-sequence DP
# ----------------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("seed", [64017, 64006])
def test_sv_sized_indels_long_alignments_against_the_oracle(seed):
    """the fuzzer's SV shape as fixed cases (picked with tools/scan_sv_seeds.py): 5-16 k base haplotypes, indels of hundreds
    to thousands of bases on one side only -- the 256 / 1024-cell windows, the dense level (seed 64017: Lq + Lr = 28.5 k, so
    its backward sweep takes the int16 score rows without being asked to), wf_ed on segments of more than a thousand bases"""
    import fuzz_parity
    shape, kw, band_mode = fuzz_parity.random_workload(seed, 6)
    batch = api.Synth(**kw).batch()
    got, want, n_nonmax, pr = compare(batch, A.default_config(band_mode=band_mode))
    t = pr.timing()
    big = max(max(batch.lens(k)[q] for q in (0, 1)) + batch.lens(k)[4] for k in range(batch.n_sc))
    print(f"seed {seed}: {batch.n_sc} superclusters, largest Lq + Lr {big}, {batch.dense_cells():.2e} dense cells, "
          f"{t.n_band_retries} retries, wf_ed {t.ms_ed:.2f} ms, largest ref_ed {int(max(got.ref_ed[2][0].max(), got.ref_ed[3][0].max()))}")
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert t.ms_ed > 0 and t.n_band_retries > 0 and any(n.startswith("k_fwd<") for n in names)
    assert "k_fwd_strip" in names and "k_bwd_strip" in names        # the wide alignments of the dense level: over column strips
    if seed == 64017:
        assert any("s16" in n for n in names)


def test_dense_strips_and_what_the_strip_planner_cannot_cut():
    """dense plan over 9 000-base haplotypes: the alignments of the query haplotype without the SV run as column strips
    (three workgroups each, pr_strip.hip); the other query haplotype carries a 5 000-base insertion, more columns than one
    strip holds and no column inside it where the planes map 1:1, so the planner leaves its two alignments to the
    one-workgroup kernels (skip lists).  Both against the oracle."""
    rng = np.random.RandomState(17)
    L = 9000
    ref = "".join(rng.choice(list("ACGT"), L))
    ins = "".join(rng.choice(list("ACGT"), 5000))
    S, I, D = A.TYPE_SUB, A.TYPE_INS, A.TYPE_DEL
    other = lambda c: "ACGT"[("ACGT".index(c) + 1) % 4]
    q1 = [(3000, I, "", ins, 30.0), (6000, S, ref[6000], other(ref[6000]), 30.0)]
    q2 = [(2000, D, ref[2000:2040], "", 20.0), (6000, S, ref[6000], other(ref[6000]), 30.0)]
    t1 = [(6000, S, ref[6000], other(ref[6000]), 40.0), (7000, S, ref[7000], other(ref[7000]), 40.0)]
    t2 = [(2000, D, ref[2000:2040], "", 40.0), (4500, I, "", ins[:90], 40.0)]
    v = A.Variants.from_sites([ref], [dict(ctg=0, beg=100, end=8900, vars=[q1, q2, t1, t2])])
    batch = api.batch_from_variants(v)
    got, want, _, pr = compare(batch, A.default_config(band_mode=0))
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert "k_fwd_strip" in names and "k_bwd_strip" in names and any(n.startswith("k_fwd<1024") for n in names)
    assert want.aln_dist[0] <= 2 and want.aln_dist[3] >= 90


def test_diagnostic_switches_leave_the_results_unchanged(monkeypatch):
    """the dense level in one workgroup instead of column strips, forward strips without the skipped blocks, one complete
    retry round at a time, edit distances by anti-diagonals: every array equal to the default run's (and that one to the
    oracle's)"""
    import fuzz_parity
    shape, kw, band_mode = fuzz_parity.random_workload(64006, 6)        # (one of the SV cases above: it reaches the strips)
    batch = api.Synth(**kw).batch()
    got, want, _, pr = compare(batch, A.default_config(band_mode=1))
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert "k_fwd_strip" in names
    for var in ("VPR_NO_STRIPS", "VPR_NO_UB", "VPR_NO_ROUND_OVERLAP", "VPR_ED_DIAG", "VPR_ED_WF"):
        monkeypatch.setenv(var, "1")
        other = api.PrecisionRecall(A.default_config(band_mode=1)).run(batch)
        monkeypatch.delenv(var)
        assert not got.diff(other), var


def test_segment_walk_and_row_walk_equal_the_oracle(monkeypatch):
    """long alignments at the 64-cell level (a fifth of them inside tandem repeats): the segment-parallel walk (default) and the
    sequential row-sweep walk (VPR_SEQ_WALK) both give the oracle's arrays"""
    batch = api.Synth(n_sc=24, len_mode=0, len_a=700.0, len_b=3000.0, len_min=700, len_max=3000, seed=415, p_repeat=0.3).batch()
    got, want, _, pr = compare(batch, A.default_config(band_mode=1))
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert "k_walk_seg" in names and "k_fwd_stripe" in names
    for var, kernel in (("VPR_SEQ_WALK", "k_walk_rows"),):
        monkeypatch.setenv(var, "1")
        pr2 = api.PrecisionRecall(A.default_config(band_mode=1))
        other = pr2.run(batch)
        monkeypatch.delenv(var)
        assert kernel in {s.kernel.decode() for s in pr2.launch_stats()}, var
        assert not got.diff(other), var


def test_long_part_threshold_leaves_the_results_unchanged(monkeypatch):
    """alignments of LONG_LT (1 024) truth rows or more start at the 64-cell level with a wavefront each, the shorter ones at the
    lane level; VPR_LONG_LT moves the border (diagnostic).  A batch with alignments on both sides of every border: the oracle's
    arrays at the default, the same arrays at 256 and at 2 048 (round 3's border)."""
    batch = api.Synth(n_sc=96, len_mode=0, len_a=150.0, len_b=2600.0, len_min=150, len_max=2600, seed=1024, p_repeat=0.2).batch()
    got, want, _, pr = compare(batch, A.default_config(band_mode=1))
    lt = np.diff(np.asarray(batch.hap_off[2][:97]))
    assert (lt < 256).any() and ((lt >= 256) & (lt < 1024)).any() and ((lt >= 1024) & (lt < 2048)).any() and (lt >= 2048).any()
    units = {}
    for v in ("256", "2048"):
        monkeypatch.setenv("VPR_LONG_LT", v)
        pr2 = api.PrecisionRecall(A.default_config(band_mode=1))
        other = pr2.run(batch)
        monkeypatch.delenv("VPR_LONG_LT")
        units[v] = max(s_.n_units for s_ in pr2.launch_stats() if s_.kernel.decode() == "k_zero_lane")
        assert not got.diff(other), v
    assert units["256"] < units["2048"]          # (the border really moved)


def test_memory_share_leaves_the_results_unchanged(monkeypatch):
    """a batch whose alignments meet container-order ties at every level (replays before and behind the repeated sweeps, lane
    levels, 16-cell round, 64-cell rounds): the oracle's arrays at the defaults, and the same arrays with half of the device set
    aside by the memory plan.  (Round 5's other switches -- replay job width, lane priority, occupancy cap, credit walks' stream,
    stream classes and padding -- are closed experiments: read only by a -DVPR_EXPERIMENTS build, pr_api.hip exp_getenv.)"""
    batch = api.Synth(n_sc=400, len_mode=0, len_a=40.0, len_b=2400.0, len_min=40, len_max=2400, seed=5150, p_repeat=0.35).batch()
    got, want, _, pr = compare(batch, A.default_config(band_mode=1))
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert any(n.startswith("k_tie_replay") for n in names) and "k_zero_lane" in names and "k_one_lane" in names
    for var, val in (("VPR_DEV_FREE_SHARE", "0.5"),):
        monkeypatch.setenv(var, val)
        other = api.PrecisionRecall(A.default_config(band_mode=1)).run(batch)
        monkeypatch.delenv(var)
        assert not got.diff(other), var


def test_memory_plan_leaves_its_share_of_the_device_free():
    """vpr_upload plans the batch's device memory: after an upload and two executes of a batch of long
    alignments -- ladders and replay scratches have grown to what they wanted -- at least VPR_DEV_FREE_SHARE of the device
    (11 %) is still free, and the results are the oracle's"""
    import ctypes
    batch = api.Synth(n_sc=300, len_mode=0, len_a=600.0, len_b=5000.0, len_min=600, len_max=5000, seed=77011, p_repeat=0.3).batch()
    got, want, _, pr = compare(batch, A.default_config(band_mode=1))
    pr.execute()
    hip = ctypes.CDLL("libamdhip64.so")          # (the runtime the library itself is linked against: already in the process)
    free, total = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free), ctypes.byref(total)) == 0
    assert free.value >= 0.105 * total.value, (free.value, total.value)


def test_device_reserve_turns_an_exhausted_device_into_an_error():
    """the library leaves VPR_DEV_RESERVE_MB of device memory to the runtime (kernels' private memory is allocated per queue behind
    its back, and a queue that cannot get it is aborted together with the process): an allocation that would go below the reserve
    fails like an exhausted device, and the caller sees VPR_ERR_NOMEM.  With the reserve set to all but 512 MB of the device (read
    once per process: a child process) nothing sizeable can be allocated -- an error comes back, not a crash; the default runs."""
    import ctypes
    import subprocess
    import sys
    api.PrecisionRecall().close()                       # (the HIP runtime is loaded and initialised by now)
    hip = ctypes.CDLL("libamdhip64.so")
    free_b, total_b = ctypes.c_size_t(0), ctypes.c_size_t(0)
    assert hip.hipMemGetInfo(ctypes.byref(free_b), ctypes.byref(total_b)) == 0
    total_mb = total_b.value >> 20
    code = ("from vcfdist_amd import api\n"
            "b = api.Synth(n_sc=2000, seed=3, len_a=10, len_b=300, len_max=300).batch()\n"
            "try:\n"
            "    r = api.PrecisionRecall().run(b)\n"
            "    print('ran', int(r.aln_dist.size))\n"
            "except api.VprError as e:\n"
            "    print('error', e)\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for reserve, want in ((str(total_mb - 512), "error"), (None, "ran")):
        env = dict(os.environ)
        env.pop("VPR_DEV_RESERVE_MB", None)
        if reserve:
            env["VPR_DEV_RESERVE_MB"] = reserve
        out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        last = [ln for ln in out.stdout.splitlines() if ln.startswith(("ran", "error"))]
        assert out.returncode == 0 and last and last[-1].startswith(want), (reserve, out.returncode, out.stdout[-400:], out.stderr[-400:])
        if want == "error":
            assert "(-3)" in last[-1], last[-1]          # VPR_ERR_NOMEM


def test_repeated_executes_of_one_batch_are_identical():
    """the planner's host loop serves fail lists and tie lists in the order the device raises them, so the launch order differs
    from execute to execute: every result array has to come out the same each time (60 executes of a batch with every level,
    retry rounds and a few thousand tie replays; a race between concurrent rounds showed up as 1 execute in 50 with
    VPR_ST_ERR_NO_PTR on a handful of alignments)"""
    syn = api.Synth(n_sc=120000, seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
    pr = api.PrecisionRecall()
    pr.upload(syn.batch(copy=False))
    pr.execute()
    first = pr.download()
    assert not (first.aln_status & (A.ST_ERR_NO_PTR | A.ST_ERR_LIMIT | A.ST_ERR_UNFINISHED)).any()
    for it in range(60):
        pr.execute()
        d = first.diff(pr.download())
        assert not d, (it, d[:3])


def test_identical_alignments_computed_once_equal_all_four_computed():
    """VPR_CFG_HAP_DEDUP: a callset that is homozygous over a supercluster has two identical haplotypes, and the alignments that
    differ only in which of them they name are computed once (k_hap_alias; not where a swap tie can occur: the reference's
    container order depends on the alignment's index there).  Same arrays as the default, which computes all four, and as the
    oracle, which always does."""
    syn = api.Synth(n_sc=4000, len_a=10, len_b=400, len_max=400, seed=314, var_per_base=0.02, p_hom=0.8, p_repeat=0.3)
    batch = syn.batch()
    got, want, _, pr = compare(batch, A.default_config(flags=A.CFG_HAP_DEDUP))
    n_run = pr.timing().n_alignments_computed
    assert 4000 < n_run < 4 * 4000 * 0.8, n_run          # a good part is aliased, not everything
    pr2 = api.PrecisionRecall()
    assert not got.diff(pr2.run(batch)) and pr2.timing().n_alignments_computed == 4 * 4000


def test_dense_backward_int16_rows_forced():
    batch = api.Synth(n_sc=30, len_a=30, len_b=900, len_min=30, len_max=900, seed=61, var_per_base=0.03).batch()
    compare(batch, A.default_config(band_mode=0, flags=A.CFG_DENSE_S16))


def test_sv_property_identical_callsets():
    """SV-sized variants carried by truth and query alike (the true positives of an SV evaluation), beyond what the oracle
    does in seconds: truth == query and all homozygous, so every distance is 0 and every variant a TP with credit 1.
    (A shared indel longer than the widest window still sends the alignment to the dense level -- the exit test is
    evaluated at the cell a swap edge leaves from, where the bound is 0.)"""
    syn = api.Synth(n_sc=24, len_mode=0, len_a=4000.0, len_b=13000.0, len_min=4000, len_max=13000, seed=62, var_per_base=0.0005,
                    p_snp=0.3, indel_mean=1500.0, p_keep=1.0, p_drop=0.0, p_hom=1.0)
    batch = syn.batch()
    pr = api.PrecisionRecall()
    r = pr.run(batch)
    assert (r.aln_dist == 0).all() and (r.sc_phase == A.PHASE_NONE).all() and not (r.aln_status & np.uint32(0xffffffff ^ A.ST_SWAP_TIE)).any()
    v = syn.variants()
    print(f"{batch.n_sc} superclusters, {batch.dense_cells():.2e} dense cells, kernels {pr.timing().ms_total:.1f} ms")
    for h in range(4):
        size = np.maximum(v.var_ref_len[h], v.var_alt_len[h])
        assert size.max() > 2000
        for w in range(2):
            assert (r.errtype[h][w] == A.ERRTYPE_TP).all() and (r.credit[h][w] == 1.0).all() and (r.query_ed[h][w] == 0).all()
            assert (r.ref_ed[h][w] >= 1).all()


def test_dense_level_beyond_one_workgroup():
    """haplotypes of 22-30 k bases at the dense level (band_mode 0): Lq + Lr is above what one workgroup's LDS rows hold
    (~40 k; an explicit error until round 3), the column strips take them.  The oracle needs minutes and 10 GB per such
    supercluster, so this is the identical-callsets property (every distance 0, every variant a TP with credit 1); the
    strips' parity with the oracle is tested at 9-16 k bases above."""
    syn = api.Synth(n_sc=2, len_mode=0, len_a=22000.0, len_b=30000.0, len_min=22000, len_max=30000, seed=63, var_per_base=0.0005,
                    p_snp=0.3, indel_mean=300.0, p_keep=1.0, p_drop=0.0, p_hom=1.0)
    batch = syn.batch()
    assert max(batch.lens(k)[0] + batch.lens(k)[4] for k in range(batch.n_sc)) > 44000
    pr = api.PrecisionRecall(A.default_config(band_mode=0))
    r = pr.run(batch)
    names = {s.kernel.decode() for s in pr.launch_stats()}
    assert "k_fwd_strip" in names and "k_bwd_strip" in names
    assert (r.aln_dist == 0).all() and (r.sc_phase == A.PHASE_NONE).all() and not (r.aln_status & np.uint32(0xffffffff ^ A.ST_SWAP_TIE)).any()
    for h in range(4):
        for w in range(2):
            assert (r.errtype[h][w] == A.ERRTYPE_TP).all() and (r.credit[h][w] == 1.0).all() and (r.query_ed[h][w] == 0).all()


# ----------------------------------------------------------------------------------------------------------------------
# config 5: hap lengths log-uniform on [32, 16384]
# ----------------------------------------------------------------------------------------------------------------------
STRESS = dict(len_mode=0, len_a=32.0, len_b=16384.0, len_min=32, len_max=16384)


def test_stress_workload_against_the_oracle():
    batch = api.Synth(n_sc=48, seed=0x5eed, **STRESS).batch()
    got, want, n_nonmax, pr = compare(batch)
    lens = np.array([batch.lens(k)[4] for k in range(batch.n_sc)])
    print(f"{batch.n_sc} superclusters, reference spans {lens.min()}..{lens.max()}, {batch.dense_cells():.2e} dense cells, "
          f"kernels {pr.timing().ms_total:.1f} ms")
    assert lens.max() > 8000 and lens.min() < 64


def test_stress_workload_full_size_properties():
    """a bench-sized slice of config 5 (beyond the oracle): results do not depend on the batch composition (a permuted batch
    gives the permuted results), repeated executes are identical, and swapping the query haps swaps the phase distances"""
    syn = api.Synth(n_sc=600, seed=0x5eed + 1, **STRESS)
    b = syn.batch()
    pr = api.PrecisionRecall()
    r1 = pr.run(b)
    perm = np.random.RandomState(3).permutation(b.n_sc)
    r2 = api.PrecisionRecall().run(b.subset(perm))
    assert np.array_equal(r1.aln_dist.reshape(-1, 4)[perm].ravel(), r2.aln_dist)
    assert np.array_equal(r1.aln_status.reshape(-1, 4)[perm].ravel(), r2.aln_status)
    assert np.array_equal(r1.sc_phase[perm], r2.sc_phase)
    for h in range(4):
        off = b.var_off[h]
        idx = np.concatenate([np.arange(off[s], off[s + 1]) for s in perm]) if len(perm) else np.zeros(0, np.int64)
        for w in range(2):
            assert np.array_equal(r1.errtype[h][w][idx], r2.errtype[h][w])
            assert np.array_equal(r1.credit[h][w][idx].view(np.uint32), r2.credit[h][w].view(np.uint32))
            assert np.array_equal(r1.sync_group[h][w][idx], r2.sync_group[h][w])
    pr.execute()
    assert not r1.diff(pr.download())


# ----------------------------------------------------------------------------------------------------------------------
# config 4 of BASELINE.json (configs[3]): whole-genome SNP + INDEL + SV joint evaluation -- the whole-genome length mix with SV-sized
# indels (50 b - 10 kb, `-l 10000 -s 10002`, globals.cpp:478-481) in a fraction of the superclusters: the lane levels, the window
# ladder, the dense strips and the deferred edit distances in ONE plan (dist.cpp:1738-1903 treats every supercluster alike)
# ----------------------------------------------------------------------------------------------------------------------
JOINT = dict(len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10002, sv_min=50, sv_max=10000)


def _summary_rows(pr, syn, batch, res):
    from vcfdist_amd import summary as S
    cls = syn.var_class()
    pb, _, _ = S.phase(res.sc_phase, np.ones(batch.n_sc, np.int32))
    got = S.pr_counts(pr, cls, pb)
    return cls, pb, got, S.pr_summary(got)


@pytest.mark.parametrize("seed,n_sc,p_sv", [(0x5eed, 400, 0.06), (0x5eed + 11, 1500, 0.012)])
def test_joint_small_variant_and_sv_workload_against_the_oracle(seed, n_sc, p_sv):
    import oracle_lib as O
    from vcfdist_amd import summary as S
    syn = api.Synth(n_sc=n_sc, seed=seed, p_sv=p_sv, **JOINT)
    batch = syn.batch()
    got, want, n_nonmax, pr = compare(batch)
    assert not (got.aln_status & (A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED | A.ST_ERR_LIMIT)).any()
    t = pr.timing()
    names = {s.kernel.decode() for s in pr.launch_stats()}
    hl = np.maximum.reduce([np.diff(batch.hap_off[h]) for h in range(4)])
    print(f"{batch.n_sc} superclusters, {int((hl > 1000).sum())} with a haplotype of 1000+ bases (largest {int(hl.max())}), "
          f"{batch.dense_cells():.2e} dense cells, largest distance {int(got.aln_dist.max())}, {t.n_band_retries} retries, "
          f"wf_ed {t.ms_ed:.2f} ms, kernels {t.ms_total:.1f} ms: {sorted(names)}")
    # one plan held the lane level, the window ladder and the dense level of wide alignments
    assert "k_zero_lane" in names and t.n_band_retries > 0 and got.aln_dist.max() >= 50
    assert any(n.startswith("k_fwd<") or n == "k_fwd_strip" for n in names)
    # the summary's SV row is populated and equals the oracle's tally of the oracle's results
    cls, pb, counts, rows = _summary_rows(pr, syn, batch, got)
    assert np.array_equal(counts, O.oracle_pr_counts(O.lib(), batch.var_off, want, cls, pb))
    sv_none = [r for r in rows if S.NAMES[r.vartype] == "SV" and not r.best][0]
    assert sv_none.truth_tp + sv_none.truth_fn > 0 and sv_none.query_tp + sv_none.query_fp > 0
    assert [r.key() for r in rows] == [r.key() for r in S.pr_summary(counts, L=O.lib(), prefix="vso")]


def test_joint_workload_from_variant_tables_on_the_device():
    """the same kind of batch uploaded as variant tables (vpr_upload_variants: generate_ptrs_strs on the device, pr_gen.hip) --
    SV-sized alleles of thousands of bases in the allele pool -- against the oracle on the host-marshalled arrays"""
    syn = api.Synth(n_sc=300, seed=0x5eed + 21, p_sv=0.08, **JOINT)
    batch = syn.batch()
    got, want, _, pr = compare(batch, variants_struct=syn.struct)
    assert got.aln_dist.max() >= 50 and not (got.aln_status & (A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED | A.ST_ERR_LIMIT)).any()


def test_joint_workload_full_size_properties():
    """a bench-sized joint batch (beyond the oracle): no alignment comes back unevaluated, a permuted batch gives the permuted
    results, repeated executes are identical, the SV row is populated"""
    from vcfdist_amd import summary as S
    syn = api.Synth(n_sc=120000, seed=0x5eed + 2, p_sv=0.0075, **JOINT)
    b = syn.batch()
    pr = api.PrecisionRecall()
    r1 = pr.run(b)
    assert not (r1.aln_status & (A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED | A.ST_ERR_LIMIT)).any()
    cls, pb, counts, rows = _summary_rows(pr, syn, b, r1)
    sv_none = [r for r in rows if S.NAMES[r.vartype] == "SV" and not r.best][0]
    assert sv_none.truth_tp > 0 and sv_none.truth_fn > 0 and sv_none.query_fp > 0
    print(f"SV row: truth_tp {sv_none.truth_tp} query_tp {sv_none.query_tp} truth_fn {sv_none.truth_fn} query_fp {sv_none.query_fp}; "
          f"largest distance {int(r1.aln_dist.max())}, kernels {pr.timing().ms_total:.1f} ms")
    perm = np.random.RandomState(4).permutation(b.n_sc)
    r2 = api.PrecisionRecall().run(b.subset(perm))
    assert np.array_equal(r1.aln_dist.reshape(-1, 4)[perm].ravel(), r2.aln_dist)
    assert np.array_equal(r1.aln_status.reshape(-1, 4)[perm].ravel(), r2.aln_status)
    assert np.array_equal(r1.sc_phase[perm], r2.sc_phase)
    for h in range(4):
        off = b.var_off[h]
        cnt = np.diff(off)[perm]
        idx = np.repeat(off[:-1][perm] - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt) + np.arange(int(cnt.sum()))
        for w in range(2):
            assert np.array_equal(r1.errtype[h][w][idx], r2.errtype[h][w])
            assert np.array_equal(r1.credit[h][w][idx].view(np.uint32), r2.credit[h][w].view(np.uint32))
            assert np.array_equal(r1.sync_group[h][w][idx], r2.sync_group[h][w])
            assert np.array_equal(r1.ref_ed[h][w][idx], r2.ref_ed[h][w])
    pr.execute()
    assert not r1.diff(pr.download())


@pytest.mark.parametrize("scaling,workload", [("weak", "wgs_synth"), ("strong", "wgs_synth"), ("strong", "joint_synth")])
def test_bench_self_launch_two_ranks_on_one_gpu(scaling, workload):
    """`python bench.py --gpus 2` as a plain process (no torchrun): it must re-launch itself as two ranks and print n_gpus 2.
    VCFDIST_BENCH_ONE_GPU puts both ranks on the test box's one GPU with the counters' all-reduce over gloo (a plumbing
    check: the numbers of such a run mean nothing); on an 8-GPU node the same command runs one rank per GPU over RCCL."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    env["VCFDIST_BENCH_ONE_GPU"] = "1"
    extra = ["--scaling", "strong", "--n-sc-total", "30000"] if scaling == "strong" else []
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--n-sc", "20000", "--steps", "2", "--warmup", "1",
                          "--no-cpu-baseline", "--workload", workload] + extra, env=env, cwd=root, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-3000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["steps"] == 2 and line["value"] > 0 and line["scaling"] == scaling
    assert line["config"]["workload"] == workload and list(line)[-1] == "summary" and line["summary"]["n_gpus"] == 2
    if workload == "joint_synth":   # the SV row of the joint evaluation is populated on the summed counters
        sv = [r for r in line["pr_summary_rank0"] if r["type"] == "SV"][0]
        assert sv["truth_tp"] + sv["truth_fn"] > 0
    if scaling == "weak":
        assert line["config"]["superclusters_per_gpu"] == 20000
    else:       # one genome dealt over the ranks by estimated cells: the shares add up, the phasing of all superclusters was gathered
        assert "30000 superclusters dealt over 2 ranks" in line["config"]["sharding"]
        assert 10000 < line["config"]["superclusters_per_gpu"] < 20000
    # every rank's own figures are in the line
    pr_ = line["per_rank"]
    assert len(pr_["ms_per_step"]) == 2 and all(x > 0 for x in pr_["ms_per_step"]) and len(pr_["collective_ms_per_step"]) == 2


@pytest.mark.parametrize("kw", [
    dict(n_sc=30000, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=3000, seed=51),        # long part + short part
    dict(n_sc=3000, len_a=5, len_b=200, len_min=5, len_max=200, seed=52, var_per_base=0.05),      # no long part
    dict(n_sc=7, len_a=1100, len_b=1500, len_min=1100, len_max=1500, seed=53),                    # long part only
])
def test_device_planner_equals_the_host_planner(kw, monkeypatch):
    """round 0's plan built on the device (radix sort by rows, scan of the workspace needs, wave headers: plan0_device) against
    make_plan on the host (VPR_HOST_PLAN): same results, same walks (vpr_download_path looks an alignment's place in the plan
    up), from host arrays and from variant tables"""
    syn = api.Synth(**kw)
    batch = syn.batch()
    cfg = A.default_config(flags=A.CFG_KEEP_PATHS)
    pr_d = api.PrecisionRecall(cfg)
    got_d = pr_d.run(batch)
    monkeypatch.setenv("VPR_HOST_PLAN", "1")
    pr_h = api.PrecisionRecall(cfg)
    got_h = pr_h.run(batch)
    monkeypatch.delenv("VPR_HOST_PLAN")
    assert not got_d.diff(got_h)
    td, th = pr_d.timing(), pr_h.timing()
    assert td.cells_touched == th.cells_touched and td.n_band_retries == th.n_band_retries
    rng = np.random.RandomState(5)
    for sc in rng.choice(batch.n_sc, size=min(batch.n_sc, 40), replace=False):
        for aln in range(4):
            a, b = pr_d.path(int(sc), aln), pr_h.path(int(sc), aln)
            assert all(np.array_equal(x, y) for x, y in zip(a, b)), (sc, aln)
    pr_v = api.PrecisionRecall()
    pr_v.upload_variants(syn.struct, batch)
    pr_v.execute()
    assert not got_d.diff(pr_v.download())
