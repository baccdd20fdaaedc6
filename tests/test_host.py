"""CPU tests of the product's host side: the C-ABI library loads and exports every symbol
include/vcfdist_pr.h declares, host marshalling equals the oracle's generate_ptrs_strs,
store_phase matches, the generator is deterministic, and without a GPU the library fails
loudly (no fallback)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O
from vcfdist_amd import _abi as A
from vcfdist_amd import api

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    api.build()
    hdr = open(os.path.join(ROOT, "include", "vcfdist_pr.h")).read()
    declared = set(re.findall(r"\b(vpr_[a-z_0-9]+)\s*\(", hdr))
    assert declared, "no declarations parsed"
    L = api.lib()
    missing = [n for n in sorted(declared) if not hasattr(L, n)]
    assert not missing, missing
    assert set(api.EXPORTED) <= declared


@pytest.mark.parametrize("header,prefix", [("vcfdist_cluster.h", "vcl_"), ("vcfdist_io.h", "vio_"), ("vcfdist_report.h", "vrp_")])
def test_library_exports_the_other_headers_too(header, prefix):
    api.build()
    hdr = open(os.path.join(ROOT, "include", header)).read()
    declared = set(re.findall(r"\b(%s[a-z_0-9]+)\s*\(" % prefix, hdr))
    assert len(declared) >= 4
    L = api.lib()
    assert not [n for n in sorted(declared) if not hasattr(L, n)]


def test_struct_sizes_match_header():
    # spot-check the ctypes mirror against the C layout through a round trip of the synth params
    p = api.synth_params(n_sc=3, seed=9)
    assert p.n_sc == 3 and p.seed == 9 and abs(p.var_per_base - 1 / 200) < 1e-12 and p.max_qual == 60
    assert p.p_sv == 0.0 and p.sv_min == 50 and p.sv_max == 10000        # (the last fields: the struct's size is right)


def test_joint_generator_adds_svs_and_keeps_the_small_variant_mix():
    """vpr_synth_params::p_sv (BASELINE configs[3]): a fraction of the superclusters carries one SV-sized indel, drawn from a PRNG
    stream of its own -- the other superclusters are those of the same seed without SVs; the oracle evaluates the batch and the
    summary's SV row is populated"""
    import numpy as np
    import oracle_lib as O
    from vcfdist_amd import summary as S
    kw = dict(n_sc=300, seed=77, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10002)
    plain, joint = api.Synth(**kw), api.Synth(p_sv=0.05, sv_min=50, sv_max=2000, **kw)
    bp, bj = plain.batch(), joint.batch()
    Lp, Lj = np.diff(bp.ref_off), np.diff(bj.ref_off)
    cls = joint.var_class()
    n_sv = sum(int((c == 2).sum()) for c in cls)
    assert 0 < (Lp != Lj).sum() <= 30 and n_sv >= 10 and all(int((c == 2).sum()) == 0 for c in plain.var_class())
    same = np.flatnonzero(Lp == Lj)
    k = int(same[0])      # a supercluster without an SV: the same variants at the same offsets inside its span
    for h in range(4):
        a = bp.var_pos[h][bp.var_off[h][k]:bp.var_off[h][k + 1]] - 0
        b = bj.var_pos[h][bj.var_off[h][k]:bj.var_off[h][k + 1]] - 0
        assert len(a) == len(b)
    res = O.run(bj)
    assert not (res.aln_status & (A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED)).any() and res.aln_dist.max() >= 50
    pb, _, _ = S.phase(res.sc_phase, np.ones(bj.n_sc, np.int32), L=O.lib(), prefix="vso")
    rows = S.pr_summary(O.oracle_pr_counts(O.lib(), bj.var_off, res, cls, pb), L=O.lib(), prefix="vso")
    sv = [r for r in rows if S.NAMES[r.vartype] == "SV" and not r.best][0]
    assert sv.truth_tp + sv.truth_fn > 0 and sv.query_tp + sv.query_fp > 0


def test_planning_pool_never_outnumbers_a_small_cpu_quota():
    """round 3's unsigned `lim - max(2, lim / 4)` wrapped around for a quota of 1 CPU (quota 8 shared by 8 local ranks) and
    started 31 detached workers"""
    L = api.lib()
    L.vpr_test_pool_workers.restype = C.c_int32
    L.vpr_test_pool_workers.argtypes = [C.c_int32]
    got = {q: L.vpr_test_pool_workers(q) for q in (1, 2, 3, 4, 8, 16, 64, 256)}
    assert got[1] == got[2] == got[3] == 0
    assert got[4] == 1 and got[8] == 5 and got[16] == 11
    assert all(0 <= w < max(q, 2) for q, w in got.items()) and got[256] <= 32


def test_timing_struct_matches_the_header_field_for_field():
    hdr = open(os.path.join(ROOT, "include", "vcfdist_pr.h")).read()
    body = hdr[hdr.index("typedef struct vpr_timing {"):hdr.index("} vpr_timing;")]
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    names = [n for decl in re.findall(r"(?:double|int64_t)\s+([^;]+);", body) for n in re.findall(r"([a-z_0-9]+)(?:\[\d+\])?", decl)]
    assert names == [f[0] for f in A.VprTiming._fields_]


def test_no_gpu_is_a_hard_error():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.VprError, match="no HIP device|no CPU fallback"):
        api.PrecisionRecall()


@pytest.mark.parametrize("kw", [
    dict(n_sc=300, len_a=6, len_b=80, len_min=5, len_max=80, var_per_base=0.1, p_snp=0.5, p_repeat=0.5, seed=3),
    dict(n_sc=50, len_a=100, len_b=3000, len_max=3000, seed=4),
])
def test_host_marshalling_equals_oracle_generate(kw):
    syn = api.Synth(**kw)
    mine = syn.batch()
    ref = O.generate(syn.variants())
    for h in range(4):
        for f in ("hap_off", "hap_seq", "hap_ptr", "hap_flag", "var_off", "var_pos", "var_qual"):
            assert np.array_equal(getattr(mine, f)[h], getattr(ref, f)[h]), (f, h)
    assert np.array_equal(mine.ref_off, ref.ref_off) and np.array_equal(mine.ref_seq, ref.ref_seq)
    for h in range(2):
        assert np.array_equal(mine.ref_ptr[h], ref.ref_ptr[h]) and np.array_equal(mine.ref_flag[h], ref.ref_flag[h])


def test_marshalling_rejects_overlapping_variants():
    ref = "ACGTACGTACGTACGT"
    bad = [(5, A.TYPE_DEL, "CGT", "", 10.0), (6, A.TYPE_SUB, "G", "A", 10.0)]
    v = A.Variants.from_sites([ref], [dict(ctg=0, beg=3, end=12, vars=[bad, [], [], []])])
    with pytest.raises(api.VprError):
        api.batch_from_variants(v)


def test_store_phase_matches_oracle():
    rng = np.random.RandomState(0)
    for _ in range(2000):
        s = rng.randint(0, 12, 4).tolist()
        assert api.store_phase(s) == O.store_phase(s), s


def test_synth_is_deterministic_and_seed_sensitive():
    a = api.Synth(n_sc=100, seed=42).variants()
    b = api.Synth(n_sc=100, seed=42).variants()
    c = api.Synth(n_sc=100, seed=43).variants()
    assert np.array_equal(a.ctg_seq, b.ctg_seq) and np.array_equal(a.var_pos[0], b.var_pos[0])
    assert not np.array_equal(a.ctg_seq, c.ctg_seq)


def test_batch_subset_roundtrip():
    b = api.Synth(n_sc=30, len_a=10, len_b=100, len_max=100, seed=1).batch()
    sub = b.subset([3, 7, 7, 29])
    assert sub.n_sc == 4 and sub.lens(1) == b.lens(7) and sub.lens(2) == b.lens(7) and sub.lens(3) == b.lens(29)
    r_all = O.run(b)
    r_sub = O.run(sub)
    assert r_sub.aln_dist.tolist() == np.concatenate([r_all.aln_dist[4 * i:4 * i + 4] for i in (3, 7, 7, 29)]).tolist()


def test_region_past_the_contig_end_is_cut_at_the_last_base():
    """A variant ending on one of the last two bases of a contig makes get_supercluster_range (cluster.cpp:595:
    pos + rlen + 1) return end >= contig length.  The reference has no defined result there (dist.cpp:232 substr comes
    back short of its pointer arrays, dist.cpp:539 takes the start cell from the pointer arrays' size: outside its
    matrices), so product marshalling and oracle both evaluate the region that exists -- cut at the contig's last base,
    strings and pointers consistent -- instead of aborting a whole-genome run.  A variant that itself reaches behind the
    contig, and a region that starts in front of it (the reference exits there too), stay errors.  Found by
    tests/fuzz_chain.py."""
    import numpy as np
    import oracle_lib as O
    from vcfdist_amd import _abi as A
    ctg = np.frombuffer(b"ACGTACGTACGTACGTACGT", np.uint8)          # 20 bases

    def variants(pos, rlen=1, beg=None):
        # one deletion of `rlen` bases at `pos` on Q1 and T1; region = [pos - 1, pos + rlen + 1] as the reference computes it
        n = [1, 0, 1, 0]
        z32, z64 = np.zeros(0, np.int32), np.zeros(0, np.int64)
        one = lambda x, dt: [np.array([x], dt) if k else np.zeros(0, dt) for k in n]
        return A.Variants(np.array([0, 20], np.int64), ctg, np.zeros(1, np.int32), np.array([pos - 1 if beg is None else beg], np.int32),
                          np.array([pos + rlen + 1], np.int32), [np.array([0, k], np.int64) for k in n],
                          one(pos, np.int32), one(3, np.uint8), one(30.0, np.float32), one(0, np.int64), one(rlen, np.int32),
                          one(rlen, np.int64), one(0, np.int32), [ctg[pos:pos + rlen].copy() if k else np.zeros(1, np.uint8) for k in n])
    for pos, rlen, want_ref, want_hap in ((17, 1, 4, 3),     # end = 19 = last base: the whole region
                                          (18, 1, 3, 2),     # end = 20: cut to [17, 19], one base behind the deletion
                                          (19, 1, 2, 1),     # end = 21: cut to [18, 19], the deletion ends the strings
                                          (17, 3, 4, 1)):    # a three-base deletion of the contig's last bases
        v = variants(pos, rlen)
        for b in (api.batch_from_variants(v), O.generate(v)):
            assert b.lens(0)[4] == want_ref and b.lens(0)[0] == want_hap and b.lens(0)[1] == want_ref, (pos, rlen, b.lens(0))
        p, o = api.batch_from_variants(v), O.generate(v)
        for f in ("hap_seq", "hap_ptr", "hap_flag"):
            assert all(np.array_equal(getattr(p, f)[h], getattr(o, f)[h]) for h in range(4)), f
        res = O.run(o)
        res = res[0] if isinstance(res, tuple) else res
        assert res.aln_dist[[1, 2, 3]].tolist() == [0, rlen, 0]        # Q1 reaches T2 over the REF plane; T1 carries the deletion, Q2 does not
        if pos + rlen < 20:      # a base behind the variant: an ordinary supercluster, TP / TP under the original phasing
            assert res.aln_dist[0] == 0 and res.errtype[0][0].tolist() == [0] and res.errtype[2][0].tolist() == [0]
        else:                    # the variant ends the strings (a one-base haplotype against a one-base truth: the reference's
            # backward pass would never terminate there, dist.cpp:549-687; here the path is the one cell)
            assert res.aln_dist[0] == 0 and res.aln_status[0] == 0
    for bad in (variants(18, 3), variants(0, 1)):                     # variant leaves the contig / region starts at -1
        with pytest.raises(api.VprError):
            api.batch_from_variants(bad)
        with pytest.raises(ValueError):
            O.generate(bad)


def test_transfer_phase_sets_matches_the_reference_walk():
    """vcfdist_amd.__main__.transfer_phase_sets (candidate sweep) against a direct restatement of the reference's loop
    (superclusterData::transfer_phase_sets, cluster.cpp:186-330) on random phase sets"""
    from vcfdist_amd.__main__ import transfer_phase_sets

    class SC:
        def __init__(self, offs):
            self.n = len(offs[0]) - 1
            self._o = offs

        def var_off(self, i):
            return self._o[i]

    def walk(slots, sc):
        first_pos, phase_set = None, 0
        for s in slots:
            nz = np.nonzero(s["phase_set"])[0]
            if len(nz) and (first_pos is None or s["pos"][nz[0]] < first_pos):
                first_pos, phase_set = int(s["pos"][nz[0]]), int(s["phase_set"][nz[0]])
        cur = [0, 0]
        out = np.zeros(sc.n, np.int32)
        for k in range(sc.n):
            for i, s in enumerate(slots):
                for v in range(int(sc.var_off(i)[k]), int(sc.var_off(i)[k + 1])):
                    ps = int(s["phase_set"][v])
                    if ps and ps > cur[i >> 1]:
                        phase_set = cur[i >> 1] = ps
            out[k] = phase_set
        return out

    rng = np.random.RandomState(11)
    for trial in range(60):
        n_sc = int(rng.randint(1, 40))
        offs, slots = [], []
        for i in range(4):
            cnt = rng.randint(0, 4, size=n_sc) if rng.rand() > 0.1 else np.zeros(n_sc, int)
            off = np.concatenate(([0], np.cumsum(cnt))).astype(np.int64)
            n = int(off[-1])
            pos = np.sort(rng.randint(0, 100000, size=n)).astype(np.int32)
            # phase sets: mostly increasing along the hap (PS = position of the block's first variant), some zeros, some disorder
            ps = np.where(rng.rand(n) < 0.2, 0, (pos // int(rng.randint(500, 20000))) * 1000 + 1).astype(np.int32)
            if n and rng.rand() < 0.3:
                ps[rng.randint(0, n)] = int(rng.randint(1, 100000))
            offs.append(off); slots.append(dict(pos=pos, phase_set=ps))
        sc = SC(offs)
        assert np.array_equal(transfer_phase_sets(slots, None, sc), walk(slots, sc)), trial


def _build_caller(tmp_path):
    import subprocess
    exe = str(tmp_path / "caller")
    subprocess.run(["g++", "-std=c++17", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "caller.cpp"),
                    "-L", os.path.join(ROOT, "vcfdist_amd", "lib"), "-lvcfdist_pr", "-Wl,-rpath," + os.path.join(ROOT, "vcfdist_amd", "lib"),
                    "-o", exe], check=True)
    return exe


def test_compiled_cpp_caller_builds_and_refuses_without_a_gpu(tmp_path):
    """examples/caller.cpp (a plain g++ program against include/vcfdist_pr.h + the shared library): builds, and without a
    HIP device stops at vpr_create with the library's "no CPU fallback" error; on a GPU box it prints the known answer"""
    import subprocess
    api.build()
    r = subprocess.run([_build_caller(tmp_path)], capture_output=True, text=True, timeout=120)
    if r.returncode == 2:
        assert "no CPU fallback" in r.stderr
    else:
        assert r.returncode == 0 and "alignment 0: s 0 end plane QUERY" in r.stdout, r.stdout + r.stderr


@pytest.mark.gpu
def test_compiled_cpp_caller_known_answer(tmp_path):
    """the reference-produced toy vector through a compiled C++ caller of the C ABI"""
    import subprocess
    r = subprocess.run([_build_caller(tmp_path)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    for i in range(4):
        assert f"alignment {i}: s 0 end plane QUERY" in r.stdout
    assert "hap slot 2 variant 0: TP" in r.stdout


def test_cxx_command_line_builds_reads_its_inputs_and_refuses_without_a_gpu(tmp_path):
    """lib/vcfdist_gpu (vcfdist_amd/csrc/main.cpp: the command line with the host orchestration in C++): built by the library's
    Makefile; on the demo inputs it parses its arguments, reads both VCFs, the BED and a FASTA through the C readers, orders the
    contigs -- and without a HIP device stops at vpr_create with the library's "no CPU fallback" error (exit code 2).  On a GPU
    box the same command runs through (tests/test_demo_known_answer.py compares its files with the Python driver's)."""
    import subprocess
    import demo_pipeline as D
    api.build()
    exe = os.path.join(ROOT, "vcfdist_amd", "lib", "vcfdist_gpu")
    assert os.path.exists(exe)
    fa = tmp_path / "s.fa"
    with open(fa, "w") as fh:
        fh.write(">chr1 surrogate\n" + bytes(D.surrogate_fasta(5_100_000)).decode() + "\n")
    r = subprocess.run([exe, os.path.join(D.DEMO, "query.vcf"), os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.vcf.gz"), str(fa),
                        "-b", os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.bed"), "-n"], capture_output=True, text=True, timeout=300)
    if r.returncode == 2:
        assert "no CPU fallback" in r.stderr
    else:
        assert r.returncode == 0 and "PRECISION-RECALL SUMMARY" in r.stdout, r.stdout + r.stderr
    bad = subprocess.run([exe, "a.vcf", "b.vcf", "c.fa", "-l", "10000"], capture_output=True, text=True, timeout=60)
    assert bad.returncode == 1 and "at least two larger" in bad.stderr          # globals.cpp:478-481
