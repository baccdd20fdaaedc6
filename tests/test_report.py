"""Output writers (include/vcfdist_report.h) against the Python restatement of the reference's writers, byte for byte.
Host code only: no GPU needed."""
import types

import numpy as np
import pytest

import report_oracle as RO
from vcfdist_amd import cluster as K, report as R


def _pack(strs):
    off, pool, o = [], bytearray(), 0
    for s in strs:
        off.append(o)
        pool += s.encode()
        o += len(s)
    return np.array(off, np.int64), np.frombuffer(bytes(pool) or b"\0", np.uint8).copy()


def make_contig(rng, name, n_sc, ploidy=2, empty_slot=None):
    seq = bytes(rng.choice(list(b"ACGT"), size=200 * n_sc + 400).tolist())
    slots = [dict(pos=[], type=[], ref=[], alt=[], var_qual=[], phase_set=[]) for _ in range(4)]
    clusters = [[0] for _ in range(4)]
    brk = [[0] for _ in range(4)]
    sc_beg, sc_end = [], []
    for k in range(n_sc):
        beg = 100 + 200 * k
        end = beg + int(rng.integers(20, 120))
        sc_beg.append(beg)
        sc_end.append(end)
        shared = None
        for i in range(4):
            if i == empty_slot or (ploidy == 1 and i in (1, 3)):
                nv = 0
            else:
                nv = int(rng.integers(0, 4))
            pos = np.sort(rng.integers(beg + 1, end - 1, size=nv))
            for j, p in enumerate(pos):
                t = int(rng.integers(1, 4))
                if t == 1:
                    ref, alt = chr(seq[p]), "ACGT"[rng.integers(0, 4)]
                elif t == 2:
                    ref, alt = "", "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5))))
                else:
                    ref, alt = "".join(rng.choice(list("ACGT"), size=int(rng.integers(1, 5)))), ""
                rec = (int(p), t, ref, alt)
                if i == 0 and j == 0:
                    shared = rec
                if i >= 2 and j == 0 and shared is not None and rng.random() < 0.5:   # truth repeats a query variant
                    rec = shared
                slots[i]["pos"].append(rec[0]); slots[i]["type"].append(rec[1]); slots[i]["ref"].append(rec[2]); slots[i]["alt"].append(rec[3])
                slots[i]["var_qual"].append(float(np.float32(rng.uniform(0, 60))))
                slots[i]["phase_set"].append(int(rng.integers(0, 3)) * 1000)
            slots[i]["pos"][len(slots[i]["pos"]) - nv:] = sorted(slots[i]["pos"][len(slots[i]["pos"]) - nv:])
            if nv:
                n0 = len(slots[i]["pos"]) - nv
                cuts = [n0] + ([n0 + int(rng.integers(1, nv))] if nv > 1 and rng.random() < 0.5 else [])
                clusters[i] = clusters[i][:-1] + cuts + [len(slots[i]["pos"])]
            brk[i].append(len(clusters[i]) - 1)
    for i in range(4):
        if not slots[i]["pos"]:
            clusters[i] = []
    sc_phase = rng.integers(0, 3, size=n_sc).astype(np.int32)
    pb_phase = rng.integers(0, 2, size=n_sc).astype(np.int32)
    sc_phase_set = np.sort(rng.integers(0, 3, size=n_sc)).astype(np.int32) * 7
    # phasing errors as phaseblockData::phase() produces them: behind the first supercluster, a supercluster is a switch or
    # a flip but not both, and no switch error on the first supercluster of a phase block (phase.cpp:310)
    starts = set(RO.phase_blocks(sc_phase_set))
    ev = rng.permutation(np.arange(1, max(n_sc, 1)))[:min(5, max(n_sc - 1, 0))]
    switches = np.sort([e for e in ev[:2] if int(e) not in starts]).astype(np.int32)
    flips = np.sort(ev[2:]).astype(np.int32)
    res = types.SimpleNamespace(sc_phase=sc_phase, orig_phase_dist=rng.integers(0, 50, size=n_sc).astype(np.int32),
                                swap_phase_dist=rng.integers(0, 50, size=n_sc).astype(np.int32))
    for nm, gen in (("errtype", lambda n: rng.choice([0, 1, 2, 5], size=n).astype(np.uint8)),
                    ("credit", lambda n: rng.choice([0.0, 1.0, 0.5, 0.7, 0.6999, 0.333333], size=n).astype(np.float32)),
                    ("sync_group", lambda n: rng.integers(0, 5, size=n).astype(np.int32)),
                    ("ref_ed", lambda n: rng.integers(0, 4, size=n).astype(np.int32)),
                    ("query_ed", lambda n: rng.integers(0, 9, size=n).astype(np.int32))):
        setattr(res, nm, [[gen(len(slots[i]["pos"])) for _ in range(2)] for i in range(4)])
    # product view
    pslots, pclusters = [], []
    for i in range(4):
        s = slots[i]
        ro, rp = _pack(s["ref"])
        ao, ap = _pack(s["alt"])
        pslots.append(dict(pos=np.array(s["pos"], np.int32), type=np.array(s["type"], np.uint8), var_qual=np.array(s["var_qual"], np.float32),
                           phase_set=np.array(s["phase_set"], np.int32), ref_len=np.array([len(x) for x in s["ref"]], np.int32),
                           alt_len=np.array([len(x) for x in s["alt"]], np.int32), ref_off=ro, alt_off=ao + len(rp),
                           pool=np.concatenate([rp, ap])))
        z = np.zeros(len(clusters[i]), np.int32)
        pclusters.append(K.Clusters(np.array(clusters[i], np.int32), z, z))
    sc = types.SimpleNamespace(n=n_sc, beg=np.array(sc_beg, np.int32), end=np.array(sc_end, np.int32),
                               brk=[np.array(b, np.int32) for b in brk])
    prod = R.Contig(name, len(seq), ploidy, seq, pslots, pclusters, sc, res, sc_phase_set, pb_phase, switches, flips)
    orc = RO.Ctg(name=name, length=len(seq), ploidy=ploidy, seq=seq, slots=slots, clusters=clusters, sc_beg=sc_beg, sc_end=sc_end,
                 sc_brk=brk, sc_phase=sc_phase, pb_phase=pb_phase, orig_dist=res.orig_phase_dist, swap_dist=res.swap_phase_dist,
                 sc_phase_set=sc_phase_set, switches=switches, flips=flips, res=res)
    return prod, orc


def _read(p):
    with open(p, "rb") as f:
        return f.read().decode()


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_result_tables_match_restatement(tmp_path, seed):
    rng = np.random.default_rng(seed)
    pairs = [make_contig(rng, "chr1", 40), make_contig(rng, "chrX", 12, ploidy=1), make_contig(rng, "chrE", 0),
             make_contig(rng, "chr9", 9, empty_slot=int(rng.integers(0, 4)))]
    prefix = str(tmp_path) + "/out_"
    R.write_results(prefix, [p for p, _ in pairs], cmd="vcfdist q.vcf t.vcf r.fa", file_date="20250919", credit_threshold=0.7)
    orc = [o for _, o in pairs]
    assert _read(prefix + "phase-blocks.tsv") == RO.phase_blocks_tsv(orc)
    assert _read(prefix + "switchflips.tsv") == RO.switchflips_tsv(orc)
    assert _read(prefix + "phasing-summary.tsv") == RO.phasing_summary_tsv(orc)
    assert _read(prefix + "switchflips.tsv").count("\n") > 6
    assert _read(prefix + "superclusters.tsv") == RO.superclusters_tsv(orc)
    assert _read(prefix + "query.tsv") == RO.variants_tsv(orc, 0)
    assert _read(prefix + "truth.tsv") == RO.variants_tsv(orc, 1)
    assert _read(prefix + "summary.vcf") == RO.summary_vcf(orc, "vcfdist q.vcf t.vcf r.fa", "20250919", 0.7)
    # the files are not trivially empty
    assert _read(prefix + "query.tsv").count("\n") > 40 and _read(prefix + "summary.vcf").count("\tPASS\t") > 80


@pytest.mark.parametrize("seed,min_qual,max_qual", [(0, 0, 60), (1, 10, 30), (2, 0, 0)])
def test_precision_recall_tables_match_restatement(tmp_path, seed, min_qual, max_qual):
    rng = np.random.default_rng(seed)
    nq = max_qual - min_qual + 1
    # cumulative shapes like the real counters: TP falls, FN rises with the threshold
    counts = np.zeros((2, 4, 3, nq), np.int64)
    for cs in range(2):
        for t in range(3):
            tot = int(rng.integers(0, 5000)) if (seed, t) != (1, 2) else 0     # one empty class
            tp = np.sort(rng.integers(0, tot + 1, size=nq))[::-1]
            counts[cs, t, 0] = tp
            counts[cs, t, 1 if cs == 0 else 2] = rng.integers(0, 50, size=nq) + (tot - tp if cs == 1 else 0)
        counts[cs, 3] = counts[cs, :3].sum(axis=0)
    prefix = str(tmp_path) + "/"
    R.write_precision_recall(prefix, counts, min_qual, max_qual)
    a, s = RO.precision_recall(counts, min_qual, max_qual)
    assert _read(prefix + "precision-recall.tsv") == a
    assert _read(prefix + "precision-recall-summary.tsv") == s
    assert a.count("\n") == 1 + 4 * nq and s.count("\n") == 9


def test_phase_blocks():
    assert list(R.phase_blocks([5, 5, 5, 9, 9, 0, 5])) == [0, 3, 5, 6, 7]
    assert list(R.phase_blocks([])) == [0]
    assert list(R.phase_blocks([0, 0])) == [0, 2]
    assert RO.phase_blocks([5, 5, 5, 9, 9, 0, 5]) == [0, 3, 5, 6, 7]


def test_unwritable_path_is_an_error(tmp_path):
    with pytest.raises(R.ReportError):
        R.write_precision_recall(str(tmp_path) + "/no/such/dir/", np.zeros((2, 4, 3, 61), np.int64), 0, 60)


def test_file_headers_are_the_documented_columns():
    """the column names and their order of every table, as the reference documents them
    (docs/v2.6.0/09-Outputs.md; the one reference-held description of these files).  Three deliberate differences of the
    reference's own writers from its documentation are kept, because the writers are what a user's scripts read:
    query.tsv / truth.tsv say ERRTYPE where the page says ERR_TYPE (print.cpp:676, 778), and precision-recall-summary.tsv
    has a THRESHOLD column (NONE / BEST) the page does not list (print.cpp:504)."""
    documented = {
        "query.tsv": "CONTIG POS HAP REF ALT QUAL TYPE ERR_TYPE CREDIT CLUSTER SUPERCLUSTER SYNC_GROUP REF_DIST QUERY_DIST LOCATION",
        "truth.tsv": "CONTIG POS HAP REF ALT QUAL TYPE ERR_TYPE CREDIT CLUSTER SUPERCLUSTER SYNC_GROUP REF_DIST QUERY_DIST LOCATION",
        "superclusters.tsv": "CONTIG SUPERCLUSTER START STOP SIZE QUERY1_VARS QUERY2_VARS TRUTH1_VARS TRUTH2_VARS ORIG_ED SWAP_ED "
                             "PHASE_STATE SC_PHASE PHASE_SET PHASE_BLOCK FLIP_ERROR",
        "precision-recall-summary.tsv": "VAR_TYPE MIN_QUAL TRUTH_TP QUERY_TP TRUTH_FN QUERY_FP PREC RECALL F1_SCORE F1_QSCORE",
        "precision-recall.tsv": "VAR_TYPE MIN_QUAL PREC RECALL F1_SCORE F1_QSCORE TRUTH_TOTAL TRUTH_TP TRUTH_FN QUERY_TOTAL QUERY_TP QUERY_FP",
        "phase-blocks.tsv": "CONTIG PHASE_BLOCK START STOP SIZE SUPERCLUSTERS FLIP_ERRORS SWITCH_ERRORS",
        "phasing-summary.tsv": "PHASE_BLOCKS SWITCH_ERRORS FLIP_ERRORS NG_50 SWITCH_NGC50 SWITCHFLIP_NGC50",
        "switchflips.tsv": "CONTIG START STOP SWITCH_TYPE SUPERCLUSTER PHASE_BLOCK",
    }
    import tempfile
    rng = np.random.default_rng(5)
    prod, _ = make_contig(rng, "chr1", 6)
    with tempfile.TemporaryDirectory() as d:
        prefix = d + "/"
        R.write_results(prefix, [prod], cmd="vcfdist", file_date="20250919")
        R.write_precision_recall(prefix, np.zeros((2, 4, 3, 61), np.int64), 0, 60)
        for name, cols in documented.items():
            got = _read(prefix + name).split("\n")[0].split("\t")
            want = cols.split()
            if name in ("query.tsv", "truth.tsv"):
                want[want.index("ERR_TYPE")] = "ERRTYPE"
            if name == "precision-recall-summary.tsv":
                want.insert(1, "THRESHOLD")
            assert got == want, name
        # summary.vcf: the FORMAT keys of the documented table, in the documented order (phase.cpp:44-70)
        vcf = _read(prefix + "summary.vcf")
        keys = [l.split("ID=")[1].split(",")[0] for l in vcf.split("\n") if l.startswith("##FORMAT")]
        assert keys == "GT BD BC RD QD BK QQ SC SG PS PB BS FE".split()
        body = [l for l in vcf.split("\n") if l and not l.startswith("#")]
        assert body and all(l.split("\t")[8] == "GT:BD:BC:RD:QD:BK:QQ:SC:SG:PS:PB:BS:FE" for l in body)
