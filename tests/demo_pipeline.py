"""End-to-end pipeline on the reference's demo inputs (tests/golden/demo/: the data files of /root/reference/demo),
TEST INFRASTRUCTURE.  The front end -- VCF parsing, BED filtering -- is a Python restatement of the reference
(variant.cpp:397-875 for the record logic, bed.cpp:73-121 for the region test); everything behind it runs either the
oracle chain (oracle/*.cpp) or the product chain (HIP library), selected by `product`.

The demo's FASTA (GRCh38 chr1:1-5Mb) is not in the repository, so a seeded surrogate is built: uniform random bases
with the REF allele of every record of both VCFs written at its position (the VCFs are therefore consistent with
it).  With it the SNP rows of demo/output.txt are reproduced exactly; the INDEL rows depend on the real repeat
context around the indels and are off by a few counts (recorded in the test)."""
import gzip
import os

import numpy as np

import oracle_lib as O
from vcfdist_amd import _abi as A
from vcfdist_amd import api, cluster as K, summary as S

DEMO = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "demo")
TYPE_REF, TYPE_SUB, TYPE_INS, TYPE_DEL, TYPE_CPX = 0, 1, 2, 3, 4
BED_INSIDE, BED_BORDER, BED_OUTSIDE, BED_OFFCTG = 0, 1, 2, 3
G = dict(min_qual=0, max_qual=60, max_size=5000, cluster_min_gap=50, reach_min_gap=10, max_cluster_itrs=4,
         max_supercluster_size=10000, sub=5, open=6, extend=2, phase_threshold=0.6, credit_threshold=0.7, sv_threshold=50)


def read_vcf_records(path):
    op = gzip.open if path.endswith(".gz") else open
    with op(path, "rt") as fh:
        for line in fh:
            if line.startswith("#"):
                continue
            f = line.rstrip("\n").split("\t")
            yield f


class Bed:
    """bedData (bed.cpp:4-36) + contains (bed.cpp:73-121)"""

    def __init__(self, path):
        self.starts, self.stops = {}, {}
        for line in open(path):
            c, a, b = line.split("\t")[:3]
            self.starts.setdefault(c, []).append(int(a))
            self.stops.setdefault(c, []).append(int(b))
        for c in self.starts:
            self.starts[c] = np.array(self.starts[c]); self.stops[c] = np.array(self.stops[c])

    def contains(self, ctg, start, stop, typ):
        if ctg not in self.starts:
            return BED_OFFCTG
        st, sp = self.starts[ctg], self.stops[ctg]
        if stop <= st[0] or start >= sp[-1]:
            return BED_OUTSIDE
        start_idx = int(np.searchsorted(st, start, side="right")) - 1     # upper_bound - 1
        stop_idx = int(np.searchsorted(sp, stop, side="left"))            # lower_bound
        if start_idx < 0 or stop_idx >= len(sp):
            return BED_BORDER
        if stop_idx == start_idx:
            if typ == TYPE_INS and start == sp[stop_idx] - 1:
                return BED_BORDER
            return BED_INSIDE
        if stop_idx == start_idx + 1:
            if start >= sp[start_idx] and stop <= st[stop_idx]:
                return BED_OUTSIDE
            return BED_BORDER
        return BED_BORDER


def parse_vcf(path, bed, ctg="chr1", bed_policy="v2.6.4"):
    """variantData::variantData, variant.cpp:556-875, for single-sample diploid records without PS tags.
    Returns per hap a dict of lists (pos, rlen, type, ref, alt, qual).

    bed_policy "v2.6.4": the reference as it is (variant.cpp:826-840): the ORIGINAL record span (anchor base included) must
    be INSIDE, BORDER records are dropped.  bed_policy "v2.3": what docs/v2.3.3 and docs/v2.3.4/03-Variant-Filtering.md
    describe ("Variants on the border of BED regions are currently included (to match with vcfeval)", the anchor base
    not counted -- changed in v2.4.0, whose page reads "are excluded ... including if the preceding reference base in
    the VCF overlaps"): the TRIMMED span is tested and BORDER records are kept.  demo/output.txt's FASTA-independent
    totals are those of the "v2.3" rule (tests/test_demo_known_answer.py)."""
    haps = [dict(pos=[], rlen=[], type=[], ref=[], alt=[], qual=[]) for _ in range(2)]
    prev_end = [-2 * G["cluster_min_gap"]] * 2
    prev_type = [TYPE_SUB] * 2
    stats = dict(n=0, unphased=0, overlap=0, border=0, outside=0, cpx=0, refcall=0)
    for f in read_vcf_records(path):
        if f[0] != ctg:
            continue
        stats["n"] += 1
        rpos = int(f[1]) - 1
        alleles = [f[3]] + f[4].split(",")
        vq = 0.0 if f[5] == "." else float(f[5])
        if vq < G["min_qual"]:
            continue
        fmt = f[8].split(":")
        gt_s = f[9].split(":")[fmt.index("GT")]
        phased = "|" in gt_s
        gt = gt_s.replace("|", "/").split("/")
        ngt = len(gt)
        same = ngt == 2 and "." not in gt and gt[0] == gt[1]
        for hap in range(ngt):
            ref = alleles[0]
            if gt[hap] == ".":
                continue                                    # unknown allele
            alt_idx = int(gt[hap])
            if alt_idx == 0:
                continue
            alt = alleles[alt_idx]
            if ngt == 2 and not same and not phased:        # unphased heterozygous
                stats["unphased"] += 1
                continue
            if alt == "*":
                continue
            pos = rpos
            lm, rm = 0, -1
            reflen, altlen = len(ref), len(alt)
            if altlen - reflen > 0:
                while lm < reflen and ref[lm] == alt[lm]:
                    lm += 1
                while reflen + rm >= lm and ref[reflen + rm] == alt[altlen + rm]:
                    rm -= 1
                typ = TYPE_INS if lm > reflen + rm else TYPE_CPX
                pos += lm
                alt = alt[lm:altlen + rm + 1]
                ref = ref[lm:reflen + rm + 1]
            elif altlen - reflen < 0:
                while lm < altlen and ref[lm] == alt[lm]:
                    lm += 1
                while altlen + rm >= lm and ref[reflen + rm] == alt[altlen + rm]:
                    rm -= 1
                typ = TYPE_DEL if lm > altlen + rm else TYPE_CPX
                pos += lm
                alt = alt[lm:altlen + rm + 1]
                ref = ref[lm:reflen + rm + 1]
            else:
                if len(ref) == 1:
                    if ref[0] == alt[0]:
                        stats["refcall"] += 1
                        continue
                    typ = TYPE_SUB
                elif ref[1:] == alt[1:]:
                    typ = TYPE_SUB
                    ref, alt = ref[0], alt[0]
                else:
                    typ = TYPE_CPX
            rlen = {TYPE_INS: 0, TYPE_SUB: 1}.get(typ, len(ref))
            if bed_policy == "v2.6.4":
                loc = bed.contains(ctg, rpos, rpos + reflen, typ)
            else:
                loc = bed.contains(ctg, pos, pos + rlen, typ)
                if loc == BED_BORDER:
                    stats["border_kept"] = stats.get("border_kept", 0) + 1
                    loc = BED_INSIDE
            if loc != BED_INSIDE:
                stats["border" if loc == BED_BORDER else "outside"] += 1
                continue
            if len(ref) > G["max_size"] or len(alt) > G["max_size"]:
                continue
            if prev_end[hap] > pos or (prev_end[hap] == pos and prev_type[hap] == TYPE_INS and typ == TYPE_INS):
                stats["overlap"] += 1
                continue
            ref, alt = ref.upper(), alt.upper()
            q = min(vq, float(G["max_qual"]))
            H = haps[hap]
            if typ == TYPE_CPX:
                stats["cpx"] += 1
                for (p_, rl, t_, r_, a_) in ((pos, 0, TYPE_INS, "", alt), (pos, rlen, TYPE_DEL, ref, "")):
                    H["pos"].append(p_); H["rlen"].append(rl); H["type"].append(t_); H["ref"].append(r_); H["alt"].append(a_); H["qual"].append(q)
            else:
                H["pos"].append(pos); H["rlen"].append(rlen); H["type"].append(typ); H["ref"].append(ref); H["alt"].append(alt); H["qual"].append(q)
            prev_end[hap] = pos + rlen
            prev_type[hap] = typ
    return haps, stats


def surrogate_fasta(length, seed=0x5eed):
    """seeded random bases with the REF allele of every record of both demo VCFs at its position"""
    rng = np.random.RandomState(seed)
    seq = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=length)
    for name in ("nist-v4.2.1_chr1_5Mb.vcf.gz", "query.vcf"):
        for f in read_vcf_records(os.path.join(DEMO, name)):
            p = int(f[1]) - 1
            r = f[3].upper().encode()
            if p + len(r) <= length:
                seq[p:p + len(r)] = np.frombuffer(r, dtype=np.uint8)
    return seq


def run(product, ctg_len=5_100_000, cluster="biwfa", bed_policy="v2.6.4", fasta_seed=0x5eed):
    """-> (summary rows, details).  product = False: oracle chain on the CPU; True: HIP library (needs a GPU).
    cluster: "biwfa" (the reference's default) or ("gap", N) for `-c gap N` (simple_cluster, cluster.cpp:826-945)."""
    bed = Bed(os.path.join(DEMO, "nist-v4.2.1_chr1_5Mb.bed"))
    q, qs = parse_vcf(os.path.join(DEMO, "query.vcf"), bed, bed_policy=bed_policy)
    t, ts = parse_vcf(os.path.join(DEMO, "nist-v4.2.1_chr1_5Mb.vcf.gz"), bed, bed_policy=bed_policy)
    fasta = surrogate_fasta(ctg_len, fasta_seed)
    slots = [q[0], q[1], t[0], t[1]]
    haps = [K.HapSeq(s["pos"], s["type"], s["ref"], s["alt"]) for s in slots]
    lib = None if product else O.lib()
    pre = "vcl" if product else "vco"
    if cluster == "biwfa":
        cl = [K.wfa_cluster(h, bytes(fasta), sub=G["sub"], open=G["open"], extend=G["extend"], max_cluster_itrs=G["max_cluster_itrs"],
                            reach_min_gap=G["reach_min_gap"], L=lib, prefix=pre)[0] for h in haps]
    else:
        # simple_cluster adds g.reach_min_gap in both of its merge passes (cluster.cpp:895, 918; globals.h:33)
        cl = [K.simple_cluster(h, 0, int(cluster[1]), G["reach_min_gap"], L=lib, prefix=pre) for h in haps]
    sc = K.supercluster(haps, cl, G["max_supercluster_size"], L=lib, prefix=pre)
    pool, roff, aoff = [], [], []
    for h in haps:
        pool.append(h.pool); roff.append(h.ref_off); aoff.append(h.alt_off)
    v = A.Variants(np.array([0, ctg_len], np.int64), fasta, np.zeros(sc.n, np.int32), sc.beg, sc.end,
                   [sc.var_off(i) for i in range(4)], [h.pos for h in haps], [h.type for h in haps],
                   [np.asarray(s["qual"], np.float32) for s in slots], roff, [h.ref_len for h in haps], aoff,
                   [h.alt_len for h in haps], pool)
    cls = [S.var_class(h.type, h.ref_len, h.alt_len, G["sv_threshold"]) for h in haps]
    if product:
        batch = api.batch_from_variants(v)
        pr = api.PrecisionRecall()
        res = pr.run(batch)
        pb, sw, fl = S.phase(res.sc_phase, np.zeros(sc.n, np.int32))
        counts = S.pr_counts(pr, cls, pb, G["min_qual"], G["max_qual"])
        rows = S.pr_summary(counts, G["min_qual"], G["max_qual"])
    else:
        batch = O.generate(v)
        res = O.run(batch)
        res = res[0] if isinstance(res, tuple) else res
        pb, sw, fl = S.phase(res.sc_phase, np.zeros(sc.n, np.int32), L=lib, prefix="vso")
        counts = O.oracle_pr_counts(lib, batch.var_off, res, cls, pb, G["min_qual"], G["max_qual"])
        rows = S.pr_summary(counts, G["min_qual"], G["max_qual"], L=lib, prefix="vso")
    det = dict(query_stats=qs, truth_stats=ts, n_var=[len(h.pos) for h in haps], n_clusters=[c.n for c in cl], n_sc=sc.n,
               counts=counts, res=res, clusters=cl, sc=sc, batch=batch, slots=slots, pb=pb, switches=sw, flips=fl, fasta=fasta)
    return rows, det


def report_view(det, name="chr1", length=0, ploidy=2):
    """the oracle chain's results as the plain-data contig of tests/report_oracle.py (the demo has no PS tags)"""
    import report_oracle as RO
    sc = det["sc"]
    slots = [dict(pos=s["pos"], type=s["type"], ref=s["ref"], alt=s["alt"], var_qual=s["qual"], phase_set=[0] * len(s["pos"]))
             for s in det["slots"]]
    clusters = [list(c.var_beg) if c.n else [] for c in sc.clusters]
    res = det["res"]
    return RO.Ctg(name=name, length=length, ploidy=ploidy, seq=bytes(det["fasta"]), slots=slots, clusters=clusters,
                  sc_beg=list(sc.beg), sc_end=list(sc.end), sc_brk=[list(b) for b in sc.brk], sc_phase=res.sc_phase, pb_phase=det["pb"],
                  orig_dist=res.orig_phase_dist, swap_dist=res.swap_phase_dist, sc_phase_set=[0] * sc.n,
                  switches=list(det["switches"]), flips=list(det["flips"]), res=res)


def known_answer():
    """rows of tests/golden/demo/output.txt: {(type, threshold): (truth_tp, query_tp, truth_fn, query_fp)}"""
    out = {}
    for line in open(os.path.join(DEMO, "output.txt")):
        f = line.split()
        if len(f) >= 9 and f[0] in ("SNP", "INDEL", "SV", "ALL"):
            out[(f[0], f[1])] = tuple(int(x) for x in f[5:9])
    return out


if __name__ == "__main__":
    import sys
    rows, det = run(product=len(sys.argv) > 1 and sys.argv[1] == "gpu")
    print({k: det[k] for k in ("query_stats", "truth_stats", "n_var", "n_clusters", "n_sc")})
    ka = known_answer()
    for r in rows:
        key = (S.NAMES[r.vartype], "BEST" if r.best else "NONE")
        print(key, "Q>=%d" % r.qual, (r.truth_tp, r.query_tp, r.truth_fn, r.query_fp), "expected", ka.get(key),
              "prec %.6f recall %.6f f1 %.6f q %.6f" % (r.precision, r.recall, r.f1_score, r.f1_qscore))
