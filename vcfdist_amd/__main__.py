"""python -m vcfdist_amd <query.vcf[.gz]> <truth.vcf[.gz]> <ref.fasta[.gz]> [-b regions.bed] [options]

The reference's command line for the precision/recall evaluation (vcfdist v2.6.4, main.cpp / globals.cpp) on top of
the MI355X path: VCF / BED / FASTA readers (include/vcfdist_io.h), biWFA or distance clustering and superclustering
(include/vcfdist_cluster.h), the precision/recall alignment on the GPU (include/vcfdist_pr.h), phasing, counters and
the PRECISION-RECALL SUMMARY and the output tables (include/vcfdist_report.h: precision-recall*.tsv, phase-blocks.tsv,
superclusters.tsv, query.tsv, truth.tsv, summary.vcf under -p PREFIX; -n writes nothing).  Realignment and the --distance
metrics are not part of it."""
import argparse
import sys

import numpy as np

from . import _abi as A
from . import api, cluster as K, io as IO, report as RP, summary as S


def transfer_phase_sets(slots, clusters, sc):
    """superclusterData::transfer_phase_sets (cluster.cpp:186-330): one phase set per supercluster from the variants' PS"""
    first_pos, phase_set = None, 0
    for s in slots:
        nz = np.nonzero(s["phase_set"])[0]
        if len(nz) and (first_pos is None or s["pos"][nz[0]] < first_pos):
            first_pos, phase_set = int(s["pos"][nz[0]]), int(s["phase_set"][nz[0]])
    cur = [0, 0]          # query / truth phase set carried across superclusters
    out = np.zeros(sc.n, np.int32)
    voff = [sc.var_off(i) for i in range(4)]
    for k in range(sc.n):
        for i, s in enumerate(slots):
            cs = i >> 1
            for v in range(int(voff[i][k]), int(voff[i][k + 1])):
                ps = int(s["phase_set"][v])
                if ps and ps > cur[cs]:
                    phase_set = cur[cs] = ps
        out[k] = phase_set
    return out


def evaluate_contig(name, seq, slots, args, device=0):
    """slots: [Q1, Q2, T1, T2] column dicts of include/vcfdist_io.h.  -> int64 counters [2][4][3][nq], n_sc, and what the
    writers need: (clusters after splitting, superclusters, results, phase sets, pb_phase, switches, flips)"""
    haps = []
    for s in slots:
        h = K.HapSeq.__new__(K.HapSeq)
        K.Hap.__init__(h, s["pos"], s["rlen"], s["type"], s["ref_len"], s["alt_len"])
        h.ref_off, h.alt_off, h.pool = s["ref_off"], s["alt_off"], s["pool"]
        haps.append(h)
    if args.cluster == "biwfa":
        cl = [K.wfa_cluster(h, bytes(seq), sub=args.sub, open=args.open, extend=args.extend, max_cluster_itrs=args.max_iterations,
                            reach_min_gap=args.reach_min_gap, device=device)[0] for h in haps]
    else:
        cl = [K.simple_cluster(h, 1 if args.cluster == "size" else 0, args.cluster_gap, args.reach_min_gap) for h in haps]
    sc = K.supercluster(haps, cl, args.max_supercluster_size)
    nq = args.max_qual - args.min_qual + 1
    if sc.n == 0:
        z = np.zeros(0, np.int32)
        return np.zeros((2, 4, 3, nq), np.int64), 0, (sc.clusters, sc, A.Results(0, [len(h.pos) for h in haps]), z, z, z, z)
    v = A.Variants(np.array([0, len(seq)], np.int64), seq, np.zeros(sc.n, np.int32), sc.beg, sc.end,
                   [sc.var_off(i) for i in range(4)], [h.pos for h in haps], [h.type for h in haps],
                   [s["var_qual"] for s in slots], [h.ref_off for h in haps], [h.ref_len for h in haps],
                   [h.alt_off for h in haps], [h.alt_len for h in haps], [h.pool for h in haps])
    cfg = A.default_config(device=device)
    cfg.max_qual = float(args.max_qual); cfg.credit_threshold = args.credit_threshold; cfg.phase_threshold = args.phase_threshold
    pr = api.PrecisionRecall(cfg)
    res = pr.run(api.batch_from_variants(v))
    phase_sets = transfer_phase_sets(slots, cl, sc)
    pb, sw, fl = S.phase(res.sc_phase, phase_sets)
    cls = [S.var_class(h.type, h.ref_len, h.alt_len, args.sv_threshold) for h in haps]
    counts = S.pr_counts(pr, cls, pb, args.min_qual, args.max_qual)
    print(f"[vcfdist_amd] {name}: {sum(len(h.pos) for h in haps)} hap-variants, {sum(c.n for c in cl)} clusters, {sc.n} superclusters, "
          f"{len(sw)} switch / {len(fl)} flip errors", file=sys.stderr)
    return counts, sc.n, (sc.clusters, sc, res, phase_sets, pb, sw, fl)


def main(argv=None):
    ap = argparse.ArgumentParser(prog="python -m vcfdist_amd", description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("query"); ap.add_argument("truth"); ap.add_argument("fasta")
    ap.add_argument("-b", "--bed")
    ap.add_argument("-f", "--filter", default="", help="comma-separated FILTER ids to keep")
    ap.add_argument("-l", "--largest-variant", type=int, default=5000, dest="max_size")
    ap.add_argument("-mn", "--min-qual", type=int, default=0)
    ap.add_argument("-mx", "--max-qual", type=int, default=60)
    ap.add_argument("-c", "--cluster", nargs="+", default=["biwfa"], help="biwfa | gap N | size N")
    ap.add_argument("-i", "--max-iterations", type=int, default=4)
    ap.add_argument("-s", "--max-supercluster-size", type=int, default=10000)
    ap.add_argument("-x", "--mismatch-penalty", type=int, default=5, dest="sub")
    ap.add_argument("-o", "--gap-open-penalty", type=int, default=6, dest="open")
    ap.add_argument("-e", "--gap-extend-penalty", type=int, default=2, dest="extend")
    ap.add_argument("-ct", "--credit-threshold", type=float, default=0.7)
    ap.add_argument("-pt", "--phasing-threshold", type=float, default=0.6, dest="phase_threshold")
    ap.add_argument("-sv", "--sv-threshold", type=int, default=50)
    ap.add_argument("--reach-min-gap", type=int, default=10)
    ap.add_argument("-p", "--prefix", default="./", help="prefix of the output files")
    ap.add_argument("-n", "--no-output-files", action="store_true")
    ap.add_argument("--device", type=int, default=0)
    args = ap.parse_args(argv)
    args.cluster_gap = 50
    if args.cluster[0] in ("gap", "size") and len(args.cluster) > 1:
        args.cluster_gap = int(args.cluster[1])
    args.cluster = args.cluster[0]
    filters = tuple(f for f in args.filter.split(",") if f)
    bed = IO.Bed(args.bed) if args.bed else None
    kw = dict(min_qual=args.min_qual, max_qual=args.max_qual, max_size=args.max_size, cluster_min_gap=args.cluster_gap, filters=filters)
    q = IO.read_vcf(args.query, bed, **kw)
    t = IO.read_vcf(args.truth, bed, **kw)
    fasta = IO.read_fasta(args.fasta)
    contigs = list(q["contigs"]) + [c for c in t["contigs"] if c not in q["contigs"]]
    nq = args.max_qual - args.min_qual + 1
    total = np.zeros((2, 4, 3, nq), np.int64)
    empty = dict(pos=np.zeros(0, np.int32), rlen=np.zeros(0, np.int32), type=np.zeros(0, np.uint8), var_qual=np.zeros(0, np.float32),
                 phase_set=np.zeros(0, np.int32), ref_len=np.zeros(0, np.int32), alt_len=np.zeros(0, np.int32),
                 ref_off=np.zeros(0, np.int64), alt_off=np.zeros(0, np.int64), pool=np.zeros(1, np.uint8))
    reports = []
    for ctg in contigs:
        if ctg not in fasta:
            raise SystemExit(f"ERROR: contig '{ctg}' not in reference FASTA")
        qs = q["vars"][q["contigs"].index(ctg)] if ctg in q["contigs"] else [empty, empty]
        ts = t["vars"][t["contigs"].index(ctg)] if ctg in t["contigs"] else [empty, empty]
        try:
            counts, n_sc, tables = evaluate_contig(ctg, fasta[ctg], [qs[0], qs[1], ts[0], ts[1]], args, device=args.device)
        except api.VprError as e:     # the library's explicit refusals end the run like the reference's ERROR()
            raise SystemExit(f"ERROR: contig '{ctg}': {e}")
        total += counts
        if not args.no_output_files:
            src = q if ctg in q["contigs"] else t     # superclusterData ctor, cluster.cpp:134-157: query's header wins
            k = src["contigs"].index(ctg)
            reports.append(RP.Contig(ctg, src["lengths"][k], src["ploidy"][k], fasta[ctg], [qs[0], qs[1], ts[0], ts[1]], *tables))
    if not args.no_output_files:
        RP.write_precision_recall(args.prefix, total, args.min_qual, args.max_qual)
        cmd = " ".join(["vcfdist"] + list(sys.argv[1:] if argv is None else argv))
        RP.write_results(args.prefix, reports, cmd=cmd, credit_threshold=args.credit_threshold)
    rows = S.pr_summary(total, args.min_qual, args.max_qual)
    print("PRECISION-RECALL SUMMARY\n")
    print("TYPE\tTHRESHOLD\tTRUTH_TP\tQUERY_TP\tTRUTH_FN\tQUERY_FP\tPREC\t\tRECALL\t\tF1_SCORE\tF1_QSCORE")
    for r in rows:
        print("%s\t%s Q >= %-2d\t%-16d%-16d%-16d%-16d%f\t%f\t%f\t%f" % (S.NAMES[r.vartype], "BEST" if r.best else "NONE", r.qual,
              r.truth_tp, r.query_tp, r.truth_fn, r.query_fp, r.precision, r.recall, r.f1_score, r.f1_qscore))
        if r.best:
            print()
    return rows


if __name__ == "__main__":
    main()
