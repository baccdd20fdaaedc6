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
    """superclusterData::transfer_phase_sets (cluster.cpp:186-330): one phase set per supercluster from the variants' PS.
    The reference walks the superclusters, inside one the four haps (query 1, 2, truth 1, 2) and their variants, keeps one
    running maximum of PS per callset, and a variant whose PS exceeds its callset's maximum makes its PS the current phase
    set; a supercluster gets the phase set current at its end.  Only a variant whose PS exceeds every PS in front of it on
    its own hap can do that, so the walk is done over those few candidates (numpy finds them) instead of all variants."""
    first_pos, phase_set = None, 0
    for s in slots:
        nz = np.nonzero(s["phase_set"])[0]
        if len(nz) and (first_pos is None or s["pos"][nz[0]] < first_pos):
            first_pos, phase_set = int(s["pos"][nz[0]]), int(s["phase_set"][nz[0]])
    cand = []
    for i, s in enumerate(slots):
        off = sc.var_off(i)
        n = int(off[-1]) if len(off) else 0
        if n == 0:
            continue
        ps = np.asarray(s["phase_set"][:n], dtype=np.int64)
        run = np.maximum.accumulate(ps)
        idx = np.nonzero((ps != 0) & (ps > np.concatenate(([0], run[:-1]))))[0]
        sc_of = np.searchsorted(off, idx, side="right") - 1
        cand += [(int(k), i, int(v), int(ps[v])) for v, k in zip(idx, sc_of)]
    cand.sort()                                   # processing order: supercluster, hap slot, variant
    cur = [0, 0]
    ev_k, ev_v = [], []
    for k, i, v, ps in cand:
        if ps > cur[i >> 1]:
            cur[i >> 1] = ps
            ev_k.append(k); ev_v.append(ps)
    out = np.full(sc.n, phase_set, np.int32)
    if ev_k:
        ek, ev = np.asarray(ev_k), np.asarray(ev_v, dtype=np.int32)
        last = np.searchsorted(ek, np.arange(sc.n), side="right") - 1      # last update at or before each supercluster
        out = np.where(last >= 0, ev[np.maximum(last, 0)], phase_set).astype(np.int32)
    return out


WARN_TEXT = (       # dist.cpp:1203-1223, raised per alignment as VPR_ST_WARN_* bits
    (A.ST_WARN_REF_ED, "Nonzero reference edit distance with no truth variants at ctg %s supercluster %d"),
    (A.ST_WARN_QUERY_ED, "Query edit distance changed with no query variants at ctg %s supercluster %d"),
    (A.ST_WARN_EXCEEDS, "Query edit distance exceeds reference edit distance at ctg %s supercluster %d"),
    (A.ST_WARN_ZERO_ED, "Zero edit distance with truth variants at ctg %s supercluster %d"),
)


ERR_TEXT = ((A.ST_ERR_LIMIT, "exceeds an implementation limit of the GPU path (more than eight swap sources on one position, or an alignment "
                             "the dense kernels cannot place)"),
            (A.ST_ERR_NO_PTR, "ended in a walk without a path pointer"), (A.ST_ERR_UNFINISHED, "was left unfinished"))


def print_warnings(name, aln_status, strict=False):
    """the reference's WARN lines for the conditions calc_prec_recall flags (one line per alignment and condition), and -- LOUDLY --
    the alignments this implementation did NOT evaluate (VPR_ST_ERR_*: the reference has no such limits and evaluates them).
    Their variants stay ERRTYPE_UN and are missing from the counts; their superclusters' phasing is taken from what was
    evaluated.  -> number of superclusters with such an alignment; with strict the run ends instead (--strict)."""
    for bit, text in WARN_TEXT:
        for a in np.nonzero(aln_status & np.uint32(bit))[0]:
            print("[WARN  vcfdist] " + text % (name, int(a) // 4), file=sys.stderr)
    bad_sc = set()
    for bit, text in ERR_TEXT:
        idx = np.nonzero(aln_status & np.uint32(bit))[0]
        if len(idx) == 0:
            continue
        scs = sorted({int(a) // 4 for a in idx})
        bad_sc.update(scs)
        print(f"[WARN  vcfdist_amd] contig '{name}': {len(idx)} alignment(s) in {len(scs)} supercluster(s) {text}: NOT EVALUATED -- their variants are "
              f"left out of every count and table (superclusters {', '.join(map(str, scs[:12]))}{' ...' if len(scs) > 12 else ''})", file=sys.stderr)
    if bad_sc and strict:
        raise SystemExit(f"ERROR: contig '{name}': {len(bad_sc)} supercluster(s) not evaluated (--strict)")
    return len(bad_sc)


def prepare_contig(name, seq, slots, args, device=0):
    """everything in front of the precision/recall path for one contig: clustering, superclustering and the host
    marshalling (generate_ptrs_strs).  A contig the library cannot marshal (vpr_batch_from_variants, include/vcfdist_pr.h)
    raises here, before any contig has been evaluated."""
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
    batch = None
    if sc.n:
        v = A.Variants(np.array([0, len(seq)], np.int64), seq, np.zeros(sc.n, np.int32), sc.beg, sc.end,
                       [sc.var_off(i) for i in range(4)], [h.pos for h in haps], [h.type for h in haps],
                       [s["var_qual"] for s in slots], [h.ref_off for h in haps], [h.ref_len for h in haps],
                       [h.alt_off for h in haps], [h.alt_len for h in haps], [h.pool for h in haps])
        batch = api.batch_from_variants(v)
    return dict(name=name, haps=haps, cl=cl, sc=sc, batch=batch, slots=slots)


def evaluate_contig(prep, args, device=0, part=None):
    """the precision/recall path on the GPU, phasing and counters for a prepared contig.  -> int64 counters [2][4][3][nq],
    n_sc, and what the writers need: (clusters after splitting, superclusters, results, phase sets, pb_phase, switches, flips).
    part = (rank, world, collective device): this rank evaluates its share of the contig's SUPERCLUSTERS -- dealt by the
    reference's own size estimate (shard.deal: LPT + snake, as precision_recall_threads_wrapper spreads superclusters over its
    threads, dist.cpp:1670-1726) -- the per-supercluster phasing is all-gathered (the contig's phasing needs all of it and
    every rank then runs it), the counters returned are this rank's share (the caller all-reduces the sum over the contigs),
    and the result records are gathered (every rank gets the contig's full tables; rank 0 writes them)."""
    name, haps, cl, sc, slots = prep["name"], prep["haps"], prep["cl"], prep["sc"], prep["slots"]
    nq = args.max_qual - args.min_qual + 1
    if sc.n == 0:
        z = np.zeros(0, np.int32)
        return np.zeros((2, 4, 3, nq), np.int64), 0, (sc.clusters, sc, A.Results(0, [len(h.pos) for h in haps]), z, z, z, z)
    cfg = A.default_config(device=device)
    cfg.max_qual = float(args.max_qual); cfg.credit_threshold = args.credit_threshold; cfg.phase_threshold = args.phase_threshold
    pr = api.PrecisionRecall(cfg)
    phase_sets = transfer_phase_sets(slots, cl, sc)
    cls = [S.var_class(h.type, h.ref_len, h.alt_len, args.sv_threshold) for h in haps]
    def mask_unevaluated(r):
        """a supercluster with an alignment the GPU path did not evaluate (VPR_ST_ERR_*): its phase distances come from alignments
        that were cut short -- it takes no side in the contig's phasing (print_warnings reports it)"""
        bad = np.unique(np.nonzero(r.aln_status & np.uint32(A.ST_ERR_LIMIT | A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED))[0] // 4)
        if len(bad):
            r.sc_phase[bad] = A.PHASE_NONE
    if part is None:
        res = pr.run(prep["batch"])
        mask_unevaluated(res)
        pb, sw, fl = S.phase(res.sc_phase, phase_sets)
        counts = S.pr_counts(pr, cls, pb, args.min_qual, args.max_qual)
    else:
        from . import shard
        rank, world, cdev = part
        whole = prep["batch"]
        idx = shard.deal(shard.estimate_cells(whole), world)[rank]
        mine = whole.subset(idx)
        # (every rank enters the collectives below: a rank whose share fails tells the others first, and all end together --
        # a lone SystemExit would leave the rest waiting in all_gather for ever)
        err = None
        try:
            local = pr.run(mine) if len(idx) else A.Results(0, [0, 0, 0, 0])
        except Exception as e:      # noqa: BLE001 -- whatever it is (VprError, MemoryError, OSError ...), the other ranks must hear of it
            err, local = e, None
        if shard.any_rank(err is not None, device=cdev):
            raise api.VprError(str(err) if err is not None else "another rank's share of the contig failed")
        mask_unevaluated(local)
        sc_phase, _, _ = shard.allgather_phase(local, idx, sc.n, device=cdev)
        pb, sw, fl = S.phase(sc_phase, phase_sets)
        if len(idx):
            cls_mine = [shard.subset_per_variant(cls[s], whole.var_off[s], idx) for s in range(4)]
            counts = S.pr_counts(pr, cls_mine, pb[idx], args.min_qual, args.max_qual)
        else:
            counts = np.zeros((2, 4, 3, nq), np.int64)
        res = shard.gather_results(local, idx, whole.var_off, device=cdev)
        if rank != 0:
            # (--strict ends the run on EVERY rank: the gathered status column is the same everywhere, rank 0 prints the report
            # below and the others leave with it instead of waiting in the next contig's collective -- ADVICE r5)
            if getattr(args, "strict", False) and (res.aln_status & np.uint32(A.ST_ERR_LIMIT | A.ST_ERR_NO_PTR | A.ST_ERR_UNFINISHED)).any():
                raise SystemExit(1)
            return counts, sc.n, (sc.clusters, sc, res, phase_sets, pb, sw, fl)
    print_warnings(name, res.aln_status, strict=getattr(args, "strict", False))
    print(f"[vcfdist_amd] {name}: {sum(len(h.pos) for h in haps)} hap-variants, {sum(c.n for c in cl)} clusters, {sc.n} superclusters, "
          f"{len(sw)} switch / {len(fl)} flip errors", file=sys.stderr)
    return counts, sc.n, (sc.clusters, sc, res, phase_sets, pb, sw, fl)


def check_contigs(q, t, fasta, bed):
    """check_contigs (bed.cpp:135-284) -> the contigs to evaluate, in the order the reference's superclusterData walks them
    (the query's list after the check: its VCF order, then what the check appends).  With a BED only its contigs count
    (others are dropped from both callsets and need not be in the FASTA); without one every truth contig has to be in the
    FASTA."""
    warn = lambda m: print("[WARN  vcfdist] " + m, file=sys.stderr)
    qc, tc = list(q["contigs"]), list(t["contigs"])
    if bed is not None:
        qc = [c for c in qc if c in bed.contigs]
        tc = [c for c in tc if c in bed.contigs]
        for c in qc:
            if c not in tc:
                warn(f"Contig '{c}' found in query VCF but not truth VCF.")
        for c in tc:
            if c not in qc:
                warn(f"Contig '{c}' found in truth VCF but not query VCF.")
        for c in bed.contigs:
            if c not in fasta:
                raise SystemExit(f"ERROR: Contig '{c}' found in BED but not reference FASTA.")
            if c not in qc:
                qc.append(c)
        return qc
    for c in tc:
        if c not in fasta:
            raise SystemExit(f"ERROR: Contig '{c}' found in truth VCF but not reference FASTA. Please provide BED file.")
        if c not in qc:
            warn(f"Contig '{c}' found in truth VCF but not query VCF. All truth variants on '{c}' will be false negatives.")
            qc.append(c)
    for c in qc:
        if c not in tc:
            warn(f"Contig '{c}' found in query VCF but not truth VCF. All query variants on '{c}' will be false positives.")
            if c not in fasta:
                raise SystemExit(f"ERROR: contig '{c}' not in reference FASTA")
    return qc


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
    ap.add_argument("--strict", action="store_true",
                    help="end with an error when a supercluster exceeds an implementation limit of the GPU path (more than eight swap "
                         "sources on one position, an alignment the dense kernels cannot place) instead of warning and leaving its "
                         "variants out of the counts and tables")
    ap.add_argument("--device", type=int, default=None, help="HIP device (default: LOCAL_RANK, else 0)")
    ap.add_argument("--shard", default="superclusters", choices=["superclusters", "contigs"],
                    help="several ranks (torch.distributed.run, one per GPU): deal every contig's superclusters over the ranks "
                         "(default; balanced whatever the contigs' sizes) or whole contigs")
    args = ap.parse_args(argv)
    args.cluster_gap = 50
    if args.cluster[0] in ("gap", "size") and len(args.cluster) > 1:
        args.cluster_gap = int(args.cluster[1])
    args.cluster = args.cluster[0]
    if args.max_size + 2 > args.max_supercluster_size:          # globals.cpp:478-481
        raise SystemExit("ERROR: Max supercluster size (-s) must be at least two larger than max variant size (-l).")
    # one process per GPU under torch.distributed.run: every contig's superclusters are dealt over the ranks (--shard; the
    # per-supercluster phasing is all-gathered, a contig's phasing needs all of it), the counters are summed with one all-reduce
    # (RCCL), the result records are gathered, and rank 0 writes the files
    import os
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    device = args.device if args.device is not None else int(os.environ.get("LOCAL_RANK", "0"))
    one_gpu = bool(os.environ.get("VCFDIST_ONE_GPU"))      # plumbing check on a one-GPU box: every rank uses device 0, gloo
    if one_gpu and args.device is None:
        device = 0
    try:      # this process's GPU, before anything allocates page-locked staging memory (it would otherwise initialise device 0)
        api.lib().vpr_select_device(int(device))
    except api.VprError:
        pass
    dist = None
    if world > 1:
        import torch
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = "nccl" if not one_gpu and torch.cuda.is_available() and torch.cuda.device_count() > device else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(device)
        dist.init_process_group(backend=backend)
    filters = tuple(f for f in args.filter.split(",") if f)
    bed = IO.Bed(args.bed) if args.bed else None
    kw = dict(min_qual=args.min_qual, max_qual=args.max_qual, max_size=args.max_size, cluster_min_gap=args.cluster_gap, filters=filters)
    q = IO.read_vcf(args.query, bed, **kw)
    t = IO.read_vcf(args.truth, bed, **kw)
    fasta = IO.read_fasta(args.fasta)
    contigs = check_contigs(q, t, fasta, bed)
    nq = args.max_qual - args.min_qual + 1
    total = np.zeros((2, 4, 3, nq), np.int64)
    empty = dict(pos=np.zeros(0, np.int32), rlen=np.zeros(0, np.int32), type=np.zeros(0, np.uint8), var_qual=np.zeros(0, np.float32),
                 phase_set=np.zeros(0, np.int32), ref_len=np.zeros(0, np.int32), alt_len=np.zeros(0, np.int32),
                 ref_off=np.zeros(0, np.int64), alt_off=np.zeros(0, np.int64), pool=np.zeros(1, np.uint8))

    def slots_of(ctg):
        qs = q["vars"][q["contigs"].index(ctg)] if ctg in q["contigs"] else [empty, empty]
        ts = t["vars"][t["contigs"].index(ctg)] if ctg in t["contigs"] else [empty, empty]
        return [qs[0], qs[1], ts[0], ts[1]]

    from . import shard
    weights = [sum(len(s["pos"]) for s in slots_of(c)) + 1 for c in contigs]
    by_sc = dist is not None and args.shard == "superclusters"
    # by superclusters: every rank clusters and marshals every contig (host work and the biWFA clustering, done redundantly)
    # and evaluates its share of each; by contigs: a rank does everything for the contigs dealt to it
    mine = list(range(len(contigs))) if by_sc else shard.deal_contigs(weights, world)[rank]
    cdev = None if dist is None or dist.get_backend() == "gloo" else f"cuda:{device}"
    # everything in front of the path for all of this rank's contigs first: input the library refuses ends the run here,
    # like the reference's ERROR(), before anything has been evaluated or written
    prepared = {}
    for k in mine:
        ctg = contigs[k]
        try:
            prepared[k] = prepare_contig(ctg, fasta[ctg], slots_of(ctg), args, device=device)
        except api.VprError as e:
            raise SystemExit(f"ERROR: contig '{ctg}': {e}")
    reports = {}
    for k in mine:
        ctg = contigs[k]
        try:
            counts, n_sc, tables = evaluate_contig(prepared.pop(k), args, device=device, part=(rank, world, cdev) if by_sc else None)
        except api.VprError as e:     # the library's explicit refusals end the run like the reference's ERROR()
            raise SystemExit(f"ERROR: contig '{ctg}': {e}")
        total += counts
        if not args.no_output_files:
            src = q if ctg in q["contigs"] else t     # superclusterData ctor, cluster.cpp:134-157: query's header wins
            if ctg in src["contigs"]:
                j = src["contigs"].index(ctg)
                length, ploidy = src["lengths"][j], src["ploidy"][j]
            else:                                     # a BED contig neither callset lists (check_contigs appends it, ploidy 0)
                length, ploidy = len(fasta[ctg]), 0
            reports[k] = (ctg, length, ploidy, slots_of(ctg), tables)
    if dist is not None:
        total = shard.allreduce_tally(total, device=cdev)       # the one all-reduce: counts[2][4][3][nq] summed over the ranks
        if not by_sc:       # (by superclusters every rank already holds every contig's gathered tables)
            gathered = [None] * world
            dist.all_gather_object(gathered, reports)
            reports = {k: v for part in gathered for k, v in part.items()}
    rows = S.pr_summary(total, args.min_qual, args.max_qual)
    if rank == 0:
        if not args.no_output_files:
            RP.write_precision_recall(args.prefix, total, args.min_qual, args.max_qual)
            cmd = " ".join(["vcfdist"] + list(sys.argv[1:] if argv is None else argv))
            RP.write_parameters(args.prefix, args, cmd)
            ctgs = [RP.Contig(c, ln, pl, fasta[c], sl, *tb) for c, ln, pl, sl, tb in (reports[k] for k in sorted(reports))]
            RP.write_results(args.prefix, ctgs, cmd=cmd, credit_threshold=args.credit_threshold)
        print("PRECISION-RECALL SUMMARY\n")
        print("TYPE\tTHRESHOLD\tTRUTH_TP\tQUERY_TP\tTRUTH_FN\tQUERY_FP\tPREC\t\tRECALL\t\tF1_SCORE\tF1_QSCORE")
        for r in rows:
            print("%s\t%s Q >= %-2d\t%-16d%-16d%-16d%-16d%f\t%f\t%f\t%f" % (S.NAMES[r.vartype], "BEST" if r.best else "NONE", r.qual,
                  r.truth_tp, r.query_tp, r.truth_fn, r.query_fp, r.precision, r.recall, r.f1_score, r.f1_qscore))
            if r.best:
                print()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    return rows


if __name__ == "__main__":
    main()
