"""CPU-baseline calibration: the oracle's precision/recall stage (oracle/pr_oracle.cpp, one thread) on the
four demo workloads the survey timed the real reference on -- `-c biwfa`, `-c gap 50`, `-c gap 200`, `-c gap 1000` on
demo/query.vcf vs the NIST truth with the seeded surrogate FASTA; reference P/R stage (its own timer, main.cpp:218-221,
`-t 1`, the survey's 8-core Xeon @ 2.1 GHz = this container): 0.252 / 1.16 / 12.9 / 163.4 s.
usage: python tools/calibrate_cpu.py [biwfa gap50 gap200 gap1000]      (run where /root/repo/tests/golden/demo exists)"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import demo_pipeline as D
import oracle_lib as O
from vcfdist_amd import cluster as K, _abi as A

REF_S = {"biwfa": 0.252, "gap50": 1.16, "gap200": 12.9, "gap1000": 163.4}
REF_NSC = {"biwfa": 6058, "gap50": 4624, "gap200": 2484, "gap1000": 530}
out = {}
for name in (sys.argv[1:] or ["biwfa", "gap50", "gap200"]):
    cluster = "biwfa" if name == "biwfa" else ("gap", int(name[3:]))
    bed = D.Bed(os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.bed"))
    q, _ = D.parse_vcf(os.path.join(D.DEMO, "query.vcf"), bed)
    t, _ = D.parse_vcf(os.path.join(D.DEMO, "nist-v4.2.1_chr1_5Mb.vcf.gz"), bed)
    fasta = D.surrogate_fasta(5_100_000)
    slots = [q[0], q[1], t[0], t[1]]
    haps = [K.HapSeq(s["pos"], s["type"], s["ref"], s["alt"]) for s in slots]
    lib = O.lib()
    if cluster == "biwfa":
        cl = [K.wfa_cluster(h, bytes(fasta), sub=D.G["sub"], open=D.G["open"], extend=D.G["extend"], max_cluster_itrs=D.G["max_cluster_itrs"],
                            reach_min_gap=D.G["reach_min_gap"], L=lib, prefix="vco")[0] for h in haps]
    else:
        cl = [K.simple_cluster(h, 0, cluster[1], D.G["reach_min_gap"], L=lib, prefix="vco") for h in haps]
    sc = K.supercluster(haps, cl, D.G["max_supercluster_size"], L=lib, prefix="vco")
    pool, roff, aoff = [h.pool for h in haps], [h.ref_off for h in haps], [h.alt_off for h in haps]
    v = A.Variants(np.array([0, 5_100_000], np.int64), fasta, np.zeros(sc.n, np.int32), sc.beg, sc.end,
                   [sc.var_off(i) for i in range(4)], [h.pos for h in haps], [h.type for h in haps],
                   [np.asarray(s["qual"], np.float32) for s in slots], roff, [h.ref_len for h in haps], aoff,
                   [h.alt_len for h in haps], pool)
    t0 = time.perf_counter()
    batch = O.generate(v)          # generate_ptrs_strs is inside the reference's timed stage
    O.run(batch)
    dt = time.perf_counter() - t0
    Ls = np.maximum(np.diff(batch.ref_off), 1).astype(np.float64)
    out[name] = {"superclusters": int(sc.n), "reference_superclusters": REF_NSC[name], "oracle_s": round(dt, 3), "reference_s": REF_S[name],
                 "oracle_over_reference": round(dt / REF_S[name], 3),
                 # where this workload's time is spent: the span length weighted by the cost model L^2 (bench.py interpolates the
                 # ratio over octaves of the supercluster length between the four workloads' figures)
                 "cost_weighted_span": round(float((Ls ** 3).sum() / (Ls ** 2).sum()), 1), "median_span": float(np.median(Ls))}
    print(name, out[name], flush=True)
print(json.dumps(out))
if os.environ.get("CALIBRATION_OUT"):
    json.dump(out, open(os.environ["CALIBRATION_OUT"], "w"), indent=1)
