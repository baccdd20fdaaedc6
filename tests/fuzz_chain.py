"""Differential fuzzing of the stages in front of the path (test infrastructure): random variants -> biWFA clustering
on the GPU -> superclustering -> marshalling -> HIP path, every stage against its oracle, over random seeds, cluster
parameters and supercluster size limits (small limits force the splitting code).

    python tests/fuzz_chain.py [seconds] [first_seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def one(seed):
    import oracle_lib as O
    import test_gpu_parity as T
    from vcfdist_amd import _abi as A, api, cluster as K
    rng = np.random.default_rng(seed)
    kw = dict(n_sc=int(rng.integers(100, 1200)), len_mode=1, len_a=float(rng.uniform(10, 60)), len_b=float(rng.uniform(0.5, 1.2)),
              len_min=8, len_max=int(rng.integers(100, 800)), seed=seed, p_repeat=float(rng.uniform(0.0, 0.9)),
              var_per_base=float(rng.uniform(0.004, 0.05)), p_snp=float(rng.uniform(0.3, 0.95)),
              indel_mean=float(rng.uniform(1, 10)))
    sub, opn, ext = int(rng.integers(3, 8)), int(rng.integers(4, 9)), int(rng.integers(1, 4))
    itrs = int(rng.integers(1, 6))
    limit = int(rng.choice([300, 1000, 2000, 10000]))
    mode = int(rng.integers(0, 3))          # 0: biWFA, 1: gap clustering, 2: size clustering
    v = api.Synth(**kw).variants()
    ctg = bytes(v.ctg_seq)
    haps, cl = [], []
    for i in range(4):
        pool = v.allele_pool[i]
        refs = [bytes(pool[o:o + n]) for o, n in zip(v.var_ref_off[i], v.var_ref_len[i])]
        alts = [bytes(pool[o:o + n]) for o, n in zip(v.var_alt_off[i], v.var_alt_len[i])]
        h = K.HapSeq(v.var_pos[i], v.var_type[i], refs, alts)
        if mode == 0:
            got, _ = K.wfa_cluster(h, ctg, sub=sub, open=opn, extend=ext, max_cluster_itrs=itrs)
            want, _ = K.wfa_cluster(h, ctg, sub=sub, open=opn, extend=ext, max_cluster_itrs=itrs, L=O.lib(), prefix="vco")
        else:
            gap = int(rng.integers(5, 80))
            got = K.simple_cluster(h, mode - 1, gap, 10)
            want = K.simple_cluster(h, mode - 1, gap, 10, L=O.lib(), prefix="vco")
        assert got == want, ("clusters", i)
        haps.append(h)
        cl.append(got)
    s = K.supercluster(haps, cl, limit)
    assert s == K.supercluster(haps, cl, limit, L=O.lib(), prefix="vco"), "superclusters"
    if s.n == 0:
        return 0, 0, 0
    v2 = A.Variants(v.ctg_off, v.ctg_seq, np.zeros(s.n, np.int32), s.beg, s.end, [s.var_off(i) for i in range(4)],
                    v.var_pos, v.var_type, v.var_qual, v.var_ref_off, v.var_ref_len, v.var_alt_off, v.var_alt_len, v.allele_pool)
    if int(s.beg.min()) < 0:
        # a variant on the first base of the contig: the region starts at -1, where the reference exits (its substr throws,
        # dist.cpp:232-238); product and oracle must both refuse the batch.  (A region that reaches behind the contig's END is
        # cut at the last base by both and evaluated: include/vcfdist_pr.h, vpr_batch_from_variants.)
        for f, exc in ((api.batch_from_variants, api.VprError), (O.generate, ValueError)):
            try:
                f(v2)
            except exc:
                continue
            raise AssertionError(f"{f.__name__} accepted a region that starts in front of the contig")
        return sum(c.n for c in cl), 0, -1
    batch = api.batch_from_variants(v2)
    gen = O.generate(v2)
    for f in ("hap_seq", "hap_ptr", "hap_flag"):
        assert all(np.array_equal(getattr(batch, f)[h], getattr(gen, f)[h]) for h in range(4)), ("marshalling", f)
    _, _, ntie, _ = T.compare(batch, A.default_config(flags=int(os.environ.get("VCFDIST_FUZZ_FLAGS", "0"))))
    return sum(c.n for c in cl), s.n, s.n_oversize


if __name__ == "__main__":
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 9000
    t_end = time.time() + budget
    runs = ncl = nsc = nsplit = nrefused = 0
    while time.time() < t_end:
        try:
            a, b, c = one(seed)
        except Exception as e:      # any failure names its seed
            print(f"FAILED seed {seed}: {type(e).__name__}: {e}")
            sys.exit(1)
        runs += 1; ncl += a; nsc += b
        if c < 0:
            nrefused += 1
        else:
            nsplit += c
        seed += 1
    print(f"fuzz_chain: {runs} contigs, {ncl} clusters, {nsc} superclusters ({nsplit} oversize splits), "
          f"{nrefused} contigs refused on both sides (variant on a contig's first / last base), no mismatch")
