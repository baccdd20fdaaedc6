"""GPU-vs-oracle debugging run (used under gpurun while bringing kernels up)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
from vcfdist_amd import api, _abi as A
import oracle_lib as O

def check(name, **kw):
    syn = api.Synth(**kw)
    b = syn.batch()
    t0 = time.time()
    ex = O.Extra(b)
    want = O.run(b, extra=ex)
    t1 = time.time()
    pr = api.PrecisionRecall(A.default_config(band_mode=int(os.environ.get("BAND", "1"))))
    got = pr.run(b)
    t2 = time.time()
    tm = pr.timing()
    nonmax = ex.swap_used_conflict_nonmax.reshape(-1, 4).sum(axis=1) > 0
    bad = got.diff(want)
    print(f"== {name}: n_sc={b.n_sc} cells={b.dense_cells():.3e} oracle {t1-t0:.2f}s gpu-total {t2-t1:.2f}s "
          f"kernels {tm.ms_total:.2f}ms (fwd {tm.ms_fwd:.2f} bwd {tm.ms_bwd:.2f} walk {tm.ms_walk:.2f} ed {tm.ms_ed:.2f}) "
          f"ties(nonmax sc)={int(nonmax.sum())} touched={tm.cells_touched:.3e} retries={tm.n_band_retries}")
    if bad:
        for line in bad[:12]:
            print("   ", line)
        # per-alignment scalar mismatches
        for f in ("aln_dist", "aln_end_plane", "aln_beg_plane", "aln_status"):
            a, bb = getattr(got, f), getattr(want, f)
            idx = np.nonzero(a != bb)[0]
            if len(idx):
                scs = np.unique(idx // 4)
                clean = [s for s in scs if not nonmax[s]]
                print(f"    {f}: {len(idx)} aln mismatches in {len(scs)} sc, {len(clean)} sc without nonmax tie; first clean sc: {clean[:5]}")
                for s in clean[:2]:
                    print("      lens", b.lens(s), "got", a[s*4:s*4+4], "want", bb[s*4:s*4+4])
        # per-variant mismatches restricted to clean superclusters
        nclean = 0
        for h in range(4):
            sc_of_var = np.repeat(np.arange(b.n_sc), np.diff(b.var_off[h]))
            for w in range(2):
                m = np.zeros(len(sc_of_var), bool)
                for name2, dt in A.Results.PER_VAR:
                    x, y = getattr(got, name2)[h][w], getattr(want, name2)[h][w]
                    if dt == np.float32: x, y = x.view(np.uint32), y.view(np.uint32)
                    m |= (x != y)
                m &= ~nonmax[sc_of_var]
                nclean += int(m.sum())
                if m.any():
                    v = np.nonzero(m)[0][0]
                    print(f"    clean var mismatch hap {h} swap {w} var {v} sc {sc_of_var[v]}: "
                          + " ".join(f"{n2}={getattr(got,n2)[h][w][v]}/{getattr(want,n2)[h][w][v]}" for n2,_ in A.Results.PER_VAR))
        print(f"    per-variant mismatches outside nonmax-tie superclusters: {nclean}")
    else:
        print("    bit-exact")
    return bad

if __name__ == "__main__":
    print(api.lib().vpr_version())
    check("tiny", n_sc=200, len_a=6, len_b=60, len_min=5, len_max=60, seed=1, var_per_base=0.08, p_snp=0.5, p_repeat=0.5)
    check("c64x4", n_sc=200, len_a=65, len_b=250, len_min=65, len_max=250, seed=2, var_per_base=0.03)
    check("c256x4", n_sc=60, len_a=260, len_b=1000, len_min=260, len_max=1000, seed=3, var_per_base=0.01)
    check("c256x8", n_sc=30, len_a=1030, len_b=2000, len_min=1030, len_max=2000, seed=4)
    check("c1024x8", n_sc=12, len_a=2100, len_b=5000, len_min=2100, len_max=5000, seed=5)
    check("mixed", n_sc=300, len_a=8, len_b=3000, len_min=5, len_max=3000, seed=6)
