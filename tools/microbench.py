"""Per-kernel timings on fixed-length synthetic workloads (kernel bring-up / tuning aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcfdist_amd import api, _abi as A

def run(L, n_sc, band=1, reps=3):
    syn = api.Synth(n_sc=n_sc, seed=5, len_mode=2, len_a=float(L), len_min=L, len_max=L)
    b = syn.batch(copy=False)
    pr = api.PrecisionRecall(A.default_config(band_mode=band))
    pr.upload(b)
    best = None
    for _ in range(reps):
        pr.execute()
        t = pr.timing()
        cur = (t.ms_fwd, t.ms_bwd, t.ms_walk, t.ms_ed, t.ms_total)
        best = cur if best is None or cur[4] < best[4] else best
    n_aln = 4 * n_sc
    rows = n_aln * L
    print(f"L={L:6d} n_aln={n_aln:8d} band={band}: fwd {best[0]:8.3f} bwd {best[1]:8.3f} walk {best[2]:8.3f} ed {best[3]:6.3f} total {best[4]:8.3f} ms"
          f" | fwd {best[0]*1e6/rows:7.2f} ns/row-aln, {best[0]*1e3/n_aln:8.3f} us/aln | retries {t.n_band_retries}")

if __name__ == "__main__":
    for L, n in [(8, 200000), (20, 200000), (20, 20000), (60, 100000), (200, 30000), (1000, 4000), (1000, 64), (4000, 16), (4000, 1000)]:
        run(L, n)
