"""One fixed-length workload, a few executes (for rocprofv3 runs)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcfdist_amd import api, _abi as A
L = int(sys.argv[1]); n = int(sys.argv[2]); band = int(sys.argv[3]) if len(sys.argv) > 3 else 1
syn = api.Synth(n_sc=n, seed=5, len_mode=2, len_a=float(L), len_min=L, len_max=L)
b = syn.batch(copy=False)
pr = api.PrecisionRecall(A.default_config(band_mode=band))
pr.upload(b)
for _ in range(2):
    pr.execute()
t = pr.timing()
print(f"L={L} n_aln={4*n} rows={4*n*L} fwd {t.ms_fwd:.3f} bwd {t.ms_bwd:.3f} walk {t.ms_walk:.3f}")
