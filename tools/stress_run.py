"""configs[4]-style stress sample (log-uniform 32..16384) on one GPU: timing + self-consistency."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api, _abi as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 200
syn = api.Synth(n_sc=n, seed=0x5eed, len_mode=0, len_a=32.0, len_b=16384.0, len_min=32, len_max=16384)
b = syn.batch(copy=False)
L = np.diff(b.ref_off)
print("n_sc", n, "dense cells %.3e" % b.dense_cells(), "L max", L.max())
pr = api.PrecisionRecall()
t0 = time.time(); pr.upload(b); t1 = time.time()
pr.execute(); t2 = time.time()
res = pr.download(); t3 = time.time()
t = pr.timing()
print("upload %.2fs execute %.2fs download %.2fs | kernels %.1f ms fwd %.1f bwd %.1f walk %.1f ed %.1f | touched %.3e retries %d"
      % (t1 - t0, t2 - t1, t3 - t2, t.ms_total, t.ms_fwd, t.ms_bwd, t.ms_walk, t.ms_ed, t.cells_touched, t.n_band_retries))
for s in pr.launch_stats():
    if s.ms > 5: print("  kind %d nt=%d C=%d n=%d ms=%.2f" % (s.kind, s.threads, s.cells_per_thread, s.n_units, s.ms))
print("status bits:", np.unique(res.aln_status, return_counts=True))
print("aln/s %.0f  dense cells/s %.3e" % (4 * n / (t.ms_total * 1e-3), b.dense_cells() / (t.ms_total * 1e-3)))
# band_mode 0 cross-check on a subset that the dense kernels can hold
idx = np.nonzero(L <= 3000)[0][:40]
sub = b.subset(idx)
r1 = api.PrecisionRecall().run(sub)
r0 = api.PrecisionRecall(A.default_config(band_mode=0)).run(sub)
print("band vs dense on %d superclusters: %s" % (len(idx), "identical" if not r1.diff(r0) else r1.diff(r0)[:3]))
