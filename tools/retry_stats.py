import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api, _abi as A
syn = api.Synth(n_sc=int(os.environ.get("NSC", "200000")), seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
b = syn.batch(copy=False)
L = np.diff(b.ref_off)
pr = api.PrecisionRecall(); pr.upload(b); pr.execute(); t = pr.timing()
res = pr.download()
print("stripe" if not os.environ.get("VPR_NO_STRIPE") else "ring", "retries", t.n_band_retries, "touched %.3e" % t.cells_touched,
      "fwd %.2f bwd %.2f walk %.2f total %.2f" % (t.ms_fwd, t.ms_bwd, t.ms_walk, t.ms_total))
for s in pr.launch_stats():
    print("  kind %d C=%d n=%d ms=%.3f" % (s.kind, s.cells_per_thread, s.n_units, s.ms))
d = res.aln_dist.reshape(-1, 4).max(axis=1)
big = np.argsort(L)[-8:]
print("largest L:", L[big].tolist(), "max s:", d[big].tolist())
