"""Per-row latency of the long-alignment kernels: a handful of superclusters of one fixed length, per-kernel times."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api
L = int(os.environ.get("LEN", "8000")); n = int(os.environ.get("NSC", "4"))
syn = api.Synth(n_sc=n, seed=0x5eed, len_mode=0, len_a=float(L), len_b=float(L), len_min=L, len_max=L)
b = syn.batch(copy=False)
pr = api.PrecisionRecall(); pr.upload(b)
for it in range(2):
    pr.execute()
t = pr.timing()
print("L", L, "n_sc", n, "total %.3f ms" % t.ms_total)
rows = int(np.diff(b.hap_off[2]).max())
for s in pr.launch_stats():
    nm = s.kernel.decode() if isinstance(s.kernel, bytes) else s.kernel
    print("  %-16s kind %d n=%d ms=%.3f  ns/row(longest)=%.0f" % (nm, s.kind, s.n_units, s.ms, s.ms * 1e6 / rows))
res = pr.download()
print("dist", res.aln_dist[:8], "status", res.aln_status[:8])
