"""Where do the alignments of the SV workload (bench.py --workload sv_synth) end up?  Run with VPR_DEBUG=1 on a GPU box:
the library prints, per final window level, the alignments, their dense cells and their distances (DESIGN.md section 8
item 1 quotes this); the script adds the execute times and the kernels by time.   usage: VPR_DEBUG=1 python tools/sv_levels.py [n_sc]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api, _abi as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
b = api.Synth(n_sc=n, seed=0x5eed, len_mode=0, len_a=2000.0, len_b=12000.0, len_min=2000, len_max=12000,
              var_per_base=0.002, p_snp=0.7, indel_mean=600.0).batch()
pr = api.PrecisionRecall()
pr.upload(b)
for it in range(2):
    t0 = time.time(); pr.execute(); t1 = time.time()
    print("execute %.3f s" % (t1 - t0), flush=True)
r = pr.download()
t = pr.timing()
print("kernels %.1f ms fwd %.1f bwd %.1f walk %.1f | retries %d touched %.3e" % (t.ms_total, t.ms_fwd, t.ms_bwd, t.ms_walk, t.n_band_retries, t.cells_touched))
d = r.aln_dist.reshape(-1)
print("aln_dist quantiles", np.quantile(d, [0, .25, .5, .75, .9, .99, 1]))
agg = {}
for s in pr.launch_stats():
    k = s.kernel.decode()
    a = agg.setdefault(k, [0, 0.0, 0]); a[0] += 1; a[1] += s.ms; a[2] += s.n_units
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:14]:
    print("  %-28s launches %3d  ms %9.2f  units %d" % (k, a[0], a[1], a[2]))
