"""Where do the alignments of the joint workload (bench.py --workload joint_synth, BASELINE configs[3]) end up?  Run with VPR_DEBUG=1 on a
GPU box: the library prints, per final window level, the alignments, their dense cells and their distances; the script adds the execute
times, the launches by time and the shapes (Lq, Lr, Lt, distance) of the alignments with a large distance.
usage: VPR_DEBUG=1 python tools/joint_levels.py [n_sc]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api, _abi as A
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100000
b = api.Synth(n_sc=n, seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10002, p_sv=0.0075, sv_min=50, sv_max=10000).batch()
pr = api.PrecisionRecall()
pr.upload(b)
for it in range(3):
    t0 = time.time(); pr.execute(); t1 = time.time()
    print("execute %.3f s" % (t1 - t0), flush=True)
r = pr.download()
t = pr.timing()
print("kernels %.1f ms fwd %.1f bwd %.1f walk %.1f ed %.1f tie %.1f | retries %d touched %.3e" % (t.ms_total, t.ms_fwd, t.ms_bwd, t.ms_walk, t.ms_ed, t.ms_tie, t.n_band_retries, t.cells_touched))
d = r.aln_dist.reshape(-1, 4)
hl = [np.diff(b.hap_off[h]) for h in range(4)]
Lr = np.diff(b.ref_off)
big = np.argwhere(d >= 48)
print("alignments with distance >= 48:", len(big), " status bits:", np.bincount(r.aln_status)[:8])
rows = []
for sc, a in big:
    lq, lt = int(hl[a >> 1][sc]), int(hl[2 + (a & 1)][sc])
    rows.append((lq, int(Lr[sc]), lt, int(d[sc, a]), (lq + int(Lr[sc])) * lt))
rows.sort(key=lambda x: -x[4])
print("largest matrices among them (Lq, Lr, Lt, dist, cells):")
for x in rows[:25]:
    print("   ", x)
print("sum of their cells %.3e; all alignments' dense cells %.3e" % (sum(x[4] for x in rows), b.dense_cells()))
agg = {}
for s in pr.launch_stats():
    k = s.kernel.decode()
    a = agg.setdefault(k, [0, 0.0, 0, 0]); a[0] += 1; a[1] += s.ms; a[2] += s.n_units; a[3] += s.cells
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    print("  %-28s launches %3d  ms %9.2f  units %7d  cells %.3e" % (k, a[0], a[1], a[2], a[3]))
print("slowest launches:")
for s in sorted(pr.launch_stats(), key=lambda s: -s.ms)[:16]:
    print("  %-28s ms %8.2f units %6d cells %.3e" % (s.kernel.decode(), s.ms, s.n_units, s.cells))
