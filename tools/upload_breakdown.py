"""Where a host thread spends its time in vpr_upload_variants (one_pass leg of bench.py): three uploads of one whole-genome-like
batch from its variant tables with the library's own lap timer (VPR_TIMING=1 prints the laps to stderr), then one execute.
usage: VPR_TIMING=1 python tools/upload_breakdown.py [n_sc]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcfdist_amd import api
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000000
syn = api.Synth(n_sc=n, seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
b = syn.batch(copy=False)
pr = api.PrecisionRecall()
for it in range(3):
    t0 = time.perf_counter(); pr.upload_variants(syn.struct, b); t1 = time.perf_counter()
    pr.execute(); t2 = time.perf_counter()
    r = pr.download(); t3 = time.perf_counter()
    print("upload_variants %.1f ms, execute %.1f ms (kernels %.1f), download %.1f ms" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, pr.timing().ms_total, (t3 - t2) * 1e3), flush=True)
print("K0 on the device (events around the upload's prep kernels): %.2f ms" % pr.timing().ms_prep)
