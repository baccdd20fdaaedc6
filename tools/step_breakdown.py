"""Host-side breakdown of one bench step: execute / download / counters."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from vcfdist_amd import api, summary
syn = api.Synth(n_sc=int(os.environ.get("NSC", "1000000")), seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000)
b = syn.batch(copy=False)
pr = api.PrecisionRecall(); pr.upload(b)
vv = syn.variants()
summary.upload_var_class(pr, [summary.var_class(vv.var_type[s], vv.var_ref_len[s], vv.var_alt_len[s]) for s in range(4)])
res = None
for it in range(4):
    t0 = time.perf_counter(); pr.execute(); t1 = time.perf_counter()
    res = pr.download(res); t2 = time.perf_counter()
    c = summary.pr_counts(pr, None, None); t3 = time.perf_counter()
    print("execute %.2f ms (kernels %.2f)  download %.2f ms  pr_counts %.2f ms" % ((t1 - t0) * 1e3, pr.timing().ms_total, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
nb = sum(a.nbytes for name, _ in res.PER_VAR for h in range(4) for w in range(2) for a in [getattr(res, name)[h][w]])
print("per-variant result bytes %.1f MB, alignment-level %.1f MB" % (nb / 1e6, (res.aln_dist.nbytes + res.aln_status.nbytes + res.aln_end_plane.nbytes * 2 + res.sc_phase.nbytes * 3) / 1e6))
