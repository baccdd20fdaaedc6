"""Which seeds of the fuzzer's SV shape (tests/fuzz_parity.py, shape 6) reach the dense level's int16 score rows and the
deferred edit distances?  GPU only (no oracle): used to pick the fixed cases of tests/test_gpu_configs.py."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import fuzz_parity
from vcfdist_amd import api, _abi as A
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    shape, kw, bm = fuzz_parity.random_workload(seed, 6)
    b = api.Synth(**kw).batch()
    pr = api.PrecisionRecall(A.default_config(band_mode=bm))
    r = pr.run(b)
    names = sorted({s.kernel.decode() for s in pr.launch_stats()})
    t = pr.timing()
    big = max(max(b.lens(k)[q] for q in (0, 1)) + b.lens(k)[4] for k in range(b.n_sc))
    print(seed, "bm", bm, "n_sc", b.n_sc, "Lq+Lr", big, "cells %.1e" % b.dense_cells(), "retries", t.n_band_retries, "ms_ed %.2f" % t.ms_ed,
          "max s", int(r.aln_dist.max()), "max ref_ed", int(max(x.max() if len(x) else 0 for h in range(4) for x in r.ref_ed[h])),
          "s16" if any("s16" in n for n in names) else "", "dense" if any(n.startswith("k_fwd<") for n in names) else "", flush=True)
