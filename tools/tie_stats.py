"""How often does an order-defined swap tie (dist.cpp:347,376) change a *result*?  Runs the HIP path and the
oracle on a WGS-like sample and counts superclusters (a) where the oracle's containers kept a swap predecessor
other than the library's rule and (b) where any per-variant result actually differs."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vcfdist_amd import api, _abi as A
n = int(os.environ.get("NSC", "100000"))
b = api.Synth(n_sc=n, seed=0x5eed, len_mode=1, len_a=20.0, len_b=1.2, len_min=4, len_max=10000).batch()
ex = O.Extra(b)
want = O.run(b, extra=ex)
got = api.PrecisionRecall().run(b)
tied = ex.swap_used_conflict_nonmax.reshape(-1, 4).sum(axis=1) > 0
flag = (want.aln_status & 1).reshape(-1, 4).any(axis=1)
diff_sc = np.zeros(b.n_sc, bool)
for h in range(4):
    sc_of_var = np.repeat(np.arange(b.n_sc), np.diff(b.var_off[h]))
    for w in range(2):
        for name, dt in A.Results.PER_VAR:
            x, y = getattr(got, name)[h][w], getattr(want, name)[h][w]
            if dt == np.float32: x, y = x.view(np.uint32), y.view(np.uint32)
            d = x != y
            np.logical_or.at(diff_sc, sc_of_var[d], True)
for f in ("aln_beg_plane",):
    d = (getattr(got, f) != getattr(want, f)).reshape(-1, 4).any(axis=1)
    print(f, "differs in", int(d.sum()), "superclusters")
print("superclusters", b.n_sc, "tie-flagged", int(flag.sum()), "oracle kept a non-rule predecessor", int(tied.sum()),
      "any per-variant result differs", int(diff_sc.sum()), "of which outside the tied set", int((diff_sc & ~tied).sum()))
