"""k_zero_lane on batches of ONE alignment length (no tail of long waves): its throughput per alignment-row (kernel tuning aid)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from vcfdist_amd import api, _abi as A
def run(L, n_sc):
    syn = api.Synth(n_sc=n_sc, seed=5, len_mode=2, len_a=float(L), len_min=L, len_max=L)
    b = syn.batch(copy=False)
    pr = api.PrecisionRecall(A.default_config(band_mode=1))
    pr.upload(b)
    best = {}
    for _ in range(3):
        pr.execute()
        for s in pr.launch_stats():
            k = s.kernel.decode()
            if s.n_units * 2 >= n_sc * 4 or k.startswith('k_zero'):
                best[k] = min(best.get(k, 1e9), s.ms)
    n_aln = 4 * n_sc; waves = n_aln / 64; rounds = max(1.0, waves / 6144)
    z = best.get('k_zero_lane', 0)
    print(f"L={L} n_aln={n_aln} waves={waves:.0f}: " + ' '.join(f"{k}={v:.3f}" for k, v in best.items()) +
          f" | zero_lane {z*1e6/(n_aln*L):.4f} ns/row-aln, {z*1e3/(rounds*L*3):.2f} us per row-step per round of 6 waves/SIMD", flush=True)
    pr.close(); syn.close()
for L, n in [(32, 1000000), (32, 96000), (128, 250000), (128, 96000), (512, 62500), (1000, 1024)]:
    run(L, n)
