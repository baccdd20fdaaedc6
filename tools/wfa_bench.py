"""biWFA clustering: GPU (vcl_wfa_cluster) vs the CPU oracle on one synthetic haplotype."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle_lib as O
from vcfdist_amd import cluster as K
n_var = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
rng = np.random.RandomState(7)
t0 = time.time()
# contig: random bases with a short tandem repeat every ~400 bases
L = n_var * 60
arr = rng.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=L)
for s in range(200, L - 200, 400):
    u = rng.randint(1, 5); k = rng.randint(4, 20)
    arr[s:s + u * k] = np.tile(arr[s:s + u], k)
ctg = arr.tobytes().decode()
pos, typ, refs, alts = [], [], [], []
p = 30
gaps = rng.choice([1, 2, 3, 5, 8, 20, 60, 200], size=n_var)
kinds = rng.choice([1, 2, 3], p=[0.6, 0.2, 0.2], size=n_var)
sizes = rng.randint(1, 9, size=n_var)
for i in range(n_var):
    p += int(gaps[i])
    if p > L - 60: break
    t = int(kinds[i]); k = int(sizes[i])
    if t == 1: r = ctg[p]; a = "A" if r != "A" else "C"
    elif t == 2: r = ""; a = ctg[p:p + k]
    else: r = ctg[p:p + k]; a = ""
    pos.append(p); typ.append(t); refs.append(r); alts.append(a)
    p += len(r) + 1
hap = K.HapSeq(pos, typ, refs, alts)
print("contig %d bases, %d variants (built in %.1f s)" % (len(ctg), len(hap.pos), time.time() - t0), flush=True)
t0 = time.time(); got, sg = K.wfa_cluster(hap, ctg); tg = time.time() - t0
t0 = time.time(); got, sg = K.wfa_cluster(hap, ctg); tg = time.time() - t0
ncpu = min(len(hap.pos), 20000)
sub = K.HapSeq(pos[:ncpu], typ[:ncpu], refs[:ncpu], alts[:ncpu])
t0 = time.time(); want, so = K.wfa_cluster(sub, ctg, L=O.lib(), prefix="vco"); tc = time.time() - t0
gsub, _ = K.wfa_cluster(sub, ctg)
print("GPU: %d clusters, %d iterations, %d align + %d reach calls, %.1f ms device, %.3f s wall  -> %.0f variants/s"
      % (got.n, sg["iterations"], sg["align_calls"], sg["reach_calls"], sg["ms_device"], tg, len(hap.pos) / tg))
print("CPU oracle (1 thread) on the first %d variants: %.3f s -> %.0f variants/s; GPU == oracle there: %s"
      % (ncpu, tc, ncpu / tc, gsub == want))
