"""How the kernels of `bench.py` with batches in flight share the device, from a `rocprofv3 --kernel-trace` run:
share of the time with any kernel / with a launch of 10 000+ workgroups running, and the launch durations per step.
usage: python tools/concurrency.py <dir with *_kernel_trace.csv> [first execute] [last execute]   (default: executes 5 .. 19)"""
import collections, csv, glob, os, sys
f = max(glob.glob(os.path.join(sys.argv[1], "**", "*_kernel_trace.csv"), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]),
             r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", ""),
             int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1)) for r in rows)
inits = [k[0] for k in ks if k[2] == "k_init_execute"]        # every vpr_execute begins with this launch
a, b = (int(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5, 19)
t0, t1, n_exec = inits[a], inits[b], b - a


def union(sel):
    ev = []
    for k in ks:
        if k[1] > t0 and k[0] < t1 and sel(k):
            ev.append((max(k[0], t0), 1)); ev.append((min(k[1], t1), -1))
    ev.sort()
    cur, last, busy = 0, t0, 0
    for t, d in ev:
        if cur > 0:
            busy += t - last
        last = t; cur += d
    return busy


tot = t1 - t0
big = lambda k: k[3] >= 10000
print(f"executes {a} .. {b}: {tot / 1e6:.1f} ms, {tot / 1e6 / n_exec:.2f} ms per execute; starts {[round((y - x) / 1e6, 1) for x, y in zip(inits[a:b], inits[a + 1:b + 1])]} ms apart")
print(f"some kernel running: {100 * union(lambda k: True) / tot:.1f} % of the time; a launch of 10 000+ workgroups: {100 * union(big) / tot:.1f} %; "
      f"a copy kernel: {100 * union(lambda k: 'copyBuffer' in k[2]) / tot:.1f} %")
agg = collections.defaultdict(lambda: [0, 0])
for k in ks:
    if t0 <= k[0] < t1 and big(k):
        agg[k[2]][0] += 1; agg[k[2]][1] += k[1] - k[0]
s = 0
for n, (c, ns) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"  {n:24s} {c / n_exec:5.1f} launches of 10 000+ workgroups per execute, {ns / 1e6 / n_exec:6.2f} ms per execute")
    s += ns
print(f"  sum {s / 1e6 / n_exec:.2f} ms per execute")
