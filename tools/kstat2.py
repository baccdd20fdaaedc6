"""per-kernel ms per step of bench.py logs (last line = the JSON line).  usage: python tools/kstat2.py <log> [<log> ...]"""
import json, sys
for f in sys.argv[1:]:
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "ms_per_step", d["ms_per_step"], "kernel_ms", d["kernel_ms_per_step"], "value", d["value"], d["step_parts_ms"])
    n = d["steps"]
    for k, v in sorted(d["kernels"].items(), key=lambda kv: -(kv[1]["ms"] + kv[1]["other_ms"])):
        print("   %-26s main %6.2f ms/step  other %6.2f" % (k, v["ms"] / n, v["other_ms"] / n))
