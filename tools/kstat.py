"""Print value, ms/step and the roofline kernel of a bench.py log (last line = the JSON line).
usage: python tools/kstat.py <log>"""
import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"])
