import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(sys.argv[1], d["value"], d["ms_per_step"], d["roofline"]["kernel"], d["roofline"]["avg_launch_ms"])
