import json,sys
d=json.loads(sys.stdin.read()); print(sys.argv[1], round(d["value"]), round(d["ms_per_step"],4), d["kernels_avg_ms"])
