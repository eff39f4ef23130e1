#!/bin/bash
# round 6, call 39: the bench line with the frame leg as the median of three frames; wall-clock of the default command
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
T0=$(date +%s)
python bench.py > gpurun_out/r6c39_bench.json 2> gpurun_out/r6c39_bench.err
echo "bench wall-clock $(( $(date +%s) - T0 )) s"
python - <<'PY'
import json
d=json.loads(open("gpurun_out/r6c39_bench.json").read().strip().splitlines()[-1])
f=d["extra"]["frame_nvi_288x512"]
print(d["value"], d["ms_per_step"], f["ms_per_frame"], f["ms_per_frame_each"], d["roofline"]["frame_ms"])
PY
