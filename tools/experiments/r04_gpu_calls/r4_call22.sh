cd /root/repo; O=gpurun_out
timeout 600 python __graft_entry__.py smoke 2>&1 | tail -2 | cut -c1-300
timeout 900 python -m pytest tests -m gpu -x -q > $O/r4c22_tests.txt 2>&1; echo tests rc=$?; tail -2 $O/r4c22_tests.txt
timeout 900 python bench.py > $O/r4c22_bench.json 2> $O/r4c22_bench.err; echo bench rc=$?
python - <<'PY'
import json
d=json.load(open('/root/repo/gpurun_out/r4c22_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['extra']['frame_nvi_288x512']['ms_per_frame'], d['extra']['power_under_step_loop'].get('package_power_w'), d['extra']['eval_loop']['ms_per_view'])
PY
