cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 600 python tools/phasebench.py > $O/r4c8_phase.txt 2>&1; echo phase rc=$?
grep -A12 "point chain" $O/r4c8_phase.txt | cut -c1-400
