cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python tools/abbench.py --frame --rounds 2 --iters 20 base r3 > $O/r4c5_ab.txt 2>&1; echo ab rc=$?
tail -7 $O/r4c5_ab.txt
timeout 900 python -m pytest tests/test_gpu_dist.py -x -q > $O/r4c5_dist.txt 2>&1; echo dist rc=$?
tail -15 $O/r4c5_dist.txt
(python tools/step_loop.py 8 > $O/r4c5_loop_step.txt 2>&1 &)
sleep 5
for i in 1 2 3; do rocm-smi --showpower --showclocks --showtemp 2>/dev/null | grep -i "power\|sclk\|junction" | tr '\n' ';'; echo; sleep 1; done > $O/r4c5_smi_step.txt 2>&1
sleep 4; cat $O/r4c5_loop_step.txt; cat $O/r4c5_smi_step.txt
rocm-smi --showmaxpower 2>/dev/null | grep -i "power" ; rocm-smi --showpowerprofile 2>/dev/null | tail -5
