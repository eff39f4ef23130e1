cd /root/repo; O=gpurun_out
for rep in 1 2; do
  (python tools/motion_loop.py 6 > $O/r4c16_loop_$rep.txt 2>&1 &)
  sleep 4
  rocm-smi --showclocks --showpower --showperflevel 2>/dev/null | grep -v "^=\|^$" | tr '\n' ';' | cut -c1-900; echo
  sleep 4; cat $O/r4c16_loop_$rep.txt | grep launches
done
python tools/motionbench.py --rounds 3 base 2>&1 | grep round
rocm-smi --showclocks 2>/dev/null | grep -v "^=\|^$" | tr '\n' ';' | cut -c1-900; echo
