cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 1200 python tools/abbench.py --rounds 3 --iters 30 base va vp vp2 va2 > $O/r4c7_ab.txt 2>&1; echo ab rc=$?
tail -8 $O/r4c7_ab.txt
