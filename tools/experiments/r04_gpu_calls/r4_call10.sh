cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r4c10_tests.txt 2>&1; echo tests rc=$?
tail -4 $O/r4c10_tests.txt | cut -c1-250
DYN_POINTS_NO_PERSIST=1 timeout 400 python tools/abbench.py --frame --rounds 1 --iters 20 base > $O/r4c10_ab_nopersist.txt 2>&1
timeout 900 python tools/abbench.py --frame --rounds 2 --iters 20 base p0 > $O/r4c10_ab.txt 2>&1; echo ab rc=$?
grep round $O/r4c10_ab_nopersist.txt | sed 's/base /nopersist/'; tail -7 $O/r4c10_ab.txt
