cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r4c6_tests.txt 2>&1; echo tests rc=$?
tail -25 $O/r4c6_tests.txt | cut -c1-250
timeout 900 python tools/abbench.py --frame --rounds 2 --iters 20 base p0 > $O/r4c6_ab.txt 2>&1; echo ab rc=$?
tail -7 $O/r4c6_ab.txt
