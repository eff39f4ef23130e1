cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r4c11_tests.txt 2>&1; echo tests rc=$?
grep -A6 "parity margins" $O/r4c11_tests.txt | cut -c1-220; tail -2 $O/r4c11_tests.txt
timeout 900 python tools/abbench.py --frame --rounds 2 --iters 20 base p0 > $O/r4c11_ab.txt 2>&1; echo ab rc=$?
tail -7 $O/r4c11_ab.txt
timeout 600 python tools/k1sweep.py 9,10,12,13 > $O/r4c11_k1.txt 2>&1; cat $O/r4c11_k1.txt | cut -c1-420
