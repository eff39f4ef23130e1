cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -x -q > $O/r4c13_tests.txt 2>&1; echo tests rc=$?
tail -3 $O/r4c13_tests.txt | cut -c1-200
timeout 600 python tools/abbench.py --frame --rounds 2 --iters 30 base > $O/r4c13_ab.txt 2>&1; grep round $O/r4c13_ab.txt | sed 's/base /ws   /'
DYN_BLEND_STREAM=1 timeout 600 python tools/abbench.py --frame --rounds 2 --iters 30 base > $O/r4c13_ab_stream.txt 2>&1; grep round $O/r4c13_ab_stream.txt | sed 's/base /stream/'
