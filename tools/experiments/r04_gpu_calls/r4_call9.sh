cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python tools/k1sweep.py > $O/r4c9_k1.txt 2>&1; echo k1 rc=$?
cat $O/r4c9_k1.txt | cut -c1-400
