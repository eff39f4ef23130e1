cd /root/repo; O=gpurun_out
timeout 200 python tools/soak.py 110 > $O/r4c17_soak.txt 2>&1; cat $O/r4c17_soak.txt | cut -c1-260
