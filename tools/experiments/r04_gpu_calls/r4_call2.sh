cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 900 python tools/motionbench.py --rounds 2 m0 base nodma nobar nolds burst saddr ah2 ah8 nodb nodbl > $O/r4c2_motion.txt 2>&1; echo mb rc=$?
tail -14 $O/r4c2_motion.txt
