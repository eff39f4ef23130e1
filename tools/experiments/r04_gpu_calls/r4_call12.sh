cd /root/repo; O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests -m gpu -x -q -k "gather or project or sampling or static_pass or render_rays_mv or frames or checkpoint" > $O/r4c12_tests.txt 2>&1; echo tests rc=$?
tail -3 $O/r4c12_tests.txt | cut -c1-200
echo "== rgba on"; timeout 600 python tools/k1sweep.py 7,8,11,15 2>&1 | head -1 | cut -c1-400
echo "== rgba off"; DYNIBAR_K1_RGBA=0 timeout 600 python tools/k1sweep.py 7,8,11,15 2>&1 | head -1 | cut -c1-400
echo "== no rgb taps at all (g4)"; DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip_g4.so timeout 600 python tools/k1sweep.py 7,8,11,15 2>&1 | head -1 | cut -c1-400
timeout 600 python tools/abbench.py --rounds 2 --iters 30 base > $O/r4c12_ab.txt 2>&1; tail -4 $O/r4c12_ab.txt
DYNIBAR_K1_RGBA=0 timeout 600 python tools/abbench.py --rounds 2 --iters 30 base > $O/r4c12_ab_off.txt 2>&1; tail -4 $O/r4c12_ab_off.txt
