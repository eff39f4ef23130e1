cd /root/repo; O=gpurun_out
PB_V=11 timeout 300 python tools/phasebench.py > $O/r4c18_phase_v11.txt 2>&1
grep -B1 -A24 "view chain (k_static_views) | workgroup middle" $O/r4c18_phase_v11.txt | cut -c1-260
