cd /root/repo
timeout 2400 bash tools/collect_profiles.sh r04 > gpurun_out/r4c14_collect.log 2>&1; echo collect rc=$?
ls gpurun_out/prof_r04 | head -30
tail -c 1500 gpurun_out/prof_r04/r04_bench_n1.json
