rocm-smi --showclocks --showcomputepartition --showmemorypartition --showperflevel --showpower 2>&1 | grep -v "^$" | head -40
python bench.py --steps 20 --warmup 5 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python tools/benchline.py "bench"
rocm-smi --showclocks 2>&1 | grep -E "sclk|mclk|fclk|socclk" | head
python tools/framebench.py --frames 1 2>&1 | grep -E "k_static_views|k_motion|k_dynamic_views|points|total"
