for v in "" _ah2 ""; do
  echo "=== variant '$v'"
  DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip$v.so python tools/framebench.py --frames 2 2>&1 | grep -E "k_static_views|k_motion|k_dynamic_views|k_static_blend|points|total"
  DYNIBAR_HIP_LIB=$PWD/dynibar_amd/csrc/libdynibar_hip$v.so python bench.py --steps 20 --warmup 5 --no-extra --no-traffic --cpu-rays 0 2>/dev/null | python tools/benchline.py "bench$v"
done
