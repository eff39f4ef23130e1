cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -q -m gpu -k "cross_axis or train_static_step" > gpurun_out/r6c24_cross.txt 2>&1; grep -v "of limit" gpurun_out/r6c24_cross.txt | tail -n 12 | cut -c1-300
