mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fused_x3 or layer or predict_448" > gpurun_out/pytest_gpu_r03i.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pytest_gpu_r03i.log; grep -n "AssertionError: (" gpurun_out/pytest_gpu_r03i.log | head -3
