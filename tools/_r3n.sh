mkdir -p gpurun_out
python tools/pipeline3_probe.py f16x3 2>&1 | grep -v Warn | tail -9

timeout 600 python -m pytest tests/test_contour_ranking.py tests/test_glue_golden.py -m gpu -q -x > gpurun_out/pytest_gpu_r03n.log 2>&1; echo "pytest rc=$?"; tail -2 gpurun_out/pytest_gpu_r03n.log
