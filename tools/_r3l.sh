mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_contour_ranking.py tests/test_glue_golden.py tests/test_gpu_parity.py -m gpu -q -x -k "contour or device or ranking or page_box or pipeline or extract_page or morph" > gpurun_out/pytest_gpu_r03l.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/pytest_gpu_r03l.log; grep -n "host fallbacks" gpurun_out/pytest_gpu_r03l.log | head -12
