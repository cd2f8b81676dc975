mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s > gpurun_out/r2b_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2b_pytest.log
