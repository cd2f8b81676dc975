mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -s -x -k "convT or conv2d_transpose or keras23 or injected or whole_image_branch_fused or scaled_page_equals_oracle or config4 or full_page_properties or c_abi_error" > gpurun_out/r2c_pytest.log 2>&1; echo "pytest rc=$?"; tail -25 gpurun_out/r2c_pytest.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
