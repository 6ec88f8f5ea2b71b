#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wg.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "gradient_all_parameters or d40_value_and_gradient or unsupported_raises" > gpurun_out/j19_pytest.log 2>&1; echo "rc $?"; tail -40 gpurun_out/j19_pytest.log | cut -c1-300
