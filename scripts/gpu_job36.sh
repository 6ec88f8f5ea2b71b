#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_wg.py tests/test_gpu_soak.py -m gpu -q --timeout=900 -k "adaptive or beyond_eight or gradient_all or other_runge" > gpurun_out/j36_pytest.log 2>&1; echo "rc $?"; tail -25 gpurun_out/j36_pytest.log | cut -c1-300
