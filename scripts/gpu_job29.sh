#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_soak.py -m gpu -q --timeout=900 -k "gradient_layouts or beyond_eight" > gpurun_out/j29_pytest.log 2>&1; echo "rc $?"; tail -25 gpurun_out/j29_pytest.log | cut -c1-300
