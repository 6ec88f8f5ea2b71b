#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_custom_drift.py tests/test_gpu_parity.py -m gpu -q --timeout=900 -k "custom or loglik_gradient" > gpurun_out/j34_pytest.log 2>&1; echo "rc $?"; tail -40 gpurun_out/j34_pytest.log | cut -c1-300
