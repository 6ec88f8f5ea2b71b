#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_soak.py -m gpu -q --timeout=900 -k "edges_of_its_lds" > gpurun_out/j38_pytest.log 2>&1; echo "rc $?"; tail -30 gpurun_out/j38_pytest.log | cut -c1-300
