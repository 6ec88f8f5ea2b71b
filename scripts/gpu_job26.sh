#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_fit.py -m gpu -q --timeout=900 -k "lorenz96" > gpurun_out/j26_pytest.log 2>&1; echo "rc $?"; tail -30 gpurun_out/j26_pytest.log | cut -c1-300
