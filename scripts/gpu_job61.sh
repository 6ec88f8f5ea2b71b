#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
AMD_LOG_LEVEL=1 timeout 1500 python -m pytest tests/test_custom_drift.py -q -m gpu -x --timeout=900 -k "gradient or fit_sgd or derivatives" > gpurun_out/j61.log 2>&1
grep -v amdgpu.ids gpurun_out/j61.log | grep -v "^  File \"/usr" | tail -30 | cut -c1-300
