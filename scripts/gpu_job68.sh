#!/bin/bash
cd $GRAFT_REPO_ROOT
export CDKF_CUSTOM_OPT=-O3
for s in 11 12; do timeout 1500 python scripts/gpu_fuzz_custom.py $s 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -3 | cut -c1-300; done
CDKF_FUZZ_DMAX=40 timeout 2400 python scripts/gpu_fuzz_custom.py 21 8 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -3 | cut -c1-300
CDKF_FUZZ_DMAX=32 timeout 2400 python scripts/gpu_fuzz_custom.py 31 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -3 | cut -c1-300
timeout 900 python -m pytest tests/test_custom_drift.py -q -m gpu -x --timeout=900 2>&1 | tail -2 | cut -c1-200
timeout 900 python scripts/gpu_time_custom_wide.py 2>&1 | grep -v amdgpu.ids | grep "as source" | cut -c1-200
