#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 401 402 403; do timeout 1200 python scripts/gpu_fuzz_r03.py $s 24 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -3 | cut -c1-300; done
for s in 41 42; do CDKF_FUZZ_DMAX=36 timeout 1800 python scripts/gpu_fuzz_custom.py $s 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -3 | cut -c1-300; done
timeout 1200 python scripts/gpu_fuzz_grads.py 404 30 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -3 | cut -c1-300
CDKF_FUZZ_LONG=1 timeout 2400 python scripts/gpu_fuzz_filters.py 405 12 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -3 | cut -c1-300
