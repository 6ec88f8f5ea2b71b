#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 11 12; do timeout 1500 python scripts/gpu_fuzz_custom.py $s 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -4 | cut -c1-300; done
CDKF_FUZZ_DMAX=40 timeout 2400 python scripts/gpu_fuzz_custom.py 21 8 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -6 | cut -c1-300
bash scripts/gpu_job64.sh
