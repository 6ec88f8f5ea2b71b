#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 11 12; do timeout 1500 python scripts/gpu_fuzz_custom.py $s 10 2>&1 | grep -v amdgpu.ids | grep -v "^  File \"/usr" | tail -12 | cut -c1-300; done
echo "== timing"; timeout 900 python scripts/gpu_time_custom_wide.py 2>&1 | grep -v amdgpu.ids | tail -10 | cut -c1-200
