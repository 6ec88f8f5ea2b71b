#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for dm in "32 8" "40 20" "46 6"; do
  echo "== $dm"; AMD_LOG_LEVEL=1 timeout 600 python scripts/dbg_cwg.py $dm 2>&1 | grep -v amdgpu.ids | grep -v "^  File" | tail -5 | cut -c1-300
done
echo "== fp32 60 4"; AMD_LOG_LEVEL=1 timeout 600 python scripts/dbg_cwg32.py 60 4 "" 2>&1 | grep -v amdgpu.ids | grep -v "^  File" | tail -5 | cut -c1-300
timeout 1500 python -m pytest tests/test_custom_drift.py -q -m gpu -x --timeout=900 > gpurun_out/j58.log 2>&1
grep -v amdgpu.ids gpurun_out/j58.log | tail -15 | cut -c1-300
