#!/bin/bash
cd $GRAFT_REPO_ROOT
for dm in "12 5" "7 3"; do
  echo "== $dm"; AMD_LOG_LEVEL=1 timeout 600 python scripts/dbg_cwg32.py $dm 2>&1 | grep -v amdgpu.ids | grep -v "^  File" | tail -8 | cut -c1-400
done
