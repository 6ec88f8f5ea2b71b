#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 101 102 103; do timeout 2400 python scripts/gpu_fuzz_misc.py $s 30 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v "^  " | tail -8 | cut -c1-300; done
