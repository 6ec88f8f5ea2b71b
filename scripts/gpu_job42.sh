#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 11 12 13; do timeout 900 python scripts/gpu_fuzz_r03.py $s 20 2>&1 | grep -v amdgpu.ids | tail -5; done
