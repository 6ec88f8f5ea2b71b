#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 31 32 33; do timeout 1200 python scripts/gpu_fuzz_filters.py $s 30 2>&1 | grep -v amdgpu.ids | tail -8; done
