#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 61 62 63; do timeout 1500 python scripts/gpu_fuzz_solvers.py $s 24 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300; done
