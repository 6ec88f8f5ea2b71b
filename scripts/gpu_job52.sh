#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 71 72 73 74; do timeout 1500 python scripts/gpu_fuzz_batches.py $s 40 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300; done
