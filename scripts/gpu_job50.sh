#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 51 52 53 54; do timeout 1500 python scripts/gpu_fuzz_grads.py $s 30 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-260; done
