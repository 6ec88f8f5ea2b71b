#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 300 python scripts/dbg_w8_f32.py 2>&1 | grep -v amdgpu.ids
for s in 31 32 33 41 42; do timeout 1200 python scripts/gpu_fuzz_filters.py $s 30 2>&1 | grep -v amdgpu.ids | tail -6 | cut -c1-250; done
