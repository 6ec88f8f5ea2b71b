#!/bin/bash
cd $GRAFT_REPO_ROOT
for s in 91 92 93; do CDKF_FUZZ_LONG=1 timeout 2400 python scripts/gpu_fuzz_filters.py $s 16 2>&1 | grep -v amdgpu.ids | tail -8 | cut -c1-300; done
