#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== -O3 full sequence"; CDKF_CUSTOM_OPT=-O3 CDKF_FUZZ_VERBOSE=1 timeout 900 python scripts/gpu_fuzz_custom.py 12 1 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-330
echo "== -O3 only grad"; CDKF_FUZZ_ONLY_GRAD=1 CDKF_CUSTOM_OPT=-O3 CDKF_FUZZ_VERBOSE=1 timeout 900 python scripts/gpu_fuzz_custom.py 12 1 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-330
