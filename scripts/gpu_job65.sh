#!/bin/bash
cd $GRAFT_REPO_ROOT
for o in -O1 -O2; do echo "== $o"; CDKF_CUSTOM_OPT=$o CDKF_FUZZ_D=24 CDKF_FUZZ_VERBOSE=1 timeout 900 python scripts/gpu_fuzz_custom.py 12 1 2>&1 | grep -v amdgpu.ids | tail -2 | cut -c1-300; done
