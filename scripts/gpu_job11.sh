#!/bin/bash
cd $GRAFT_REPO_ROOT
echo "== config5"; timeout 300 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
export CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_prof.so
timeout 300 python scripts/run_config.py config5 1 > gpurun_out/j11_prof.log 2>&1
grep "adjoint cycles" gpurun_out/j11_prof.log | sort | uniq | awk 'NR%4==1' | head -12
