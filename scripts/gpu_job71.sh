#!/bin/bash
cd $GRAFT_REPO_ROOT
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_prof_lib/libcdkf_hip_prof.so timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 2>&1 | grep -v amdgpu.ids | grep "awg cycles" | tail -2 | cut -c1-400
