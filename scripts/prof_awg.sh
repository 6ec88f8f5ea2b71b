#!/bin/bash
# Cycles per phase of the shape-generic reverse sweep (AWG_TICK, cdkf_adjoint_wg_kernels.h).  Here (no GPU needed):
#   cd cd_dynamax_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DCDKF_AWG_PROFILE -c launch_adjwg.hip -o /tmp/adjwg_prof.o &&
#   hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_prof_lib/libcdkf_hip_prof.so $(ls ../../build/csrc/*.o | grep -v launch_adjwg.o) /tmp/adjwg_prof.o -lhiprtc -ldl
# then on the GPU box: gpurun -- 'bash scripts/prof_awg.sh'
cd $GRAFT_REPO_ROOT
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_prof_lib/libcdkf_hip_prof.so timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 2>&1 | grep -v amdgpu.ids | grep "awg cycles" | cut -c1-500
