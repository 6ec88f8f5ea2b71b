#!/bin/bash
# Cycles per phase of the wavefront-per-trajectory Lorenz-96 reverse sweep (W40A_TICK, cdkf_adjoint_w40_kernels.h).  Here (no GPU needed):
#   cd cd_dynamax_amd/csrc && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -DCDKF_W40A_PROFILE -c launch_adjw40.hip -o /tmp/adjw40_prof.o &&
#   mkdir -p ../../gpurun_prof_lib && hipcc --offload-arch=gfx950 -shared -fPIC -o ../../gpurun_prof_lib/libcdkf_hip_prof.so $(ls ../../build/csrc/*.o | grep -v launch_adjw40.o) /tmp/adjw40_prof.o -lhiprtc -ldl
# then on the GPU box: gpurun -- 'bash scripts/prof_w40a.sh'
cd $GRAFT_REPO_ROOT
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/gpurun_prof_lib/libcdkf_hip_prof.so timeout 600 python scripts/time_awg.py 256 100 2 2>&1 | grep -v amdgpu.ids | cut -c1-700
