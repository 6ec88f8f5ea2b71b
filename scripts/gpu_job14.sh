#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 200 python scripts/dbg_wg28.py 2>&1 | grep -v amdgpu.ids
timeout 900 python -m pytest tests -m gpu -q --timeout=300 -k "wavefront_kernels_other_state or reverse_sweep_on_the_lane_grid or linear or lorenz96 or wg_and_reg" > gpurun_out/j14_pytest.log 2>&1; echo "rc $?"; tail -8 gpurun_out/j14_pytest.log | cut -c1-250
