#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python scripts/gpu_time_grad_l96.py d=40 n=2048 t=500 2>&1 | grep -v amdgpu.ids | tail -2
timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 all 2>&1 | grep -v amdgpu.ids | tail -4
bash scripts/prof_r02.sh r03_h_config4_value_and_grad config4_value_and_grad 2>&1 | tail -1 | cut -c1-300
