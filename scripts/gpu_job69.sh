#!/bin/bash
cd $GRAFT_REPO_ROOT
F='grep -v amdgpu.ids | grep -v Warning | grep -v "^  "'
timeout 1200 python scripts/gpu_fuzz_filters.py 201 30 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -4 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_grads.py 202 30 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -4 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_r03.py 203 24 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -4 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_solvers.py 204 24 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -4 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_batches.py 205 40 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -4 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_misc.py 206 30 2>&1 | grep -v amdgpu.ids | grep -v Warning | grep -v "^  " | tail -4 | cut -c1-300
