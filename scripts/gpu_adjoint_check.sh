#!/bin/bash
# Dev aid (gpurun -- 'bash scripts/gpu_adjoint_check.sh'): time the shape-generic reverse sweep, run every test that reaches it and one fuzz seed
cd $GRAFT_REPO_ROOT
timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 all 2>&1 | grep -v amdgpu.ids | tail -4
timeout 600 python scripts/gpu_time_grad_l96.py d=40 m=20 n=256 t=100 2>&1 | grep -v amdgpu.ids | tail -2
timeout 1500 python -m pytest tests/test_gpu_wg.py tests/test_gpu_soak.py tests/test_fit.py tests/test_custom_drift.py -q -m gpu -x --timeout=900 -k "gradient or reverse or lorenz96 or adjoint" 2>&1 | tail -3 | cut -c1-300
timeout 1200 python scripts/gpu_fuzz_r03.py ${1:-303} 24 2>&1 | grep -v amdgpu.ids | grep -v "^  " | tail -3 | cut -c1-300
