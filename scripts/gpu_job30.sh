#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_soak.py tests/test_gpu_wg.py tests/test_fit.py -m gpu -q --timeout=900 -k "gradient_layouts or beyond_eight or gradient_all_parameters or d40_value_and_gradient or lorenz96_forcing" > gpurun_out/j30_pytest.log 2>&1; echo "rc $?"; tail -8 gpurun_out/j30_pytest.log | cut -c1-300
timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 all 2>&1 | grep -v amdgpu.ids
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_awgprof.so timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 2>&1 | grep "awg cycles" | awk 'NR%3==1'
