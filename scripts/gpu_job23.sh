#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wg.py tests/test_gpu_parity.py -m gpu -q --timeout=600 -k "gradient_all_parameters or d40_value_and_gradient or unsupported_raises" > gpurun_out/j23_pytest.log 2>&1; echo "rc $?"; tail -5 gpurun_out/j23_pytest.log | cut -c1-300
timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 all 2>&1 | grep -v amdgpu.ids
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/ab/libcdkf_awgprof.so timeout 600 python scripts/gpu_time_grad_l96.py d=40 n=256 t=100 2>&1 | grep "awg cycles" | awk 'NR%3==1'
