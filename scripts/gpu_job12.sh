#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 600 python -m pytest tests/test_gpu_wg.py -m gpu -q --timeout=300 -k "wavefront_kernels_other_state or lorenz96 or c4" > gpurun_out/j12_pytest.log 2>&1; echo "rc $?"; tail -12 gpurun_out/j12_pytest.log | cut -c1-250
timeout 300 python scripts/gpu_time_w40dims.py 2>&1 | grep -v amdgpu.ids
echo "== config5"; timeout 300 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
