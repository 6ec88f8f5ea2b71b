#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== config5"; timeout 300 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
timeout 900 python -m pytest tests/test_gpu_wg.py tests/test_fit.py -m gpu -q --timeout=600 -k "mlp or c5 or wg_and_reg or checkpoints" > gpurun_out/j28_pytest.log 2>&1; echo "rc $?"; tail -5 gpurun_out/j28_pytest.log | cut -c1-300
