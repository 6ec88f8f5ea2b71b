#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== ept8"; timeout 300 python scripts/dbg_wg_ept8.py 2>&1 | grep -v amdgpu.ids
echo "== wg tests"; timeout 900 python -m pytest tests/test_gpu_wg.py -m gpu -q --timeout=300 -x > gpurun_out/j18_pytest.log 2>&1; echo "rc $?"; tail -15 gpurun_out/j18_pytest.log | cut -c1-300
