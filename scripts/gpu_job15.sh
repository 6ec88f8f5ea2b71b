#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== ept8"; timeout 300 python scripts/dbg_wg_ept8.py 2>&1 | grep -v amdgpu.ids
echo "== GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/j15_pytest.log 2>&1; echo "rc $?"; tail -8 gpurun_out/j15_pytest.log | cut -c1-300
echo "== bench"; timeout 600 python bench.py > gpurun_out/j15_bench.json 2> gpurun_out/j15_bench.err; echo "bench rc $?"; tail -3 gpurun_out/j15_bench.err
