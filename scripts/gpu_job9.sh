#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== wg kernels (EPT = 1) under other solvers"; timeout 120 python scripts/dbg_wg.py 2>&1 | grep -v amdgpu.ids | grep "tsit5" | grep "mlp5\|l96_6 "
echo "== GPU suite"; timeout 1500 python -m pytest tests -m gpu -q --timeout=300 > gpurun_out/j9_pytest.log 2>&1; echo "rc $?"; tail -30 gpurun_out/j9_pytest.log | cut -c1-300
echo "== bench"; timeout 600 python bench.py > gpurun_out/j9_bench.json 2> gpurun_out/j9_bench.err; echo "bench rc $?"; tail -3 gpurun_out/j9_bench.err
export CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_prof.so
timeout 300 python scripts/run_config.py config5 1 > gpurun_out/j9_prof.log 2>&1
grep "cycles" gpurun_out/j9_prof.log | sort | uniq | awk 'NR%4==1' | head -24
