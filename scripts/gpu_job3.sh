#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== the test that hung"; timeout 200 python -X faulthandler -m pytest "tests/test_gpu_wg.py::test_reverse_sweep_other_runge_kutta_methods" -x -q -o faulthandler_timeout=50 --timeout=150 > gpurun_out/j3_hang.log 2>&1; echo "rc $?"; tail -40 gpurun_out/j3_hang.log | cut -c1-200
echo "== wave8 / gradient tests"; timeout 900 python -m pytest tests/test_gpu_wg.py tests/test_fit.py -m gpu -q --timeout=240 -k "not other_runge_kutta" > gpurun_out/j3_pytest.log 2>&1; echo "rc $?"; tail -15 gpurun_out/j3_pytest.log | cut -c1-300
echo "== config5"; timeout 300 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
export CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_prof.so
timeout 300 python scripts/run_config.py config5 1 > gpurun_out/j3_prof.log 2>&1
grep "cycles" gpurun_out/j3_prof.log | sort | uniq | head -40
