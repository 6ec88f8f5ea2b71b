#!/bin/bash
# MLP stage checkpoints: correctness of the reverse sweep with them, A/B timing, phase profile of the reverse sweep.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_wg.py tests/test_fit.py tests/test_gpu_soak.py -m gpu -x -q > gpurun_out/j2_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/j2_pytest.log
tail -4 gpurun_out/j2_pytest.log
echo "== with MLP checkpoints"; timeout 600 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
echo "== without (CDKF_ADJ_MLP_CKPT=0)"; CDKF_ADJ_MLP_CKPT=0 timeout 600 python scripts/run_config.py config5 1 2>&1 | grep -o "'[a-z_0-9]*_ms': [0-9.]*\|config5[a-z_0-9]*" | tr '\n' ' '; echo
export CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_prof.so
timeout 600 python scripts/run_config.py config5 1 > gpurun_out/j2_prof_ckm.log 2>&1
grep "cycles" gpurun_out/j2_prof_ckm.log | sort | uniq -c | sort -rn | head -30
CDKF_ADJ_MLP_CKPT=0 timeout 600 python scripts/run_config.py config5 1 > gpurun_out/j2_prof_nockm.log 2>&1
echo "== no ckm"; grep "adjoint cycles" gpurun_out/j2_prof_nockm.log | sort | uniq -c | sort -rn | head -12
