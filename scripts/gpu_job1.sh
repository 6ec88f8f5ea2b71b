#!/bin/bash
# Round-3 first GPU job: the GPU suite, the bench line, the phase profile of the state_dim <= 8 sweeps, the crossover table.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/j1_pytest.log 2>&1; echo "pytest rc $?" >> gpurun_out/j1_pytest.log
tail -5 gpurun_out/j1_pytest.log
timeout 600 python bench.py > gpurun_out/j1_bench.json 2> gpurun_out/j1_bench.err; echo "bench rc $?"
CDKF_LIB_PATH=$GRAFT_REPO_ROOT/cd_dynamax_amd/lib/libcdkf_hip_prof.so timeout 600 python scripts/run_config.py config5 1 > gpurun_out/j1_w8prof.log 2>&1
grep "w8 cycles" gpurun_out/j1_w8prof.log | head -20
timeout 900 python scripts/n_sweep_table.py gpurun_out/j1_n_sweep.json > gpurun_out/j1_n_sweep.log 2>&1
tail -70 gpurun_out/j1_n_sweep.log
