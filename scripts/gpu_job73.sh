#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== GPU suite"; timeout 2400 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/j73_pytest.log 2>&1; echo "rc $?"; tail -4 gpurun_out/j73_pytest.log | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
echo "== bench"; timeout 900 python bench.py > gpurun_out/r03_h_bench.json 2> gpurun_out/r03_h_bench.err; echo "bench rc $?"; tail -2 gpurun_out/r03_h_bench.err
