#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
echo "== GPU suite"; timeout 2400 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/j33_pytest.log 2>&1; echo "rc $?"; tail -8 gpurun_out/j33_pytest.log | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -5
