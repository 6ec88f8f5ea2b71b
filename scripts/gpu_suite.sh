#!/bin/bash
# The round-end checks in one gpurun call: /usr/local/graft/bin/gpurun --timeout 3600 -- 'bash scripts/gpu_suite.sh [tag]'
# (-m gpu suite, __graft_entry__.smoke(), the default bench line into gpurun_out/<tag>_bench.json)
cd $GRAFT_REPO_ROOT
TAG=${1:-run}
mkdir -p gpurun_out
echo "== GPU suite"; timeout 2400 python -m pytest tests -m gpu -q --timeout=600 > gpurun_out/${TAG}_pytest.log 2>&1; echo "rc $?"; grep -E "passed|failed|error" gpurun_out/${TAG}_pytest.log | tail -3 | cut -c1-300
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | cut -c1-300
echo "== bench"; timeout 900 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err; echo "bench rc $?"; tail -2 gpurun_out/${TAG}_bench.err
