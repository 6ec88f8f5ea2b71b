#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_comm.py -m gpu -q --timeout=600 2>&1 | tail -3
timeout 900 python bench.py --no-cpu-baseline > gpurun_out/j41_bench.json 2> gpurun_out/j41_bench.err; echo "bench rc $?"
python - <<'P'
import json
d=json.load(open('gpurun_out/j41_bench.json'))
print(d['value'], d['roofline']['traffic_source'], d['collective'])
for k,v in d['other_configs'].items(): print(k, v.get('collective'), {kk: round(vv,2) for kk,vv in v.items() if kk.endswith('_ms')})
P
