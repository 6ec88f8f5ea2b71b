#!/bin/bash
# Toolchain investigation: the fp64 workgroup kernels with eight covariance entries per thread return NaN
# when launch_wg8.hip is built at -O3 (it ships at -O1).  Stage 1: a scan of compiler settings, compiled in parallel ON the GPU box and
# run there; stage 2 (argument "bisect"): LLVM's -opt-bisect-limit binary search.   gpurun -- 'bash scripts/bisect_wg8.sh [bisect]'
cd $GRAFT_REPO_ROOT/cd_dynamax_amd/csrc
OUT=$GRAFT_REPO_ROOT/gpurun_out/bisect_wg8; mkdir -p $OUT
F="-std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed"
OBJS=$(ls ../../build/csrc/*.o | grep -v launch_wg8.o)
cat > /tmp/chk46.py <<'PY'
import os, sys
ROOT = os.environ["GRAFT_REPO_ROOT"]
sys.path[:0] = [ROOT, os.path.join(ROOT, "oracle"), os.path.join(ROOT, "tests")]
os.environ["CDKF_NO_WAVE40"] = "1"
import numpy as np
import cdkf_oracle as o
import cd_dynamax_amd as cd
from helpers import lorenz96_model, params_from, relerr
rng = np.random.default_rng(5)
mdl = lorenz96_model(46, 46)
t = o.irregular_times(rng, 1, 3, 0.036); y = o.simulate(mdl, t, rng)
ref = o.ekf_filter(mdl, t, y)
hp = cd.EKFHyperParams(diffeqsolve_settings={"max_steps": 50})
post = cd.cdnlgssm_filter(params_from(mdl), y, t[..., None], hp)
fm = np.asarray(post.filtered_means)
e = [relerr(fm[0, k], ref["filtered_means"][0, k]) for k in range(3)]
eP = [relerr(np.asarray(post.predicted_covariances)[0, k], ref["predicted_covariances"][0, k]) for k in range(3)]
pc = np.asarray(post.predicted_covariances)[0, 0].reshape(-1); rc = ref["predicted_covariances"][0, 0].reshape(-1)
err = np.abs(pc - rc) / np.abs(rc).max()
groups = [float(err[u * 512:(u + 1) * 512].max()) for u in range(5)]
bad_e = np.nonzero(err > 1e-9)[0]
print("RESULT", "OK" if max(e) < 1e-9 else "BAD", "means per step", ["%.2e" % v for v in e], "pred cov step 0: max err per entry group u = e // 512:", ["%.2e" % g for g in groups],
      "bad entries", len(bad_e), "first", bad_e[:12].tolist(), "last", bad_e[-6:].tolist(), "tid of bad (e % 512) min/max", (int((bad_e % 512).min()), int((bad_e % 512).max())) if len(bad_e) else None)
PY
run() {  # $1 = object
  hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/libcdkf_b.so $OBJS $1 -lhiprtc -ldl || return 2
  CDKF_LIB_PATH=/tmp/libcdkf_b.so timeout 120 python3 /tmp/chk46.py 2>&1 | grep RESULT | cut -c1-700
}
if [ "$1" != "bisect" ]; then
  i=0
  while IFS= read -r FL; do
    i=$((i+1))
    ( hipcc $FL $F -c launch_wg8.hip -o /tmp/wg8_v$i.o 2> /tmp/wg8_v$i.log ) &
  done <<'LIST'
-O1
-O3
LIST
  wait
  i=0
  while IFS= read -r FL; do
    i=$((i+1))
    echo "[$FL] $(run /tmp/wg8_v$i.o)" | tee -a $OUT/scan.txt
  done <<'LIST'
-O1
-O3
LIST
  exit 0
fi
probe() {  # $1 = limit
  hipcc -O3 $F -mllvm -opt-bisect-limit=$1 -c launch_wg8.hip -o /tmp/wg8_b.o 2> /tmp/bis_$1.log || { echo "compile failed at $1"; tail -3 /tmp/bis_$1.log; return 2; }
  run /tmp/wg8_b.o
}
hipcc -O3 $F --cuda-device-only -mllvm -opt-bisect-limit=-1 -c launch_wg8.hip -o /tmp/x.o 2> /tmp/bis_all.log
TOTAL=$(grep -c "BISECT: running pass" /tmp/bis_all.log)
echo "device passes: $TOTAL" | tee $OUT/log.txt
lo=0; hi=$TOTAL
while [ $((hi - lo)) -gt 1 ]; do
  mid=$(( (lo + hi) / 2 ))
  r=$(probe $mid)
  echo "limit $mid: $r" | tee -a $OUT/log.txt
  if echo "$r" | grep -q "RESULT OK"; then lo=$mid; else hi=$mid; fi
done
echo "first bad limit: $hi" | tee -a $OUT/log.txt
grep "BISECT: running pass ($hi)" /tmp/bis_all.log | tee -a $OUT/log.txt
grep "BISECT: running pass ($lo)" /tmp/bis_all.log | tee -a $OUT/log.txt
