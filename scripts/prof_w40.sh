#!/bin/bash
# Where does a d = 40 step of ekf_filter_wave_l96_kernel go?  Ablation: the config-4 slice timed with phases skipped
# (CDKF_W40_ABLATE mask: 1 factorisation, 2 triangular solves, 4 rank-d products, 8 predict, 16 covariance stores).
export CDKF_LIB_PATH=${CDKF_LIB_PATH:-$(dirname "$0")/../cd_dynamax_amd/lib/libcdkf_hip_prof.so}  # scripts/w40_prof_build.sh: the shipped library has no ablation switch
for m in 0 1 2 4 8 16 31 32 34 33; do
  echo "mask $m: $(CDKF_W40_ABLATE=$m python3 scripts/run_config.py config4 1 2>/dev/null | grep -o "'ekf_filter_ms': [0-9.]*")"
done
