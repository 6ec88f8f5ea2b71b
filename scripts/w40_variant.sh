#!/bin/bash
# Local A/B aid: scripts/w40_variant.sh <name> [-DFLAG ...] builds cd_dynamax_amd/lib/ab/libcdkf_<name>.so = the current library with
# launch_w40.hip recompiled under the given flags.  Time several variants in ONE gpurun call (box-to-box variance is ~20 %):
#   for v in cd_dynamax_amd/lib/ab/*.so; do CDKF_LIB_PATH=$PWD/$v python scripts/run_config.py config4 1; done
set -e
name=$1; shift
cd "$(dirname "$0")/../cd_dynamax_amd/csrc"
mkdir -p ../lib/ab
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "$@" -Rpass-analysis=kernel-resource-usage -c launch_w40.hip -o /tmp/launch_w40_$name.o 2>&1 | grep -E "error|VGPRs Spill" || true
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libcdkf_$name.so $(ls ../../build/csrc/*.o | grep -v launch_w40) /tmp/launch_w40_$name.o -lhiprtc -ldl
