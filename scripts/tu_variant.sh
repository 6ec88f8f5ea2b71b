#!/bin/bash
# Local A/B aid: scripts/tu_variant.sh <name> "<tu> [<tu> ...]" [flags ...] builds cd_dynamax_amd/lib/ab/libcdkf_<name>.so = the current
# library with the named translation units (e.g. "launch_ekf launch_grad") recompiled under the given flags.  Time the variants in ONE
# gpurun call:  for v in cd_dynamax_amd/lib/ab/*.so; do CDKF_LIB_PATH=$PWD/$v python scripts/gpu_time_grad.py bench; done
set -e
name=$1; tus=$2; shift 2
cd "$(dirname "$0")/../cd_dynamax_amd/csrc"
mkdir -p ../lib/ab
objs=""; skip=""
for tu in $tus; do
  hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed "$@" -c $tu.hip -o /tmp/${tu}_$name.o 2>&1 | grep -E "error" || true
  objs="$objs /tmp/${tu}_$name.o"; skip="$skip|$tu\.o"
done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/ab/libcdkf_$name.so $(ls ../../build/csrc/*.o | grep -Ev "${skip#|}") $objs -lhiprtc -ldl
