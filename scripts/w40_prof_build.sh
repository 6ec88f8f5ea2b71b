#!/bin/bash
# Local diagnostic: build libcdkf_hip_prof.so = the library with launch_w40.hip compiled under -DCDKF_W40_PROFILE (per-phase cycle
# counters of the d = 40 sweeps, printed by trajectory 0).  Run with CDKF_LIB_PATH=.../libcdkf_hip_prof.so.
set -e
cd "$(dirname "$0")/../cd_dynamax_amd/csrc"
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -DCDKF_W40_PROFILE -c launch_w40.hip -o /tmp/launch_w40_prof.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcdkf_hip_prof.so $(ls ../../build/csrc/*.o | grep -v launch_w40) /tmp/launch_w40_prof.o -lhiprtc -ldl
