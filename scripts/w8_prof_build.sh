#!/bin/bash
# Local diagnostic: build libcdkf_hip_prof.so = the library with launch_w8.hip / launch_adj.hip compiled under -DCDKF_W8_PROFILE
# (per-phase cycle counters of the state_dim <= 8 sweeps, printed by trajectory 0).  Run with CDKF_LIB_PATH=.../libcdkf_hip_prof.so.
set -e
cd "$(dirname "$0")/../cd_dynamax_amd/csrc"
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -DCDKF_W8_PROFILE"
hipcc $F -c launch_w8.hip -o /tmp/launch_w8_prof.o &
hipcc $F -c launch_adj.hip -o /tmp/launch_adj_prof.o &
wait
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcdkf_hip_prof.so $(ls ../../build/csrc/*.o | grep -v -e launch_w8 -e launch_adj) /tmp/launch_w8_prof.o /tmp/launch_adj_prof.o -lhiprtc -ldl
