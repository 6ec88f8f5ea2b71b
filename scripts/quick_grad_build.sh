#!/bin/bash
# dev helper: rebuild launch_grad.o only and relink (the Makefile rebuilds every object when a header changes)
set -e
cd "$(dirname "$0")/../cd_dynamax_amd/csrc"
B=../../build/csrc
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-pass-failed "$@" -c launch_grad.hip -o $B/launch_grad.o
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/libcdkf_hip.so $B/cdkf_api.o $B/cdkf_comm.o $B/launch_ekf.o $B/launch_ukf.o $B/launch_eks.o $B/launch_wg.o $B/launch_w40.o $B/launch_adj.o $B/launch_grad.o $B/launch_custom.o -lhiprtc -ldl
touch $B/*.o ../lib/libcdkf_hip.so
