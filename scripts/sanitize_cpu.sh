#!/bin/bash
# CPU sanitizer runs (AddressSanitizer + UndefinedBehaviorSanitizer; never on the GPU box: gpurun refuses GPU sanitizers).
#  1. the C restatement of the reference algorithm (oracle/cdkf_oracle.c) under gcc's ASan + UBSan, driven by tests/test_oracle_c.py;
#  2. the host side of the C ABI -- argument checking, the parameter-slot ring, the TCP rendezvous (cdkf_api.hip, cdkf_comm.hip) --
#     built with clang's ASan + UBSan on the HOST code only (-fno-gpu-sanitize: the device code objects are the shipped ones), driven
#     by tests/test_abi.py and the library-rendezvous tests of tests/test_distributed.py (no GPU needed: they end at the first HIP call).
# Usage: scripts/sanitize_cpu.sh [oracle|host|all]      (exit code != 0 on any sanitizer report or test failure)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
WHAT=${1:-all}
export ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:halt_on_error=1
export UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
if [ "$WHAT" = oracle ] || [ "$WHAT" = all ]; then
  make -C $ROOT/oracle -s OUT=_build/asan CFLAGS="-O1 -g -fno-omit-frame-pointer -fsanitize=address,undefined -fno-sanitize-recover=all -fopenmp -fPIC -std=c11 -Wall -Wno-unknown-pragmas" \
       CC="gcc" >/dev/null
  # (the Makefile's link line has no -fsanitize: relink with the runtime)
  gcc -shared -fopenmp -fsanitize=address,undefined -o $ROOT/oracle/_build/asan/libcdkf_oracle.so $ROOT/oracle/_build/asan/oracle_f64.o $ROOT/oracle/_build/asan/oracle_f32.o -lm
  LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) CDKF_ORACLE_SO=$ROOT/oracle/_build/asan/libcdkf_oracle.so \
    python -m pytest $ROOT/tests/test_oracle_c.py -x -q -p no:cacheprovider
  echo "sanitize_cpu: oracle/cdkf_oracle.c clean under ASan + UBSan"
fi
if [ "$WHAT" = host ] || [ "$WHAT" = all ]; then
  B=$ROOT/build/csrc_asan
  mkdir -p $B
  F="-O1 -g -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -Wno-pass-failed -fsanitize=address,undefined -fno-gpu-sanitize -fno-omit-frame-pointer"
  for tu in cdkf_api cdkf_comm; do
    hipcc $F -c $ROOT/cd_dynamax_amd/csrc/$tu.hip -o $B/$tu.o
  done
  hipcc --offload-arch=gfx950 -shared -fPIC -fsanitize=address,undefined -fno-gpu-sanitize -o $B/libcdkf_hip_asan.so \
        $B/cdkf_api.o $B/cdkf_comm.o $(ls $ROOT/build/csrc/*.o | grep -v -e cdkf_api.o -e cdkf_comm.o) -lhiprtc -ldl
  RT=$(hipcc -print-file-name=libclang_rt.asan-x86_64.so 2>/dev/null || true)
  [ -f "$RT" ] || RT=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
  LD_PRELOAD=$RT CDKF_LIB_PATH=$B/libcdkf_hip_asan.so \
    python -m pytest $ROOT/tests/test_abi.py $ROOT/tests/test_distributed.py -x -q -p no:cacheprovider -k "not graft_entry and not gloo"
  echo "sanitize_cpu: host side of the C ABI (cdkf_api.hip, cdkf_comm.hip) clean under ASan + UBSan"
fi
