#!/bin/bash
# CPU-side sanitizer runs (GPU ASan / XNACK are unavailable on the pool).  Run from the repo root.
#  1. the C oracle under ASan + UBSan
#  2. the Ed448 field / point / scalar code the kernels run (__host__ __device__), built for the host, under ASan + UBSan
set -e
make -C oracle libcapyoracle_asan.so
LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so) ASAN_OPTIONS=detect_leaks=0 \
    python tests/sanitizers/asan_oracle.py
hipcc -O1 -g -std=c++17 -fPIC -shared -x hip --offload-arch=gfx950 --cuda-host-only -fsanitize=address,undefined \
    -o /tmp/libed448host_asan.so tests/native/ed448_host_test.cpp
LD_PRELOAD=$(/opt/rocm/lib/llvm/bin/clang --print-file-name=libclang_rt.asan-x86_64.so) ASAN_OPTIONS=detect_leaks=0 \
    python tests/sanitizers/asan_ed448_host.py
