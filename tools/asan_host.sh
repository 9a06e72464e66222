#!/bin/bash
# AddressSanitizer + UBSan over the HOST code of the library (set-up, halo plan, create-time validation); CPU only --
# GPU ASan is not available on the pool.  Run from the repo root in the build container:
#   bash tools/asan_host.sh
set -e
T=/tmp/esdg_asan; mkdir -p $T
# 1. the stand-alone set-up code with gcc's sanitizers (leak check on)
g++ -fsanitize=address,undefined -fno-omit-frame-pointer -g -O1 -Iinclude -x c++ tools/asan_setup.c esdg_cns_amd/csrc/esdg_setup.cpp -o $T/asan_setup
ASAN_OPTIONS=detect_leaks=1 $T/asan_setup | tail -2
# 2. the whole library with host-side ASan/UBSan (device code unsanitized), swapped in for the host-only tests
# (source list and link flags come from esdg_cns_amd/build.py, so a new source file or library cannot be forgotten here)
SRCS=$(python -c "from esdg_cns_amd import build as b; print(' '.join(b._sources()))")
LINK=$(python -c "from esdg_cns_amd import build as b; print(' '.join(b.LINK))")
(cd esdg_cns_amd && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O1 -g -std=c++17 -fPIC -shared -fsanitize=address,undefined \
   -fno-gpu-sanitize -shared-libsan -o $T/libesdg_hip.so $SRCS $LINK)
cp esdg_cns_amd/libesdg_hip.so $T/keep.so
trap 'cp $T/keep.so esdg_cns_amd/libesdg_hip.so; touch esdg_cns_amd/libesdg_hip.so' EXIT
cp $T/libesdg_hip.so esdg_cns_amd/libesdg_hip.so
RT=$(find /opt/rocm/lib/llvm -name "libclang_rt.asan-x86_64.so" | head -1)
LD_PRELOAD=$RT ASAN_OPTIONS=detect_leaks=0 python -m pytest tests/test_halo.py tests/test_setup_c.py tests/test_abi.py -x -q
