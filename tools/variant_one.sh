#!/bin/bash
# A/B variant of the library that differs from the in-tree build in ONE source file's compile flags:
#   bash tools/variant_one.sh NAME esdg_kernels_tensor3.hip -DESDG_T3_PREFETCH=1024 [...]   -> esdg_cns_amd/variants/NAME.so
# (the other objects are those of the main build, esdg_cns_amd/build/main/ -- run `python -m esdg_cns_amd.build` first)
set -e
cd "$(dirname "$0")/../esdg_cns_amd"
name=$1; src=$2; shift 2
mkdir -p build/$name variants
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c csrc/$src -o build/$name/$src.o
objs=""
for b in $(python3 -c "import build; print(' '.join(__import__('os').path.basename(s) + '.o' for s in build.SOURCES))"); do   # (the objects build.py links)
  if [ "$b" = "$src.o" ]; then objs="$objs build/$name/$src.o"; else objs="$objs build/main/$b"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o variants/$name.so $objs -L/opt/rocm/lib -lrccl -Wl,-rpath,/opt/rocm/lib
echo "built esdg_cns_amd/variants/$name.so"
