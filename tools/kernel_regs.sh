#!/bin/bash
# Register / spill / LDS table of the kernels of one source file (device-only compile, no GPU needed):
#   bash tools/kernel_regs.sh esdg_kernels_tensor3.hip [name-filter] [extra hipcc flags...]
set -e
cd "$(dirname "$0")/../esdg_cns_amd/csrc"
src=$1; flt=${2:-}; shift; shift || true
tmp=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only "$@" -c "$src" -o $tmp/dev.o 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$tmp/dev.o --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=$tmp/gfx.o
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $tmp/gfx.o > $tmp/notes.txt
python3 - "$tmp/notes.txt" "$flt" <<'PY'
import re, subprocess, sys
txt = open(sys.argv[1]).read()
for k in txt.split('.agpr_count:')[1:]:
    name = re.search(r'\.name:\s+(\S+)', k).group(1)
    g = lambda f: re.search(r'\.' + f + r':\s+(\d+)', k).group(1)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().split('(')[0].replace('void esdg::', '')
    ag = re.match(r'\s+(\d+)', k).group(1)
    if sys.argv[2] in dem:
        print("%-60s vgpr %3s agpr %3s spill %3s sgpr %3s lds %6s" % (dem, g('vgpr_count'), ag, g('vgpr_spill_count'), g('sgpr_count'), g('group_segment_fixed_size')))
PY
rm -rf $tmp
