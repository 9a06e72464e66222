#!/bin/bash
# Memory operations, waits and barriers of one kernel in hipcc's assembly (first line that matches the name pattern):
#   bash tools/asm_waits.sh csrc-file.hip 'kt2_rhsILi5ELb1ELb1ELb0' [extra hipcc flags]
# An `s_waitcnt vmcnt(0)` near the top of a kernel means some load was sunk into a branch and the wave waits there for every
# load in flight (round 3: the geometry / operator staging of kt2_rhs).
src=$1; pat=$2; shift 2
out=/tmp/asm_$(basename $src .hip).s
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only "$@" -o $out $src 2>/dev/null || exit 1
awk -v pat="^_Z[A-Za-z0-9_]*${pat}[A-Za-z0-9_]*:" '$0 ~ pat {on=1} on {print} on && /s_endpgm/ {exit}' $out > /tmp/asm_kernel.s
echo "$(wc -l < /tmp/asm_kernel.s) lines -> /tmp/asm_kernel.s"
grep -n "global_load\|global_store\|buffer_load\|s_waitcnt vmcnt\|s_barrier\|scratch_" /tmp/asm_kernel.s
