#!/bin/bash
# round-3 GPU session W: regression check over degrees and workloads: HEAD against the round-2 library (built from commit b39c9c8)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03w; mkdir -p $O
run() {  # label, bench args...
  local label=$1; shift
  for v in main r02; do
    if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
    echo -n "$label $v: "
    timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
  done
  unset ESDG_HIP_LIB
}
{
run "cns N=4 512x512"
run "cns N=3 512x512" --N 3
run "cns N=2 512x512" --N 2
run "cns N=5 384x384" --N 5 --kx 384 --ky-per-gpu 384
run "cns N=6 256x256" --N 6 --kx 256 --ky-per-gpu 256
run "euler N=4 256x256" --formulation euler --kx 256 --ky-per-gpu 256
run "euler N=3 512x512" --formulation euler --N 3
run "hex N=3 128x128x16 (r02: element record)" --formulation hex
run "hex N=2 128x128x16" --formulation hex --N 2
} 2>&1 | tee $O/regression.log
