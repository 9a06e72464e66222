#!/bin/bash
# ns per element over mesh shapes / sizes on one GPU (same box):  bash tools/shape_sweep.sh "512 512" "509 512" ...
cd "$GRAFT_REPO_ROOT" || exit 1
[ $# -eq 0 ] && set -- "512 512" "2048 128" "1024 256" "2048 256" "1024 512" "512 1024" "256 2048"
for s in "$@"; do
  set -- $s
  echo -n "kx=$1 ky=$2: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --kx $1 --ky-per-gpu $2 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
K = r['config']['elements']
print('ms %.4f  ns/el %.3f  phases %s' % (r['ms_per_step'], r['ms_per_step']*1e6/K, ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done
