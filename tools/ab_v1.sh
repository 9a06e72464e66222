#!/bin/bash
# Same-box A/B of the round-1 tensor kernels (ESDG_V1=1) against the current ones: ms per RHS and per-phase kernel times.
#   bash tools/ab_v1.sh [bench.py args...]
cd "$GRAFT_REPO_ROOT" || exit 1
for v in new v1 new v1; do
  if [ $v = v1 ]; then export ESDG_V1=1; else unset ESDG_V1; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done
