#!/bin/bash
# Same-box A/B of the hex workload: main build vs esdg_cns_amd/variants/NAME.so.   bash tools/ab_hex.sh NAME [bench args]
cd "$GRAFT_REPO_ROOT" || exit 1
V=$1; shift
for rep in 1 2; do
for v in main $V; do
  if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --formulation hex --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done
done
