#!/bin/bash
# Same-box comparison of the production build with every esdg_cns_amd/variants/*.so (ESDG_HIP_LIB): ms per RHS and the live
# per-phase kernel times of bench.py, two passes.   bash tools/ab_variants.sh [bench.py args...]
cd "$GRAFT_REPO_ROOT" || exit 1
for pass in 1 2; do
  for v in main esdg_cns_amd/variants/*.so; do
    if [ "$v" = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/$v; fi
    echo -n "$(basename $v .so): "
    timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
  done
done
