#!/bin/bash
# Same-box A/B of library variants (esdg_cns_amd/variants/*.so, built with `python -m esdg_cns_amd.build -D... --out ...`)
# against the main build: ms per RHS and per-phase kernel times.   bash tools/ab_variants.sh name1 name2 ...
cd "$GRAFT_REPO_ROOT" || exit 1
for rep in 1 2; do
for v in main "$@"; do
  if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
  echo -n "$v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s  rough %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms']), '%.4f' % r['ms_per_step_rough_state'] if r.get('ms_per_step_rough_state') else '-'))"
done
done
