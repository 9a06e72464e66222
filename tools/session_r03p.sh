#!/bin/bash
# round-3 GPU session P: 32-B trace record (logs / energy / wavespeed rebuilt by the consumer): full suite + A/B (cns, euler)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03p; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
bash tools/ab_variants.sh r03m > $O/ab.log 2>&1; cat $O/ab.log
for v in main r03m main r03m; do
  if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
  echo -n "euler256 $v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --formulation euler --kx 256 --ky-per-gpu 256 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/ab_other.log
