#!/bin/bash
# round-3 GPU session J: traces laid out by face (MeshDev::KN1): full suite + A/B against the element-major layout
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03j; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
bash tools/ab_variants.sh r03i > $O/ab.log 2>&1; cat $O/ab.log
for v in main r03i main r03i; do
  if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
  echo -n "euler256 $v: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --formulation euler --kx 256 --ky-per-gpu 256 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/ab_other.log
