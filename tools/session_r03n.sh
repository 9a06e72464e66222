#!/bin/bash
# round-3 GPU session N: hex kh_rhs with the split last face round: hex tests + A/B; cfg3 test with the 512^2 truth by default
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03n; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py tests/test_gpu_drivers.py -m gpu -q -x > $O/pytest_hex.log 2>&1; rc=$?; echo "pytest hex rc=$rc"; tail -3 $O/pytest_hex.log
for v in main r03m main r03m; do
  if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
  echo -n "hex $v: "
  timeout -k 10 400 python bench.py --no-cpu-baseline --formulation hex 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/hex_ab.log
unset ESDG_HIP_LIB
ESDG_HEX_GEOMETRY=element timeout -k 10 400 python bench.py --no-cpu-baseline --formulation hex 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('hex main, element record: ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))" | tee -a $O/hex_ab.log
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -k cfg3 --durations=3 2>&1 | tail -6
