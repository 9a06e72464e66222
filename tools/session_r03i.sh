#!/bin/bash
# round-3 GPU session I: full suite with the degree-generic hex kernels; quick time of a N=4 / N=5 hex box
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03i; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -6 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
for N in 4 5; do
  echo -n "hex N=$N 32x32x16: "
  timeout -k 10 400 python bench.py --no-cpu-baseline --formulation hex --N $N --kx 32 --kz-per-gpu 16 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s  DOF/s %.3e' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms']), r['value']))"
done 2>&1 | tee $O/hex_highN.log
