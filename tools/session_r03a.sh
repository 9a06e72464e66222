#!/bin/bash
# round-3 GPU session A: parity suite, same-box A/B (main vs round-2 build vs no-guess build), cache-residency check, strip pipeline
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03a; mkdir -p $O
timeout -k 10 600 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
bash tools/ab_variants.sh r02 nospec > $O/ab.log 2>&1 && cat $O/ab.log
for ky in 512 128 64 48 32; do
  echo -n "resident ky=$ky: "
  timeout -k 10 300 python bench.py --no-cpu-baseline --ky-per-gpu $ky 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
K = r['config']['elements']
print('ms_per_step %.4f  ns/elem %.3f  phases(ns/elem) %s' % (r['ms_per_step'], r['ms_per_step'] * 1e6 / K, ' '.join('%.3f' % (p * 1e6 / K) for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/resident.log
timeout -k 10 400 python tools/strip_pipeline.py 16 32 64 128 2>&1 | tee $O/strip_pipeline.log
