#!/bin/bash
# round-3 GPU session Y: kt2_project (v2 phase 0) against the round-1 kt_project (ESDG_V1=project), same box: 2D parity
# tests first, then ms per RHS / per phase for cfg3 (CNS 512^2 N=4), cfg2 (Euler 256^2 N=4) and other degrees
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03y; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
[ $rc = 0 ] || exit $rc
ab() {
  for v in new p0v1 new p0v1; do
    if [ $v = p0v1 ]; then export ESDG_V1=project; else unset ESDG_V1; fi
    echo -n "$1 $v: "
    timeout -k 10 300 python bench.py --no-cpu-baseline "${@:2}" 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
  done
  unset ESDG_V1
}
ab cfg3 | tee $O/ab_cfg3.log
ab cfg2 --formulation euler --kx 256 --ky-per-gpu 256 | tee $O/ab_cfg2.log
ab cnsN3 --N 3 | tee $O/ab_cnsN3.log
ab cnsN5 --N 5 --kx 384 --ky-per-gpu 384 | tee $O/ab_cnsN5.log
ab cnsN7 --N 7 --kx 256 --ky-per-gpu 256 | tee $O/ab_cnsN7.log
