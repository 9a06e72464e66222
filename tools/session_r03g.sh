#!/bin/bash
# round-3 GPU session G: hex geometry modes (0 element record, 1 full per-node arrays, 2 record + 8-bit differences): tests, errors, time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03g; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py tests/test_gpu_drivers.py -m gpu -q -s > $O/pytest_hex.log 2>&1; rc=$?; echo "pytest hex rc=$rc"; tail -5 $O/pytest_hex.log; grep -a "e_orc.*mode" $O/pytest_hex.log
for mode in 2 0 1 2 0; do
  unset ESDG_HEX_PER_NODE ESDG_HEX_GEOMETRY
  if [ $mode = 1 ]; then export ESDG_HEX_PER_NODE=1; fi
  if [ $mode = 0 ]; then export ESDG_HEX_GEOMETRY=element; fi
  echo -n "hex 128x128x16 mode $mode: "
  timeout -k 10 400 python bench.py --no-cpu-baseline --formulation hex 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/hex_time.log
unset ESDG_HEX_PER_NODE ESDG_HEX_GEOMETRY
timeout -k 10 900 python tools/hex_geometry_probe.py 24 2>&1 | grep -v amdgpu.ids | tee $O/hex_geometry_probe.log
