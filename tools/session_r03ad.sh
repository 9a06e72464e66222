#!/bin/bash
# round-3 GPU session AD: kt2_sigma with the nodal-basis viscous operators in the elements that touch a wall:
# viscous-alone probe (8x8, 16x16, BCTYPE 1 and 2), the 2D parity + engine tests, cavity timing against the element-record build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ad; mkdir -p $O
for a in "4 8 8 1" "4 16 16 1" "4 8 8 2" "3 6 5 1"; do
  python tools/cavity_visc_probe.py $a 2>&1 | grep -v "amdgpu.ids\|^ \[\|node errors\|same element" > "$O/probe_${a// /_}.log"; grep -a "oracle  \|^v2\|^round" "$O/probe_${a// /_}.log" | cut -c1-150
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_drivers.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -5 $O/pytest.log
for v in node element; do
  if [ $v = element ]; then export ESDG_WALL_GEOMETRY=element; else unset ESDG_WALL_GEOMETRY; fi
  for rep in 1 2; do echo -n "cavity N=4 256x256, wall geometry $v: "; python tools/cavity_ab.py --child 4 256 /tmp/cav_$v.npy 2>/dev/null | tail -1; done
done 2>&1 | tee $O/cavity_ab.log
unset ESDG_WALL_GEOMETRY
python -c "
import numpy as np
a, b = np.load('/tmp/cav_node.npy'), np.load('/tmp/cav_element.npy')
print('max rel difference node vs element geometry: %.2e' % (np.abs(a - b).max() / np.abs(b).max()))" | tee -a $O/cavity_ab.log
