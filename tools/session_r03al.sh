#!/bin/bash
# round-3 GPU session AL: per-node J after Pq in kt2_rhs for wall elements: viscous-alone probes 8^2 ... 128^2, the 2D GPU tests,
# cavity timings against ESDG_WALL_GEOMETRY=element
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03al; mkdir -p $O
for k in 8 16 32 64 128; do
  timeout -k 10 900 python tools/cavity_visc_probe.py 4 $k $k 1 2>&1 | grep -v "amdgpu.ids\|^ \[\|node errors\|same element" > $O/probe_4_${k}.log; grep -a "oracle  \|^v2 " $O/probe_4_${k}.log | cut -c1-150
done
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_drivers.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -3 $O/pytest.log
for cfg in "4 64" "4 256" "3 256" "4 512"; do
  set -- $cfg
  for v in node element node element; do
    if [ $v = element ]; then export ESDG_WALL_GEOMETRY=element; else unset ESDG_WALL_GEOMETRY; fi
    echo -n "cavity N=$1 $2x$2, wall geometry $v: "; python tools/cavity_ab.py --child $1 $2 /tmp/cav_$v.npy 2>/dev/null | tail -1
  done
done 2>&1 | tee $O/cavity_ab.log
