#!/bin/bash
# round-3 GPU session AE: cost of the nodal-basis wall path on cavity meshes (node = default, element = ESDG_WALL_GEOMETRY=element)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ae; mkdir -p $O
python tools/cavity_visc_probe.py 4 8 8 1 2>&1 | grep -a "oracle  \|^v2\|^round" | cut -c1-150
for cfg in "4 256" "4 64" "3 256" "4 512"; do
  set -- $cfg
  for v in node element node element; do
    if [ $v = element ]; then export ESDG_WALL_GEOMETRY=element; else unset ESDG_WALL_GEOMETRY; fi
    echo -n "cavity N=$1 $2x$2, wall geometry $v: "; python tools/cavity_ab.py --child $1 $2 /tmp/cav_$v.npy 2>/dev/null | tail -1
  done
done 2>&1 | tee $O/cavity_ab.log
