#!/bin/bash
# round-3 GPU session AG: profile set of HEAD (tools/session_r03m.sh TAG) + the cavity viscous-alone probes with the final kernels
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03w}
bash tools/session_r03m.sh $TAG
O=gpurun_out/$TAG
for a in "4 8 8 1" "4 16 16 1" "4 8 8 2" "3 6 5 1"; do
  python tools/cavity_visc_probe.py $a 2>&1 | grep -v "amdgpu.ids\|^ \[\|node errors\|same element" > "$O/probe_${a// /_}.log"; grep -a "oracle  \|^v2\|^round" "$O/probe_${a// /_}.log" | cut -c1-150
done
