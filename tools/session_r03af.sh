#!/bin/bash
# round-3 GPU session AF: full GPU suite with the nodal-basis wall path (viscous-alone gate at 2 x everywhere), sharded bitwise
# check on wall meshes, periodic A/B against the library before the change (variants/prewall.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03af; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?
echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
[ $rc = 0 ] || exit $rc
bash tools/ab_variants.sh prewall 2>&1 | tee $O/ab_periodic.log
