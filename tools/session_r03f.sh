#!/bin/bash
# round-3 GPU session F: float-difference normals in the v2 kernels (phase 0 back on the face means): suite + A/B against the fp64-normal build
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -5 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
bash tools/ab_variants.sh r03d wpe3 > $O/ab.log 2>&1; cat $O/ab.log
