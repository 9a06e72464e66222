#!/bin/bash
# round-3 GPU session L: kt2_sigma with its stores deferred to the next iteration: suite + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03l; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q -x > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
bash tools/ab_variants.sh r03k > $O/ab.log 2>&1; cat $O/ab.log
