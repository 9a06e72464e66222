#!/bin/bash
# round-3 GPU session Q: neighbour-trace logs taken lazily at the interface flux: parity subset + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03q; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "not cfg3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
bash tools/ab_variants.sh r03p > $O/ab.log 2>&1; cat $O/ab.log
