#!/bin/bash
# round-3 final GPU pass, part 2: rocprofv3 + PMC passes of euler (cfg2) and hex (cfg5 per GPU), SQ counters of cfg3, the binary128 truth at 512^2
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03z}
O=gpurun_out/$TAG; mkdir -p $O
bash tools/profile_other_configs.sh $TAG
bash tools/pmc_sq.sh $TAG > $O/pmc_sq.log 2>&1; tail -25 $O/pmc_sq.log
ESDG_TRUTH_512=1 timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "cfg3" > $O/pytest_truth512.log 2>&1; echo "truth512 rc=$?"; grep -a "cfg3\|512x512\|passed\|failed" $O/pytest_truth512.log | tail -5
cp gpurun_out/parity_errors.json $O/parity_errors_truth512.json
