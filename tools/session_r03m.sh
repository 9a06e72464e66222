#!/bin/bash
# round-3 GPU session M: the profile set of HEAD: full GPU suite log, smoke, rocprofv3 + PMC passes for cns (cfg3), euler (cfg2), hex (cfg5 per
# GPU), default bench lines; the binary128 truth at 512^2 once (ESDG_TRUTH_512=1)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03m}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/run_round.sh $TAG
bash tools/profile_other_configs.sh $TAG
ESDG_TRUTH_512=1 timeout -k 10 1500 python -m pytest tests/test_gpu_parity.py -m gpu -q -s -k "cfg3" > $O/pytest_truth512.log 2>&1; echo "truth512 rc=$?"; grep -a "cfg3\|512x512\|passed\|failed" $O/pytest_truth512.log | tail -5
cp gpurun_out/parity_errors.json $O/parity_errors_truth512.json
