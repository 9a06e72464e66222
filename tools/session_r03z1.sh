#!/bin/bash
# round-3 final GPU pass, part 1: full GPU suite log, smoke, rocprofv3 + PMC passes and the default bench line for cns (cfg3)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03z}
O=gpurun_out/$TAG; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 $O/smoke.log
bash tools/run_round.sh $TAG
