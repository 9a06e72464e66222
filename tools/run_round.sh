#!/bin/bash
# One GPU-box pass that produces everything profiles/<TAG>_* is made of:
#   bash tools/run_round.sh TAG     (then locally: python3 tools/summarize_pmc.py TAG cns_N4_512x512)
TAG=${1:-r02}
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/$TAG
bash tools/profile_round.sh $TAG > gpurun_out/$TAG/profile.log 2>&1; tail -3 gpurun_out/$TAG/profile.log
python3 tools/summarize_pmc.py $TAG cns_N4_512x512 > gpurun_out/$TAG/pmc.log 2>&1; tail -6 gpurun_out/$TAG/pmc.log
python bench.py > gpurun_out/$TAG/bench_default.json 2> gpurun_out/$TAG/bench_default.err
tail -c 3500 gpurun_out/$TAG/bench_default.json
