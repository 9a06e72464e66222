#!/bin/bash
# Copies the summaries of a tools/run_round.sh pass from gpurun_out/ (scratch) into profiles/ (tracked):
#   bash tools/collect_profiles.sh TAG [KEY]
TAG=${1:?tag}; KEY=${2:-cns_N4_512x512}
cd "$(dirname "$0")/.." || exit 1
S=gpurun_out/prof_$TAG
cp $S/kernel_stats.csv profiles/${TAG}_${KEY}_kernel_stats.csv
for c in fetch write valu; do [ -f $S/${c}_counters.csv ] && cp $S/${c}_counters.csv profiles/${TAG}_${c}_counters.csv; done
[ -f gpurun_out/$TAG/bench_default.json ] && tail -1 gpurun_out/$TAG/bench_default.json > profiles/${TAG}_bench.json
python3 tools/summarize_pmc.py $TAG $KEY
