#!/bin/bash
# per-kernel average durations of a bench.py run:  bash tools/kstats.sh TAG [bench args]
set -e
TAG=${1:-k}
EXTRA="${@:2}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
OUT=gpurun_out/kstats_$TAG
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-rough-state $EXTRA > $OUT/trace.log 2>&1
find $OUT -name "*kernel_stats.csv" -exec cp {} $OUT/kernel_stats.csv \;
python3 - <<PY
import csv
for r in csv.DictReader(open("$OUT/kernel_stats.csv")):
    if "esdg" in r["Name"]:
        print(f'{r["Name"].split("(")[0][:60]:60s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:9.1f} us')
PY
