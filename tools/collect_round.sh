#!/bin/bash
# Copies what the `round1 TAG` / `round2 TAG` steps of tools/gpu_session.sh left under gpurun_out/ into profiles/ (tracked): kernel stats, PMC passes and bench lines
# of the three workloads (-> pmc_traffic.json), the GPU suite log, the parity records.   bash tools/collect_round.sh TAG
TAG=${1:?tag}
cd "$(dirname "$0")/.." || exit 1
bash tools/collect_profiles.sh $TAG cns_N4_512x512
for w in euler:euler_N4_256x256 hex:hex_N3_128x128x16; do
  n=${w%%:*}; k=${w##*:}; S=gpurun_out/prof_${TAG}_$n
  cp $S/kernel_stats.csv profiles/${TAG}_${n}_${k#*_}_kernel_stats.csv
  for c in fetch write valu; do [ -f $S/${c}_counters.csv ] && cp $S/${c}_counters.csv profiles/${TAG}_${n}_${c}_counters.csv; done
  tail -1 gpurun_out/${TAG}_$n/bench_default.json > profiles/${TAG}_${n}_bench.json
  python3 tools/summarize_pmc.py ${TAG}_$n $k
done
cp gpurun_out/$TAG/pytest_gpu.log profiles/${TAG}_pytest_gpu.log
cp gpurun_out/$TAG/smoke.log profiles/${TAG}_smoke.log
for f in sq_counters hex_sq_counters euler_sq_counters; do [ -f gpurun_out/$TAG/$f.txt ] && cp gpurun_out/$TAG/$f.txt profiles/${TAG}_$f.txt; done
python3 - "$TAG" <<'PY'
import json, sys
tag = sys.argv[1]
rows = json.load(open(f"gpurun_out/{tag}/parity_errors.json"))
extra = f"gpurun_out/{tag}/parity_errors_truth512.json"
try:
    rows += [r for r in json.load(open(extra)) if r["case"] not in {x["case"] for x in rows}]
except OSError:
    pass
json.dump(rows, open(f"profiles/parity_{tag[:3]}.json", "w"), indent=1)
w = [(r["e_gpu"] / max(r["e_orc"], 1e-300), r["case"]) for r in rows if r["e_gpu"] > 1e-12 and "viscous" not in r["case"]]
print(len(rows), "parity records; largest e_gpu/e_orc among those with e_gpu > 1e-12 (viscous-alone diagnostics aside):", max(w))
PY
