#!/bin/bash
# round-3 GPU session X: viscous trace record (rho, v2, v3, v4): full suite (viscous-alone ratios) + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03x; mkdir -p $O
timeout -k 10 1200 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
cp gpurun_out/parity_errors.json $O/parity_errors.json
python3 - <<'PY'
import json
d = json.load(open("gpurun_out/parity_errors.json"))
v = [(r["e_gpu"] / max(r["e_orc"], 1e-300), r["case"]) for r in d if "viscous" in r["case"]]
print("viscous-alone cases: e_gpu/e_orc", " ".join("%.2f" % x for x, _ in sorted(v)))
w = [(r["e_gpu"] / max(r["e_orc"], 1e-300), r["case"]) for r in d if r["e_gpu"] > 1e-12 and "viscous" not in r["case"]]
print("others: max", max(w))
for r in d:
    if "512" in r["case"] or "256x256" in r["case"] or "64x64" in r["case"]: print(r["case"], "%.3e %.3e" % (r["e_gpu"], r["e_orc"]))
PY
bash tools/ab_variants.sh r03u > $O/ab.log 2>&1; cat $O/ab.log
