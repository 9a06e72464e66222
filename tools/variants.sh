#!/bin/bash
# Same-box comparison of several builds: every esdg_cns_amd/variants/*.so is swapped in for libesdg_hip.so in turn
# (two passes), bench.py gives ms_per_step, a rocprofv3 kernel trace the per-kernel averages.
#   bash tools/variants.sh [bench.py args...]
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
cp esdg_cns_amd/libesdg_hip.so /tmp/variants_keep.so
for pass in 1 2; do
  for v in esdg_cns_amd/variants/*.so; do
    cp "$v" esdg_cns_amd/libesdg_hip.so
    ms=$(timeout -k 10 120 python3 bench.py --no-cpu-baseline "$@" 2>/dev/null | grep -o '"ms_per_step": [0-9.]*' | cut -d' ' -f2)
    line="$(basename $v .so) pass$pass ms_per_step=$ms"
    if [ -z "$ms" ]; then echo "$line FAILED -- stopping"; cp /tmp/variants_keep.so esdg_cns_amd/libesdg_hip.so; exit 1; fi
    if [ $pass = 1 ]; then
      OUT=gpurun_out/var_$(basename $v .so); rm -rf $OUT; mkdir -p $OUT
      timeout -k 10 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py --steps 30 --warmup 3 --no-cpu-baseline "$@" > $OUT/log 2>&1
      f=$(find $OUT -name "*kernel_stats.csv" | head -1)
      [ -n "$f" ] && line="$line $(python3 - "$f" <<'PY'
import csv, sys
out = []
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if n.startswith("void esdg::k") or "esdg::k" in n:
        out.append(f'{n.split("esdg::")[1].split("<")[0]}={float(r["AverageNs"])/1e3:.1f}us')
print(" ".join(sorted(set(out))))
PY
)"
    fi
    echo "$line"
  done
done
cp /tmp/variants_keep.so esdg_cns_amd/libesdg_hip.so
