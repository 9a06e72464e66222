#!/bin/bash
# A/B of two builds on the SAME GPU box (box-to-box variance is about +-3 %):
#   esdg_cns_amd/libesdg_hip.so      = candidate ("new")
#   esdg_cns_amd/libesdg_hip_alt.so  = baseline  ("base"), e.g. built from `git stash` / another commit
#   bash tools/ab.sh [bench.py args...]
cd "$GRAFT_REPO_ROOT/esdg_cns_amd" || exit 1
cp libesdg_hip.so /tmp/ab_new.so
for v in new base new base; do
  if [ $v = base ]; then cp libesdg_hip_alt.so libesdg_hip.so; else cp /tmp/ab_new.so libesdg_hip.so; fi
  echo -n "$v: "
  (cd .. && timeout -k 10 300 python bench.py --no-cpu-baseline "$@" 2>/dev/null | grep -o "ms_per_step[^,]*\|kernel_ms[^,]*" | tr "\n" " ")
  echo
done
cp /tmp/ab_new.so libesdg_hip.so
