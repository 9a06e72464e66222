#!/bin/bash
# round-3 GPU session AM: wave issue priority at the two ends of the one-shot kernels (-DESDG_PRIO_ENTRY / -DESDG_PRIO_EXIT): same-box A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03am; mkdir -p $O
bash tools/ab_variants.sh prio_e3 prio_x3 prio_e3x2 prio_e1x3 > $O/ab_cns.log 2>&1; cat $O/ab_cns.log
for v in prio_x3 prio_e3x2; do bash tools/ab_hex.sh $v > $O/ab_hex_$v.log 2>&1; cat $O/ab_hex_$v.log; done
