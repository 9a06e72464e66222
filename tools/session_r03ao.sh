#!/bin/bash
# round-3 GPU session AO: noise floor of the variant A/B (dummy = the main sources with an unused define) and kh_project's entry priority on / off
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ao; mkdir -p $O
for v in dummy khp0; do bash tools/ab_hex.sh $v > $O/ab_hex_$v.log 2>&1; cat $O/ab_hex_$v.log; done
bash tools/ab_variants.sh dummy > $O/ab_cns.log 2>&1; cat $O/ab_cns.log
