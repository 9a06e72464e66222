#!/bin/bash
# e_gpu / e_orc (tools/parity_truth.py) for the production build and, when present, the accuracy-attribution builds in
# esdg_cns_amd/variants/ (-DESDG_IEEE_DIV, -DESDG_LIBM_LOG, -ffp-contract=off), one process each, on the same box.
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out/parity
timeout -k 10 400 python3 tools/parity_truth.py gpurun_out/parity/base.json > gpurun_out/parity/base.log 2>&1 || { tail -20 gpurun_out/parity/base.log; exit 1; }
for v in esdg_cns_amd/variants/*.so; do
  [ -f "$v" ] || continue
  n=$(basename $v .so)
  ESDG_HIP_LIB=$PWD/$v timeout -k 10 400 python3 tools/parity_truth.py gpurun_out/parity/$n.json > gpurun_out/parity/$n.log 2>&1 || exit 1
done
cat gpurun_out/parity/base.log
