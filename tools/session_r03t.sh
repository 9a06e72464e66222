#!/bin/bash
# round-3 GPU session T: where a wave of the current kernels spends its life (s_memtime stamps), and the compute-only times (window build) of HEAD
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03t; mkdir -p $O
ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/stamp.so timeout -k 10 300 python tools/stamps.py 2>&1 | grep -v amdgpu.ids | tee $O/stamps.log
bash tools/ab_variants.sh window > $O/ab_window.log 2>&1; cat $O/ab_window.log
