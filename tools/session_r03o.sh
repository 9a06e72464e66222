#!/bin/bash
# round-3 GPU session O: experiment -DESDG_EXP_HALF1 (phase 0 writes and the last phase reads only the first trace half; logs, wavespeed, energy rebuilt)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03o; mkdir -p $O
bash tools/ab_variants.sh half1 > $O/ab.log 2>&1; cat $O/ab.log
ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/half1.so timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cns_modal or euler_collocated or cfg2" 2>&1 | tail -4
