#!/bin/bash
# round-3 GPU session AQ: phase 1 and the last phase concurrently on two streams over strips (tools/strip_concurrent.py), persistent-grid sizes of kt2_sigma
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03aq; mkdir -p $O
timeout -k 10 300 python tools/strip_concurrent.py 32 64 128 2>&1 | grep -v amdgpu.ids | tee $O/default.log
for w in 2 1; do ESDG_T2_WG_PER_CU=$w timeout -k 10 300 python tools/strip_concurrent.py 32 64 128 2>&1 | grep -v amdgpu.ids | tee $O/wg$w.log; done
