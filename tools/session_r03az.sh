#!/bin/bash
# round-3 GPU session AZ: coalesced trace stores also in the degree-generic hex phase 0 (kh_project_g; main) vs strided (khnoco): A/B at N = 4, 5, hex tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03az; mkdir -p $O
for rep in 1 2; do for a in "--N 4 --kx 32 --kz-per-gpu 16" "--N 5 --kx 32 --kz-per-gpu 8"; do for v in main khnoco; do if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi; echo -n "$a $v: "; timeout -k 10 300 python bench.py --formulation hex --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"ms_per_step %.4f  phases %s  value %.3e\" % (r[\"ms_per_step\"], \" \".join(\"%.4f\" % p for p in r[\"roofline\"][\"phase_ms\"]), r[\"value\"]))"; done; done; done 2>&1 | tee $O/ab.log
unset ESDG_HIP_LIB
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
