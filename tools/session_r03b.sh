#!/bin/bash
# round-3 GPU session B: attribution A/B (window build = all global addresses folded into L2-resident windows; spec build),
# micro-benchmark outputs, the new headline-size parity tests (with the binary128 truth at 512^2 once)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03b; mkdir -p $O
bash tools/ab_variants.sh r02 window spec > $O/ab.log 2>&1; cat $O/ab.log
for f in euler; do
  for v in main window; do
    if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
    echo -n "$f 256x256 $v: "
    timeout -k 10 300 python bench.py --no-cpu-baseline --formulation $f --kx 256 --ky-per-gpu 256 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
  done
done 2>&1 | tee $O/ab_euler.log
unset ESDG_HIP_LIB
{
  echo "== tools/ubench/hbm_bw.py"; timeout -k 10 120 python tools/ubench/hbm_bw.py
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/fma64 tools/ubench/fma64.hip && { echo "== tools/ubench/fma64.hip"; timeout -k 10 120 /tmp/fma64; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_read tools/ubench/lds_read.hip && { echo "== tools/ubench/lds_read.hip"; timeout -k 10 120 /tmp/lds_read; }
} > $O/ubench.txt 2>&1; tail -5 $O/ubench.txt
ESDG_TRUTH_512=1 timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -s -k "cfg2 or cfg3 or 64" --durations=5 > $O/pytest_cfg.log 2>&1; echo "pytest cfg rc=$?"; grep -a "cfg\|passed\|failed\|slowest\|call" $O/pytest_cfg.log | tail -20
cp gpurun_out/parity_errors.json $O/parity_errors_cfg.json
