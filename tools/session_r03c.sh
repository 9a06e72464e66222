#!/bin/bash
# round-3 GPU session C: suite after the prologue rewrites (kt_project, kh_project, kh_rhs), A/B against the previous build,
# kt2_sigma occupancy variants, hex / euler benches, parity scaling probe
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_parity.py::test_cfg2_euler_256_exact_size_values_against_oracle_and_truth --deselect tests/test_gpu_parity.py::test_cfg3_cns_512_exact_size_values_against_oracle > $O/pytest_gpu.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -4 $O/pytest_gpu.log
[ $rc = 0 ] || exit $rc
fi
bash tools/ab_variants.sh r03a wpe2 wpe3 wpe4 > $O/ab.log 2>&1; cat $O/ab.log
for f in hex euler; do
  for v in main r03a main r03a; do
    if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi
    echo -n "$f $v: "
    if [ $f = euler ]; then a="--kx 256 --ky-per-gpu 256"; else a=""; fi
    timeout -k 10 300 python bench.py --no-cpu-baseline --formulation $f $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
  done
done 2>&1 | tee $O/ab_other.log
unset ESDG_HIP_LIB
timeout -k 10 600 python tools/parity_scaling.py euler 256 2>&1 | grep -v amdgpu.ids | tee $O/parity_scaling_euler.log
