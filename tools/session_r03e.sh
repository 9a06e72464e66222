#!/bin/bash
# round-3 GPU session E: hex geometry fidelity (element record vs per-node arrays): errors against the truth, and time
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03e; mkdir -p $O
timeout -k 10 900 python tools/hex_geometry_probe.py 8 16 24 2>&1 | grep -v amdgpu.ids | tee $O/hex_geometry_probe.log
for mode in element pernode element pernode; do
  if [ $mode = pernode ]; then export ESDG_HEX_PER_NODE=1; a="--hex-per-node"; else unset ESDG_HEX_PER_NODE; a=""; fi
  echo -n "hex 128x128x16 $mode: "
  timeout -k 10 400 python bench.py --no-cpu-baseline --formulation hex $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('ms_per_step %.4f  phases %s' % (r['ms_per_step'], ' '.join('%.4f' % p for p in r['roofline']['phase_ms'])))"
done 2>&1 | tee $O/hex_time.log
unset ESDG_HEX_PER_NODE
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "cavity_64 or unshifted" 2>&1 | tail -3
