#!/bin/bash
# round-3 GPU session AA: hex flux with wave-uniform log-mean variants (rho and beta forms chosen independently) against the
# per-lane selects (variants/hexold.so = the build before): hex parity tests, then same-box ms per RHS, three geometry modes
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03aa; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py -m gpu -q -x > $O/pytest_hex.log 2>&1; rc=$?
echo "pytest hex rc=$rc"; tail -3 $O/pytest_hex.log
[ $rc = 0 ] || exit $rc
{ echo "== per-node geometry (mode 2)"; bash tools/ab_hex.sh hexold; echo "== element record (mode 0)"; bash tools/ab_hex.sh hexold --hex-geometry element; echo "== N=2"; bash tools/ab_hex.sh hexold --N 2; } 2>&1 | tee $O/ab_hex.log
