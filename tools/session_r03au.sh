#!/bin/bash
# round-3 GPU session AU: degree-generic hex kernel kh_rhs_g with every volume-face flux evaluated once (N = 4, 5; main) against the
# row-wise form (variant hexg_rowwise): same-box A/B, bitwise comparison, hex GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03au; mkdir -p $O
for rep in 1 2; do for a in "--N 4 --kx 32 --kz-per-gpu 16" "--N 5 --kx 32 --kz-per-gpu 8" "--N 4 --kx 32 --kz-per-gpu 16 --hex-geometry element"; do for v in main hexg_rowwise; do if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi; echo -n "$a $v: "; timeout -k 10 300 python bench.py --formulation hex --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"ms_per_step %.4f  phases %s  value %.3e\" % (r[\"ms_per_step\"], \" \".join(\"%.4f\" % p for p in r[\"roofline\"][\"phase_ms\"]), r[\"value\"]))"; done; done; done 2>&1 | tee $O/ab.log
unset ESDG_HIP_LIB
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
out = []
for N, pn in ((4, True), (5, True), (4, False)):
    rd, md, ops, Q = bench.build_hex_problem(N, 6, 5, 4, 0, 6 * 5 * 4, 0.0, pn)
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.25)
    out.append(np.stack(eng.download(eng.rhs(eng.upload(Q)))).ravel())
np.save(sys.argv[1], np.concatenate(out))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/hexg_rowwise.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("hex N=4, 5 (LF on, geometry modes 2 and 0) 6x5x4: once == row-wise bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
