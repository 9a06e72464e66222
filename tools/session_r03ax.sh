#!/bin/bash
# round-3 GPU session AX: kh_project with coalesced trace stores through a wave-private LDS block (main) against 8-B stores at a stride of
# 40 B (variant khnoco): same-box A/B of the hex workload, bitwise comparison, hex GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ax; mkdir -p $O
bash tools/ab_hex.sh khnoco > $O/ab_hex.log 2>&1; cat $O/ab_hex.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
out = []
for N in (1, 2, 3):
    rd, md, ops, Q = bench.build_hex_problem(N, 7, 5, 6, 0, 7 * 5 * 6, 0.0, True)
    eng = engine.RhsEngine(rd, md, ops, engine.EULER_HEX_COLLOCATED, lf_scale=0.25)
    out.append(np.stack(eng.download(eng.rhs(eng.upload(Q)))).ravel())
np.save(sys.argv[1], np.concatenate(out))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/khnoco.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("hex N=1, 2, 3 (LF on) 7x5x6 (210 elements: a partial last workgroup): coalesced == strided stores bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_hex.py -m gpu -q -x > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
