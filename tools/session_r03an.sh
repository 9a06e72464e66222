#!/bin/bash
# round-3 GPU session AN: kt2_sigma with the IQ / face-extrapolation rows in LDS (three waves per SIMD at N=4) against the rows in
# registers (variant sigma_rows_reg): same-box A/B, bitwise comparison of the two builds, hex A/B of the kh_project entry priority
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03an; mkdir -p $O
bash tools/ab_variants.sh sigma_rows_reg > $O/ab_cns.log 2>&1; cat $O/ab_cns.log
python - <<'PY' 2>&1 | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
rd, md, ops, Q = bench.build_problem(4, 256, 256, 0, 256 * 256, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
r = eng.download(eng.rhs(eng.upload(Q)))
np.save(sys.argv[1], np.stack(r))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/sigma_rows_reg.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("cns 256x256 rhs: rows in LDS == rows in registers bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "not cfg3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
bash tools/ab_hex.sh sigma_rows_reg > $O/ab_hex.log 2>&1; cat $O/ab_hex.log
