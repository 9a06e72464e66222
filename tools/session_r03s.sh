#!/bin/bash
# round-3 GPU session S: kt2_rhs skips the partner record's log plane in all-series waves: parity subset (bitwise vs the previous build) + A/B
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03s; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "not cfg3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
bash tools/ab_variants.sh r03r > $O/ab.log 2>&1; cat $O/ab.log
python - <<'PY'
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
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/r03r.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("cns 256x256 rhs: new build == previous build bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
