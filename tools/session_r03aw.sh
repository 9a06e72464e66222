#!/bin/bash
# round-3 GPU session AW: CNS kt2_rhs at N=4 with four waves per SIMD (variant cns4w = -DESDG_T2_ACC_REUSE=7: accumulator sets used twice,
# B / SG and the lift rows loaded after the volume-face rounds, 128 VGPRs with 4 spills) against the three-wave kernel (main)
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03aw; mkdir -p $O
bash tools/ab_variants.sh cns4w > $O/ab_cns.log 2>&1; cat $O/ab_cns.log
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
rd, md, ops, Q = bench.build_problem(4, 253, 131, 0, 253 * 131, "cns")
eng = engine.RhsEngine(rd, md, ops, engine.CNS_MODAL)
r = eng.download(eng.rhs(eng.upload(Q)))
np.save(sys.argv[1], np.stack(r))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/cns4w.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("cns 253x131 rhs: four-wave variant == main bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
