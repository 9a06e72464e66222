#!/bin/bash
# round-3 GPU session AT: accumulator plane sets used twice also at N1 = 4 (N = 3: LDS 16.4 -> 12.3 KB = 12 instead of 9 waves per CU; main)
# against the N1 = 5 collocated case only (variant reuse5only): same-box A/B at N = 3 (CNS 512^2, Euler 512^2, cfg1), bitwise, tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03at; mkdir -p $O
for rep in 1 2; do for a in "--N 3" "--N 3 --formulation euler" "--N 3 --formulation euler --kx 16 --ky-per-gpu 16"; do for v in main reuse5only; do if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi; echo -n "$a $v: "; timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"ms_per_step %.4f  phases %s\" % (r[\"ms_per_step\"], \" \".join(\"%.4f\" % p for p in r[\"roofline\"][\"phase_ms\"])))"; done; done; done 2>&1 | tee $O/ab.log
unset ESDG_HIP_LIB
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
out = []
for form, F in (("cns", engine.CNS_MODAL), ("euler", engine.EULER_COLLOCATED)):
    rd, md, ops, Q = bench.build_problem(3, 250, 130, 0, 250 * 130, form)
    eng = engine.RhsEngine(rd, md, ops, F)
    out.append(np.stack(eng.download(eng.rhs(eng.upload(Q)))).ravel())
np.save(sys.argv[1], np.concatenate(out))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/reuse5only.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("N=3 cns + euler 250x130 rhs: sets used twice == separate sets bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "not cfg3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
