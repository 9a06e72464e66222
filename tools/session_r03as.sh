#!/bin/bash
# round-3 GPU session AS: collocated Euler kt2_rhs at N=4 with the accumulator plane sets used twice (18 KB LDS, 4 waves per SIMD; main)
# against four sets (26 KB, 3 waves; variant noreuse): same-box A/B at cfg2 and 512x512, bitwise comparison, Euler parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03as; mkdir -p $O
for rep in 1 2; do for a in "--formulation euler --kx 256 --ky-per-gpu 256" "--formulation euler"; do for v in main noreuse; do if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi; echo -n "$a $v: "; timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"ms_per_step %.4f  phases %s\" % (r[\"ms_per_step\"], \" \".join(\"%.4f\" % p for p in r[\"roofline\"][\"phase_ms\"])))"; done; done; done 2>&1 | tee $O/ab_euler.log
unset ESDG_HIP_LIB
python - <<'PY' 2>&1 | grep -v amdgpu.ids | tee $O/bitwise.log
import os, subprocess, sys
code = '''
import sys, numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import bench
from esdg_cns_amd import engine
rd, md, ops, Q = bench.build_problem(4, 256, 256, 0, 256 * 256, "euler")
eng = engine.RhsEngine(rd, md, ops, engine.EULER_COLLOCATED)
r = eng.download(eng.rhs(eng.upload(Q)))
np.save(sys.argv[1], np.stack(r))
'''
open("/tmp/dump.py", "w").write(code)
e = dict(os.environ)
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/a.npy"], env=e)
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/noreuse.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("euler 256x256 rhs: sets used twice == four sets bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "euler or ranged or shard" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
