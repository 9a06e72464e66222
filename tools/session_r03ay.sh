#!/bin/bash
# round-3 GPU session AY: kt2_sigma stores a group's normal-stress records through an LDS block, 1 KB of consecutive doubles per store instruction (main), against 8-B stores at a stride of 24 B (variant nocoal):
# same-box A/B, bitwise comparison of the two builds, parity subset, other degrees and Euler cfg2
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03ay; mkdir -p $O
bash tools/ab_variants.sh nocoal > $O/ab_cns.log 2>&1; cat $O/ab_cns.log
python - <<'PY' 2>&1 | tee $O/bitwise.log
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
e["ESDG_HIP_LIB"] = os.path.abspath("esdg_cns_amd/variants/nocoal.so")
subprocess.check_call([sys.executable, "/tmp/dump.py", "/tmp/b.npy"], env=e)
import numpy as np
a, b = np.load("/tmp/a.npy"), np.load("/tmp/b.npy")
print("cns 253x131 rhs: coalesced == strided B stores bit for bit:", bool(np.array_equal(a, b)), "max |diff|", float(np.abs(a - b).max()))
PY
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -q -x -k "not cfg3" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc"; tail -3 $O/pytest.log
for a in "--N 3" "--N 2" "--N 6 --kx 256 --ky-per-gpu 256" "--formulation euler --kx 256 --ky-per-gpu 256"; do for v in main nocoal; do if [ $v = main ]; then unset ESDG_HIP_LIB; else export ESDG_HIP_LIB=$PWD/esdg_cns_amd/variants/$v.so; fi; echo -n "$a $v: "; timeout -k 10 300 python bench.py --no-cpu-baseline $a 2>/dev/null | python3 -c "
import sys, json
r = json.loads(sys.stdin.read().strip().splitlines()[-1])
print(\"ms_per_step %.4f  phases %s\" % (r[\"ms_per_step\"], \" \".join(\"%.4f\" % p for p in r[\"roofline\"][\"phase_ms\"])))"; done; done 2>&1 | tee $O/ab_other.log
