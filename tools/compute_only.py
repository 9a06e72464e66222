#!/usr/bin/env python3
"""Compute-only phase times (GPU box): bench.py on a build whose every global address is folded into an L2-resident window
(`python -m esdg_cns_amd.build -DESDG_EXP_WINDOW=1023 --out esdg_cns_amd/variants/win.so` -- wrong results, same instruction
stream), per workload -> gpurun_out/compute_only.json (copied to profiles/compute_only.json, which bench.py reads as
`roofline.compute_only_ms` for the kernel sources it was measured on).

    python tools/compute_only.py [cns] [euler] [hex]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

WORK = {"cns": (["--formulation", "cns"], "cns_N4_512x512"),
        "euler": (["--formulation", "euler", "--kx", "256", "--ky-per-gpu", "256"], "euler_N4_256x256"),
        "hex": (["--formulation", "hex"], "hex_N3_128x128x16")}


def run(extra, lib=None):
    env = dict(os.environ)
    if lib:
        env["ESDG_HIP_LIB"] = lib
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-rough-state"] + extra,
                         env=env, cwd=ROOT, capture_output=True, text=True, timeout=600).stdout
    r = json.loads(out.strip().splitlines()[-1])
    return r["roofline"]["phase_ms"], r["ms_per_step"]


def main():
    win = os.path.join(ROOT, "esdg_cns_amd", "variants", "win.so")
    if not os.path.exists(win):
        raise SystemExit("missing " + win)
    res = {}
    for w in (sys.argv[1:] or ["cns"]):
        extra, key = WORK[w]
        full, ms = run(extra)
        comp, msc = run(extra, win)
        res[key] = {"kernel_src_sha": bench.kernel_source_hash(), "phase_ms": comp, "phase_ms_full_same_box": full,
                    "ms_per_step_full_same_box": ms, "ms_per_step_compute_only": msc,
                    "how": "-DESDG_EXP_WINDOW=1023: every global address folded into the first 1024 elements (L2-resident)"}
        print(key, "full", ["%.4f" % x for x in full], "compute-only", ["%.4f" % x for x in comp], flush=True)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", "compute_only.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
