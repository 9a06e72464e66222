"""Build libesdg_hip.so in-tree with hipcc for gfx950 (MI355X).

One object per source under esdg_cns_amd/build/ (recompiled when the source or any header is newer), compiled in
parallel, then linked into esdg_cns_amd/libesdg_hip.so together with librccl (the halo transport of esdg_comm_* /
esdg_halo_exchange lives inside the library).  `python -m esdg_cns_amd.build [-DNAME ...] [--out path.so]` builds a
variant (A/B builds under esdg_cns_amd/variants/).

The shipped library reads no environment variable.  build() also links esdg_cns_amd/libesdg_hip_ab.so: the SAME kernel
objects with csrc/esdg_api.hip compiled once more under -DESDG_AB_HOOKS, which is where the environment switches that select
partner kernels, geometry modes and schedule variants live (tests and tools that compare kernel sets load that build:
esdg_cns_amd._lib.lib(ab=True), engine.RhsEngine(..., ab_hooks=True))."""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/esdg_kernels.hip", "csrc/esdg_kernels_tensor2.hip", "csrc/esdg_kernels_tensor3.hip", "csrc/esdg_kernels_hex.hip", "csrc/esdg_kernels_err.hip",
           "csrc/esdg_api.hip", "csrc/esdg_setup.cpp"]
HEADERS = ["csrc/esdg_dev.hpp", "csrc/esdg_devmath.hpp", "csrc/esdg_t2_physics.hpp", "csrc/esdg_tensor_tables.hpp", "csrc/esdg_hex_tables.hpp",
           "../include/esdg_hip.h"]
OUT = os.path.join(HERE, "libesdg_hip.so")
OUT_AB = os.path.join(HERE, "libesdg_hip_ab.so")
AB_SOURCE = "csrc/esdg_api.hip"   # the only file that reads ESDG_AB_HOOKS
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"]
LINK = ["-L" + os.path.join(ROCM, "lib"), "-lrccl", "-Wl,-rpath," + os.path.join(ROCM, "lib")]


def _listed(names):
    missing = [n for n in names if not os.path.exists(os.path.join(HERE, n))]
    if missing:   # a typo in a file name must not silently drop the file from the build
        raise FileNotFoundError("esdg_cns_amd/build.py lists files that do not exist: " + ", ".join(missing))
    return list(names)


def _sources():
    return _listed(SOURCES)


def _headers():
    return _listed(HEADERS)


def needs_build(out=OUT):
    if not os.path.exists(out) or (out == OUT and not os.path.exists(OUT_AB)):
        return True
    t = os.path.getmtime(out)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in _sources() + _headers())


def build(force=False, verbose=False, defines=(), out=OUT, tag=""):
    if not (force or needs_build(out)):
        return out
    hipcc = os.environ.get("HIPCC", os.path.join(ROCM, "bin", "hipcc"))
    odir = os.path.join(HERE, "build", tag or "main")
    os.makedirs(odir, exist_ok=True)
    hdr_t = max(os.path.getmtime(os.path.join(HERE, h)) for h in _headers())
    jobs = []
    for s in _sources():
        src = os.path.join(HERE, s)
        obj = os.path.join(odir, os.path.basename(s) + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_t):
            jobs.append([hipcc] + FLAGS + list(defines) + ["-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        subprocess.check_call(cmd, cwd=HERE)

    with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4) or 1) as ex:
        list(ex.map(run, jobs))
    os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
    objs = [os.path.join(odir, os.path.basename(s) + ".o") for s in _sources()]
    run([hipcc, "--offload-arch=gfx950", "-shared", "-o", out] + objs + LINK)
    if out == OUT and not defines:   # the A/B build beside the shipped one: same objects, the API file once more with its hooks
        ab_obj = os.path.join(odir, os.path.basename(AB_SOURCE) + ".ab.o")
        run([hipcc] + FLAGS + ["-DESDG_AB_HOOKS", "-c", os.path.join(HERE, AB_SOURCE), "-o", ab_obj])
        api_obj = os.path.join(odir, os.path.basename(AB_SOURCE) + ".o")
        run([hipcc, "--offload-arch=gfx950", "-shared", "-o", OUT_AB] + [ab_obj if o == api_obj else o for o in objs] + LINK)
    return out


if __name__ == "__main__":
    args = sys.argv[1:]
    out, tag = OUT, ""
    if "--out" in args:
        i = args.index("--out")
        out = os.path.abspath(args[i + 1])
        tag = os.path.splitext(os.path.basename(out))[0]
        del args[i:i + 2]
    build(force=True, verbose=True, defines=args, out=out, tag=tag)
