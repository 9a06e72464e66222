"""Build libesdg_hip.so in-tree with hipcc for gfx950 (MI355X)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
SOURCES = ["csrc/esdg_kernels.hip", "csrc/esdg_kernels_tensor.hip", "csrc/esdg_kernels_hex.hip", "csrc/esdg_kernels_err.hip",
           "csrc/esdg_api.hip", "csrc/esdg_setup.cpp"]
HEADERS = ["csrc/esdg_dev.hpp", "csrc/esdg_devmath.hpp", "csrc/esdg_tensor_tables.hpp", "csrc/esdg_hex_tables.hpp", "../include/esdg_hip.h"]
OUT = os.path.join(HERE, "libesdg_hip.so")


def needs_build():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(os.path.join(HERE, f)) > t for f in SOURCES + HEADERS)


def build(force=False, verbose=False):
    if not (force or needs_build()):
        return OUT
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-o", OUT] + SOURCES
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd, cwd=HERE)
    return OUT


if __name__ == "__main__":
    build(force=True, verbose=True)
