#!/usr/bin/env python3
"""ISA-level attribution of one kernel's instruction stream to source buckets (no GPU needed).

    python tools/isa_buckets.py [smooth|rough] [--kernel kt3|project|sigma] [--n1 5] [-D...]

Compiles ONE explicit instantiation of the kernel for gfx950 with line tables (`-gline-tables-only`) and with the wave-uniform
switches of the production kernel turned into constants (`-DESDG_T3_ATTR=1|2 -DESDG_T2_FORCE_MODE=1|2`: a smooth wave -- every
flux all-series, no logarithms -- or a rough one -- every flux logarithmic), so that the listing is straight-line code and the
static count IS the executed count of such a wave.  Every instruction is attributed to the source line of its `.loc` (the
innermost inlined function) and summed into the buckets below, split into fp64 arithmetic / other VALU / LDS / VMEM / SALU.
What the production build executes on top of a forced-mode build (the per-flux series tests and ballots of ec_flux_dir, the
moves at the joins of its three variants) is reported by the SQ counters (profiles/r0x_sq_counters.txt), not here.
"""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "esdg_cns_amd", "csrc")

FP64 = re.compile(r"^v_(fma|fmac|mul|add|rcp|rsq|frexp_mant|frexp_exp_i32|ldexp|max|min|fract|floor|rndne|trunc|ceil|div_\w+|sqrt|cvt_f64_\w+|cvt_\w+_f64|cmp\w*|cmpx\w*)_f64")
FP64_ALT = re.compile(r"^v_(cvt_f64_i32|cvt_f64_u32|cvt_f64_f32|cvt_i32_f64|cvt_f32_f64|cmp_\w+_f64|cmpx_\w+_f64|frexp_exp_i32_f64|cmp_class_f64)")


def classify(op):
    if op.startswith("v_"):
        if FP64.match(op) or FP64_ALT.match(op):
            return "fp64"
        return "valu_other"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "vmem"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_"):
        return "salu"
    return None


def function_ranges(path):
    """(first line, last line, name) of the device functions / lambdas of a header, by a light scan for `name(` ... `{` at depth of a definition."""
    out = []
    txt = open(path).read().split("\n")
    pat = re.compile(r"^\s*(?:template\s*<[^>]*>\s*)?(?:__device__|__host__|static|inline|__forceinline__|constexpr|\s)*[\w:<>\*&\s]+?\b(\w+)\s*\([^;]*$")
    i = 0
    n = len(txt)
    while i < n:
        line = txt[i]
        m = pat.match(line) if ("__device__" in line or "__forceinline__" in line) else None
        if m and "{" in "".join(txt[i:i + 4]) and not line.strip().endswith(";"):
            # find the opening brace and its match
            depth = 0
            j = i
            started = False
            while j < n:
                for ch in txt[j]:
                    if ch == "{":
                        depth += 1
                        started = True
                    elif ch == "}":
                        depth -= 1
                if started and depth == 0:
                    break
                j += 1
            if started:
                out.append((i + 1, j + 1, m.group(1)))
                i = j + 1
                continue
        i += 1
    return out


# buckets of the kernel body itself, by marker comments in the source (first line that contains the marker starts the bucket)
KT3_MARKERS = [
    ("entry: ids, addresses, loads", "const unsigned tid = threadIdx.x;"),
    ("staging -> LDS", "// ---- staging: geometry, tables, nodal values"),
    ("Vq", "if (MODAL) {\n#pragma unroll\n    for (int r = 0; r < NR; ++r) {\n      sA[slot[r]]"),
    ("primitives, smooth test, node logs", "bool ok = true;"),
    ("line stage: set-up", "// ---- line stage"),
    ("face turn: trace rest, penalty, interface flux, LF", "auto face_turn = [&]"),
    ("face turn: volume-face pairs", "if (inviscid) {   // (uniform)\n        const double wt = sTab[TL.WTF"),
    ("line stage: set-up", "face_turn(0, fA, qMA"),
    ("volume-volume pairs", "if (inviscid) {   // volume-volume pairs of the line"),
    ("SG loads", "// viscous volume divergence of the wave's nodes"),
    ("line -> node exchange (projection, lift)", "{   // collocated projection and lift along the line"),
    ("node rounds: -(r0+r1)/J, viscous divergence", "// ---- node rounds: rhs at the Gauss nodes"),
    ("Pq", "double out[NR][4];"),
    ("store / RK epilogue", "if (STG || ESDG_T3_PIN_OUT)"),
]


def marker_lines(src_path, markers):
    txt = open(src_path).read()
    res = []
    for name, mk in markers:
        pos = txt.find(mk)
        if pos < 0:
            raise SystemExit("marker not found in %s: %r" % (src_path, mk[:50]))
        res.append((txt.count("\n", 0, pos) + 1, name))
    res.sort()
    return res


def main():
    args = sys.argv[1:]
    mode = "smooth"
    n1 = 5
    defs = []
    keep = None
    i = 0
    while i < len(args):
        a = args[i]
        if a in ("smooth", "rough", "production"):
            mode = a
        elif a == "--n1":
            n1 = int(args[i + 1]); i += 1
        elif a == "--keep":
            keep = args[i + 1]; i += 1
        elif a.startswith("-D"):
            defs.append(a)
        i += 1
    tmp = tempfile.mkdtemp(prefix="isa_")
    stub = os.path.join(tmp, "stub.hip")
    with open(stub, "w") as f:
        f.write('#define ESDG_T3_NO_DISPATCH\n#include "esdg_kernels_tensor3.hip"\n'
                "template __global__ void esdg::t3::kt3_rhs<%d, true, true, false, false>(esdg::TensorTables, esdg::MeshDev, esdg::Phys, "
                "const double*, const double*, const double*, const double*, double*, esdg::LsrkFuse, esdg::StageFuse);\n" % n1)
    flags = {"smooth": ["-DESDG_T3_ATTR=1", "-DESDG_T2_FORCE_MODE=1"], "rough": ["-DESDG_T3_ATTR=2", "-DESDG_T2_FORCE_MODE=2"], "production": []}[mode]
    asm = os.path.join(tmp, "k.s")
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-gline-tables-only", "-I" + CSRC] + flags + defs + ["-o", asm, stub]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    if keep:
        subprocess.run(["cp", asm, keep])
    lines = open(asm).read().split("\n")
    files = {}
    for ln in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', ln)
        if m:
            files[int(m.group(1))] = os.path.basename(m.group(3))
    src3 = os.path.join(CSRC, "esdg_kernels_tensor3.hip")
    kmarks = marker_lines(src3, KT3_MARKERS)
    hdr_ranges = {h: function_ranges(os.path.join(CSRC, h)) for h in ("esdg_devmath.hpp", "esdg_t2_physics.hpp")}

    def bucket(fname, line):
        if fname == "merged":
            return "fn ec_flux_core + series + rcp (merged line info)"
        if fname == "esdg_kernels_tensor3.hip":
            name = "kernel body (before the first marker)"
            for l0, nm in kmarks:
                if line >= l0:
                    name = nm
            return name
        if fname in hdr_ranges:
            for a, b, nm in hdr_ranges[fname]:
                if a <= line <= b:
                    return "fn " + nm
            return "hdr " + fname
        return "lib " + fname

    # the kernel's instructions
    start = next(i for i, ln in enumerate(lines) if re.match(r"^_ZN4esdg2t37kt3_rhs\w+:", ln))
    cur = ("esdg_kernels_tensor3.hip", 0)
    tab = collections.defaultdict(lambda: collections.Counter())
    ops = collections.defaultdict(lambda: collections.Counter())
    for ln in lines[start:]:
        s = ln.strip()
        m = re.match(r"\.loc\s+(\d+)\s+(\d+)", s)
        if m:
            cur = (files.get(int(m.group(1)), "?"), int(m.group(2)))
            if cur[1] == 0:   # merged locations (code common to the arms of a uniform branch): the call site of the inline chain
                c = re.search(r"@\[\s*\S*?([\w.]+):(\d+)", s)
                cur = ("merged", int(c.group(2))) if c else cur
            continue
        if not s or s.startswith((".", ";", "_Z")) or s.endswith(":"):
            continue
        op = s.split()[0]
        c = classify(op)
        if c:
            b = bucket(*cur)
            tab[b][c] += 1
            ops[c][op] += 1
        if op == "s_endpgm":
            break
    tot = collections.Counter()
    for b in tab:
        tot.update(tab[b])
    print("kernel kt3_rhs<%d,1,1,0,0>, %s wave, flags %s" % (n1, mode, " ".join(flags + defs)))
    print("%-58s %6s %6s %6s %5s %5s %5s" % ("bucket", "VALU", "fp64", "other", "LDS", "VMEM", "SALU"))
    for b, c in sorted(tab.items(), key=lambda kv: -(kv[1]["fp64"] + kv[1]["valu_other"])):
        v = c["fp64"] + c["valu_other"]
        print("%-58s %6d %6d %6d %5d %5d %5d" % (b, v, c["fp64"], c["valu_other"], c["lds"], c["vmem"], c["salu"]))
    v = tot["fp64"] + tot["valu_other"]
    print("%-58s %6d %6d %6d %5d %5d %5d" % ("TOTAL", v, tot["fp64"], tot["valu_other"], tot["lds"], tot["vmem"], tot["salu"]))
    print("\nnon-fp64 VALU by opcode:", ", ".join("%s %d" % kv for kv in ops["valu_other"].most_common(14)))
    print("fp64 VALU by opcode:", ", ".join("%s %d" % kv for kv in ops["fp64"].most_common(12)))
    m = re.search(r"\.vgpr_count:\s+(\d+)", "\n".join(lines))
    print("vgprs", m.group(1) if m else "?")


if __name__ == "__main__":
    main()
