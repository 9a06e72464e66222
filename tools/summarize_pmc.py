"""Summarises the FETCH_SIZE / WRITE_SIZE passes of tools/profile_round.sh into profiles/pmc_traffic.json.

Units and gfx950 corrections follow MI355X_MICROARCH.md section "HBM": rocprofv3 reports FETCH_SIZE / WRITE_SIZE
in KiB; on gfx950 FETCH_SIZE counts 128-B fabric requests as 64 B for wide coalesced streaming reads (x2
correction), WRITE_SIZE is exact for 16-B-per-lane streaming stores; other access widths are uncalibrated.
Our kernels read the state with 8-B-per-lane coalesced loads and the traces with 16-B loads, so both the raw
and the x2-corrected read figures are recorded and the corrected one is what bench.py reports as `traffic`."""
import collections
import csv
import json
import os
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
key = sys.argv[2] if len(sys.argv) > 2 else "cns_N4_512x512"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")


def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


f = per_kernel(os.path.join(src, "fetch_counters.csv"), "FETCH_SIZE")
w = per_kernel(os.path.join(src, "write_counters.csv"), "WRITE_SIZE")
# fp64 VALU work: wave-instructions by kind (one pass); flops = 64 lanes x (ADD + MUL + TRANS + 2 FMA).  Lanes switched
# off by EXEC still occupy their issue slot, so this is the issued fp64 work the 78.6 TFLOP/s vector peak is quoted for.
valu = {}
vpath = os.path.join(src, "valu_counters.csv")
if os.path.exists(vpath):
    parts = {c: per_kernel(vpath, c) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                               "SQ_INSTS_VALU_TRANS_F64", "SQ_INSTS_VALU", "SQ_WAVES")}
    for k in parts["SQ_INSTS_VALU"]:
        a, m, fm, t = (parts[c].get(k, 0.0) for c in ("SQ_INSTS_VALU_ADD_F64", "SQ_INSTS_VALU_MUL_F64", "SQ_INSTS_VALU_FMA_F64",
                                                       "SQ_INSTS_VALU_TRANS_F64"))
        valu[k] = {"insts_valu": parts["SQ_INSTS_VALU"][k], "insts_f64": a + m + fm + t, "waves": parts["SQ_WAVES"].get(k, 0.0),
                   "fp64_flops": 64.0 * (a + m + t + 2 * fm)}
out = {}
for k in sorted(set(f) | set(w)):
    if "esdg::" not in k:
        continue
    rd_raw, wr = f.get(k, 0.0) * 1024, w.get(k, 0.0) * 1024
    out[k] = {"fetch_bytes_raw": rd_raw, "fetch_bytes_x2": 2 * rd_raw, "write_bytes": wr,
              "hbm_bytes_per_launch": 2 * rd_raw + wr}
    out[k].update(valu.get(k, {}))
dst = os.path.join(root, "profiles", "pmc_traffic.json")
allj = json.load(open(dst)) if os.path.exists(dst) else {}
rhs = [v for k, v in out.items() if "_rhs<" in k or "_rhs_l<" in k]
# average kernel durations of the --kernel-trace --stats pass of the same command (profiles/<tag>_*_kernel_stats.csv)
stats = os.path.join(src, "kernel_stats.csv")
avg_us = {}
if os.path.exists(stats):
    for r in csv.DictReader(open(stats)):
        n = r["Name"].split("(")[0].replace("void ", "")
        if "esdg::" in n:
            avg_us[n] = float(r["AverageNs"]) / 1e3
for k in out:
    if k in avg_us:
        out[k]["rocprofv3_avg_us"] = avg_us[k]
rhs_name = [k for k in out if "_rhs<" in k or "_rhs_l<" in k]
sys.path.insert(0, root)
import bench  # noqa: E402
main = {k: v for k, v in out.items() if any(t in k for t in ("kt_project", "kt2_project", "kt_sigma", "kt_rhs", "kt2_sigma", "kt2_rhs", "kt3_rhs", "kh_project", "kh_rhs"))}
# the hash recorded on the GPU box when the passes ran (tools/profile_round.sh); never the hash of whatever the sources are now
sha_file = os.path.join(src, "kernel_src_sha.txt")
measured_sha = open(sha_file).read().strip() if os.path.exists(sha_file) else None
extra = {"kernel_src_sha": measured_sha,
         "whole_rhs_hbm_bytes": sum(v["hbm_bytes_per_launch"] for v in main.values()),
         "whole_rhs_fp64_flops": sum(v.get("fp64_flops", 0.0) for v in main.values()) or None,
         "k_rhs_fp64_flops_per_launch": (rhs[0].get("fp64_flops") if rhs else None),
         "k_rhs_insts_valu_per_launch": (rhs[0].get("insts_valu") if rhs else None),
         "whole_rhs_insts_valu": sum(v.get("insts_valu", 0.0) for v in main.values()) or None}
allj[key] = {"source": f"gpurun_out/prof_{tag} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; durations from the "
                       "--kernel-trace --stats pass)",
             "kernels": out, "k_rhs_hbm_bytes_per_launch": rhs[0]["hbm_bytes_per_launch"] if rhs else None,
             "k_rhs_rocprofv3_avg_us": avg_us.get(rhs_name[0]) if rhs_name else None}
allj[key].update(extra)
json.dump(allj, open(dst, "w"), indent=1)
for k, v in out.items():
    print(f"{k:45s} fetch(raw) {v['fetch_bytes_raw']/1e6:9.1f} MB  x2 {v['fetch_bytes_x2']/1e6:9.1f} MB  write {v['write_bytes']/1e6:9.1f} MB")
