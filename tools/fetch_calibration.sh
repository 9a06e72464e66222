#!/bin/bash
# FETCH_SIZE / WRITE_SIZE calibration on known byte counts (tools/ubench/fetch_calib.hip): plain run (rates), then one rocprofv3
# --pmc pass per counter; summary -> gpurun_out/fetch_calibration.txt (copy to profiles/r04_fetch_calibration.txt)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/fetch_calib; rm -rf $O; mkdir -p $O
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -o $O/fetch_calib tools/ubench/fetch_calib.hip || exit 1
$O/fetch_calib > $O/plain.txt 2>&1 || { cat $O/plain.txt; exit 1; }
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$c -- $O/fetch_calib > $O/$c.log 2>&1
  find $O/$c -name "*counter_collection.csv" -exec cp {} $O/$c.csv \;
done
python3 - <<PY > gpurun_out/fetch_calibration.txt
import csv, collections, re
O = "$O"
known = {}
for l in open(O + "/plain.txt"):
    m = re.match(r"(\S+)\s+read\s+([\d.]+) MB\s+write\s+([\d.]+) MB\s+([\d.]+) ms\s+([\d.]+) TB/s", l)
    if m: known[m.group(1)] = (float(m.group(2)) * 1e6, float(m.group(3)) * 1e6, float(m.group(4)), float(m.group(5)))
names = {"k_stream8": "stream8", "k_stream16": "stream16", "k_copy16": "copy16", "k_copy8": "copy8", "k_store24": "store24", "k_store32": "store32", "k_store8": "store8"}
def per_kernel(path, counter):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return acc
f = per_kernel(O + "/FETCH_SIZE.csv", "FETCH_SIZE"); w = per_kernel(O + "/WRITE_SIZE.csv", "WRITE_SIZE")
print("FETCH_SIZE / WRITE_SIZE calibration on MI355X (gfx950, ROCm 7.2), tools/ubench/fetch_calib.hip; counters in KiB x 1024")
print("%-14s %12s %12s %9s %9s | %14s %8s | %14s %8s" % ("kernel", "read MB", "write MB", "ms", "TB/s", "FETCH_SIZE MB", "/read", "WRITE_SIZE MB", "/write"))
def row(label, kname, sel):
    rb, wb, ms, tb = known[label]
    fv = [v for v in f.get(kname, [])]; wv = [v for v in w.get(kname, [])]
    fv = sel(fv); wv = sel(wv)
    fm = sum(fv) / len(fv) * 1024 if fv else 0.0; wm = sum(wv) / len(wv) * 1024 if wv else 0.0
    print("%-14s %12.1f %12.1f %9.4f %9.2f | %14.1f %8s | %14.1f %8s" % (label, rb / 1e6, wb / 1e6, ms, tb, fm / 1e6, ("%.3f" % (fm / rb)) if rb else "-", wm / 1e6, ("%.3f" % (wm / wb)) if wb else "-"))
for k, lab in names.items(): row(lab, k, lambda v: v)
# k_trace runs twice per pass set: first 23 launches neighbour records, next 23 own records
row("trace_nbr32", "k_trace", lambda v: v[:len(v) // 2]); row("trace_own32", "k_trace", lambda v: v[len(v) // 2:])
PY
cat gpurun_out/fetch_calibration.txt
