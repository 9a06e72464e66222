"""Durations of the large launches, sharded (interior) vs stand-alone, from a rocprofv3 --kernel-trace of
tools/strip_overhead.py (STRIP_QUICK=1):  python3 tools/strip_modes.py gpurun_out/strip_traceN"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Queue_Id"], int(r["Grid_Size_X"])) for r in rows]
gmax = max(k[4] for k in ks if "kt_project" in k[2])
mode, acc, starts = None, collections.defaultdict(list), collections.defaultdict(list)
for k in ks:
    if "kt_project" in k[2] and k[4] > 1e7:
        mode = "single" if k[4] == gmax else "sharded"
        starts[mode].append(k[0])
    if k[4] > 1e5 and "esdg" in k[2] and mode:
        acc[(mode, k[2][:32], k[4])].append((k[1] - k[0]) / 1e3)
for k, v in sorted(acc.items()):
    v2 = v[len(v) // 2:]
    print(f"{k[0]:8s} {k[1]:32s} grid {k[2]:9d} n={len(v):3d}  avg {sum(v2)/len(v2):7.1f}  min {min(v2):7.1f}  max {max(v2):7.1f}")
for m, st in starts.items():
    per = [(b - a) / 1e3 for a, b in zip(st[:-1], st[1:]) if b - a < 5e6]
    per = per[len(per) // 2:]
    print(f"{m}: period avg {sum(per)/len(per):.1f} us  min {min(per):.1f}")
