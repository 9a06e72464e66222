"""Per-evaluation timeline of the sharded schedule from a rocprofv3 --kernel-trace of tools/strip_overhead.py (STRIP_QUICK=1):
   python3 tools/strip_timeline.py gpurun_out/strip_trace"""
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
ks = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0].replace("void ", ""), r.get("Queue_Id", "?"),
       int(r.get("Grid_Size") or r.get("Grid_Size_X") or 0)) for r in rows]
big = [k for k in ks if "esdg" in k[2] and k[1] - k[0] > 100000]
# classify big launches by grid: the stand-alone engine launches the full grid, the sharded engine's interior a smaller one
grids = collections.Counter((k[2], k[4]) for k in big)
print("large launches (kernel, grid): count, avg us")
for (n, g), c in sorted(grids.items()):
    d = [(k[1] - k[0]) / 1e3 for k in big if k[2] == n and k[4] == g]
    print(f"  {n[:44]:44s} grid {g:9d}  n={c:3d}  avg {sum(d)/len(d):7.1f}  min {min(d):7.1f}")
proj = [k for k in big if "kt_project" in k[2]]
gmax = max(k[4] for k in proj)
for label, sel in (("sharded", [k for k in proj if k[4] != gmax]), ("stand-alone", [k for k in proj if k[4] == gmax])):
    st = sorted(k[0] for k in sel)
    per = [(b - a) / 1e3 for a, b in zip(st[:-1], st[1:]) if (b - a) < 5e6]
    per = per[len(per) // 2:]
    print(f"{label}: period between project launches (second half of the run): avg {sum(per)/len(per):.1f} us, min {min(per):.1f}")
i0 = [i for i, k in enumerate(ks) if "kt_project" in k[2] and k[4] != gmax and k[1] - k[0] > 100000][-3]
t0 = ks[i0][0]
print("one sharded evaluation (us relative to the interior project launch):")
for k in ks[i0 - 2:i0 + 22]:
    print(f"  {(k[0]-t0)/1e3:9.1f} -> {(k[1]-t0)/1e3:9.1f}  ({(k[1]-k[0])/1e3:7.1f})  q{k[3]}  {k[2][:60]}  grid {k[4]}")
