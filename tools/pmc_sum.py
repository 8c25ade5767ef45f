"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel (last launches only).  python tools/pmc_sum.py <dir> [filter]"""
import csv, glob, os, re, sys
from collections import defaultdict

root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
vals = defaultdict(lambda: defaultdict(list))
dur = defaultdict(list)
for f in sorted(glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if n.startswith("void at::") or "rocclr" in n:
            continue
        m = re.search(r"(\w+_kernel)(<[^>]*>)?", n)
        key = (m.group(1) + (m.group(2) or "")) if m else n[:60]
        vals[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
        if "Start_Timestamp" in r and r["Start_Timestamp"]:
            dur[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k in sorted(vals):
    if flt and flt not in k:
        continue
    d = sorted(dur[k])
    print(f"{k}   (median {d[len(d)//2]:.1f} us under the profiler)")
    for c, v in sorted(vals[k].items()):
        v = v[len(v) // 2:]                      # skip warm-up launches
        print(f"    {c:32s} {sum(v) / len(v):16.0f}")
