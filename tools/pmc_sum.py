"""Summarise rocprofv3 --pmc CSVs: mean counter value per kernel (last launches only).
python tools/pmc_sum.py <dir> [filter] [--json out.json]   (the JSON adds mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024)
when both counters are there)"""
import csv, glob, json, os, re, sys
from collections import defaultdict

jout = None
if "--json" in sys.argv:
    i = sys.argv.index("--json")
    jout = sys.argv[i + 1]
    del sys.argv[i:i + 2]
root = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
doc = {}
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
    ent = {"median_us_profiled": round(d[len(d) // 2], 1), "counters": {}}
    for c, v in sorted(vals[k].items()):
        v = v[len(v) // 2:]                      # skip warm-up launches
        print(f"    {c:32s} {sum(v) / len(v):16.0f}")
        ent["counters"][c] = round(sum(v) / len(v))
    cn = ent["counters"]
    if cn.get("GRBM_GUI_ACTIVE") and "SQ_VALU_MFMA_BUSY_CYCLES" in cn:
        ent["mfma_util"] = round(cn["SQ_VALU_MFMA_BUSY_CYCLES"] / (cn["GRBM_GUI_ACTIVE"] / 8 * 1024), 4)
    doc[k] = ent
if jout:
    with open(jout, "w") as f:
        json.dump({"note": "rocprofv3 --pmc, one counter group per pass, kernel trace only; whole-GPU sums per launch (mean of the later half of "
                           "the launches); mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs)", "kernels": doc}, f, indent=1)
