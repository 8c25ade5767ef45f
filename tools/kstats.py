"""Print (name, calls, average us) of the kernels in a rocprofv3 --stats kernel_stats.csv: python tools/kstats.py <dir> [filter]"""
import csv, glob, sys
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for f in glob.glob(sys.argv[1] + "/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if flt in r["Name"]:
            print(f"{r['Name'][:70]:70s} calls {r['Calls']:>5s}  avg {float(r['AverageNs']) / 1e3:8.1f} us  min {float(r['MinNs']) / 1e3:8.1f}")
