"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean over dispatches).

    python tools/pmc_summary.py <pass dir or csv> [...]
"""
import collections, csv, glob, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.pmc_traffic import short

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for arg in sys.argv[1:]:
    paths = [arg] if arg.endswith(".csv") else glob.glob(os.path.join(arg, "**", "*counter_collection.csv"), recursive=True)
    for path in paths:
        for r in csv.DictReader(open(path)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0])) / max(len(kv[1].get("SQ_BUSY_CYCLES", [0])), 1)):
    print(k)
    for n in names:
        if n in cs:
            print("    %-28s %.6g" % (n, sum(cs[n]) / len(cs[n])))
