"""Summarise rocprofv3 --pmc counter_collection.csv files per kernel (mean over dispatches)."""
import csv, sys, collections, re
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"^void ", "", n)
    m = re.match(r"_ZN12_GLOBAL__N_1\d+([a-z_0-9]+?)(ILi(\d+)(ELi(\d+))?E)?Ev", n)
    if m:
        return m.group(1) + ("<%s%s>" % (m.group(3), "," + m.group(5) if m.group(5) else "") if m.group(3) else "")
    return n.split("(")[0][:40]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
names = sorted({c for k in agg.values() for c in k})
print("kernel".ljust(28), " ".join(n[-14:].rjust(14) for n in names))
for k, cs in sorted(agg.items(), key=lambda kv: -sum(kv[1].get("SQ_BUSY_CYCLES", [0]))):
    print(k[:28].ljust(28), " ".join(("%.4g" % (sum(cs[n]) / len(cs[n])) if n in cs else "-").rjust(14) for n in names))
