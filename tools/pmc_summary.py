"""Summarise rocprofv3 --pmc CSV output: mean counter value per kernel over the dispatches found.
usage: python tools/pmc_summary.py <dir> [skip_first_n_dispatches_per_kernel]"""
import csv, glob, os, sys
from collections import defaultdict
d = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
acc = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
    seen = defaultdict(int)
    rows = list(csv.DictReader(open(f)))
    for r in rows:
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in sorted(acc.items()):
    print(k)
    for c, v in sorted(cs.items()):
        v = v[skip:] if len(v) > skip else v
        print(f"    {c:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
