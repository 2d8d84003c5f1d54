"""Mean counter value per kernel from rocprofv3 --pmc runs (rocpd SQLite output).
usage: python tools/rocpd_pmc.py <dir> [skip_first_n_dispatches_per_kernel]"""
import glob, os, sqlite3, sys
from collections import defaultdict
src = sys.argv[1]; skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for db in sorted(glob.glob(os.path.join(src, "**", "*results.db"), recursive=True)):
    c = sqlite3.connect(db)
    cols = [d[1] for d in c.execute("pragma table_info(counters_collection)")]
    name_col = "kernel_name" if "kernel_name" in cols else "name"
    acc = defaultdict(lambda: defaultdict(list))
    q = f"select {name_col}, counter_name, value, dispatch_id from counters_collection order by dispatch_id"
    per = defaultdict(lambda: defaultdict(float))
    for k, cn, v, disp in c.execute(q):
        per[(k.split('(')[0].replace('void ', '')[:64], disp)][cn] += float(v)   # sum over the instances (XCDs / SEs) of a dispatch
    for (k, disp), cs in sorted(per.items(), key=lambda kv: kv[0][1]):
        for cn, v in cs.items():
            acc[k][cn].append(v)
    for k, cs in sorted(acc.items()):
        print(k)
        for cn, v in sorted(cs.items()):
            v = v[skip:] if len(v) > skip else v
            print(f"    {cn:34s} n={len(v):4d} mean={sum(v)/len(v):.6g}")
