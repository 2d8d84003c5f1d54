"""Per-kernel statistics from a rocprofv3 `--kernel-trace --stats` run (rocpd SQLite output, ROCm 7.2).
usage: python tools/rocpd_stats.py <dir-or-db> [min_calls]  ->  CSV on stdout: name, calls, total_us, avg_us, min_us, max_us, pct,
vgpr, agpr, scratch_B, lds_B, grid, workgroup"""
import glob, os, sqlite3, sys
src = sys.argv[1]
dbs = [src] if src.endswith(".db") else sorted(glob.glob(os.path.join(src, "**", "*results.db"), recursive=True))
min_calls = int(sys.argv[2]) if len(sys.argv) > 2 else 1
print("name,calls,total_us,avg_us,min_us,max_us,pct,vgpr,agpr,scratch_B,lds_B,grid,workgroup")
for db in dbs:
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(vgpr_count), "
                          "max(accum_vgpr_count), max(scratch_size), max(lds_size), max(grid_x)||'x'||max(grid_y), max(workgroup_x) "
                          "from kernels group by name order by sum(duration) desc"))
    tot = sum(r[2] for r in rows) or 1
    for r in rows:
        if r[1] < min_calls:
            continue
        name = r[0].split("(")[0].replace("void ", "").replace(",", ";")
        print(f"{name},{r[1]},{r[2]/1e3:.1f},{r[3]/1e3:.2f},{r[4]/1e3:.2f},{r[5]/1e3:.2f},{100*r[2]/tot:.2f},{r[6]},{r[7]},{r[8]},{r[9]},{r[10]},{r[11]}")
