"""Which source lines a kernel's scratch traffic belongs to.

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -S -g1 -o /tmp/k.s csrc/<file>.hip     (device assembly with .loc lines)
    python tools/spill_lines.py /tmp/k.s <mangled kernel name prefix>

Counts scratch_store / scratch_load (and MFMA / VALU / LDS / waitcnt totals) per `.loc file line` of the kernel: the loop
header line collects what lives across the loop, a phase's line what it alone overflows (profiles/r03_notes.md)."""
import collections
import re
import sys

path, name = sys.argv[1], sys.argv[2]
lines = open(path).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith(name + ":")][0]
cur = None
st, ld, tot = collections.Counter(), collections.Counter(), collections.Counter()
for l in lines[start:]:
    if ".end_amdhsa_kernel" in l or l.startswith(".Lfunc_end"):
        break
    m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
    if m:
        cur = (int(m.group(1)), int(m.group(2)))
        continue
    t = l.strip()
    if not t or t.startswith(".") or t.startswith(";"):
        continue
    op = t.split()[0]
    if op.startswith("scratch_store"):
        st[cur] += 1
    elif op.startswith("scratch_load"):
        ld[cur] += 1
    kind = ("mfma" if "mfma" in op else "lds" if op.startswith("ds_") else "vmem" if op.startswith(("global_", "buffer_", "scratch_"))
            else "wait" if op.startswith("s_waitcnt") else "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "other")
    tot[kind] += 1
print("totals", dict(tot), "scratch st", sum(st.values()), "ld", sum(ld.values()))
for k in sorted(set(st) | set(ld)):
    print(k, "st", st[k], "ld", ld[k])

# per-line instruction mix (lines with the most instructions first): python tools/spill_lines.py k.s <kernel> mix
if len(sys.argv) > 3 and sys.argv[3] == "mix":
    per = collections.defaultdict(collections.Counter)
    cur = None
    for l in lines[start:]:
        if ".end_amdhsa_kernel" in l or l.startswith(".Lfunc_end"):
            break
        m = re.match(r"\s*\.loc\s+(\d+)\s+(\d+)", l)
        if m:
            cur = (int(m.group(1)), int(m.group(2)))
            continue
        t = l.strip()
        if not t or t.startswith(".") or t.startswith(";"):
            continue
        op = t.split()[0]
        kind = ("accvgpr" if "accvgpr" in op else "mfma" if "mfma" in op else "lds" if op.startswith("ds_") else
                "vmem" if op.startswith(("global_", "buffer_", "scratch_")) else "wait" if op.startswith("s_waitcnt") else
                "nop" if op.startswith("s_nop") else "valu" if op.startswith("v_") else "salu")
        per[cur][kind] += 1
    for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1].values()))[:40]:
        print(k, dict(v))
