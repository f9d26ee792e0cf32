#!/usr/bin/env python3
"""Instruction counts of a kernel between its s_barrier instructions (the phases of a barrier-synchronised step), from `make asm`
output.  usage: asm_segments.py <file.s> <kernel symbol substring> [first segment] [segments]"""
import collections, sys
f, sym = sys.argv[1], sys.argv[2]
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0
count = int(sys.argv[4]) if len(sys.argv) > 4 else 1000
lines = open(f).read().split("\n")
start = [i for i, l in enumerate(lines) if l.startswith("_Z") and sym in l.split(":")[0]][0]
end = [i for i in range(start, len(lines)) if "s_endpgm" in lines[i]][0]
cur, segs = collections.Counter(), []
for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith(".") or t.endswith(":"):
        continue
    op = t.split()[0]
    if op == "s_barrier":
        segs.append(cur); cur = collections.Counter(); continue
    cur[op] += 1
segs.append(cur)
print("%s: %d segments" % (lines[start].split(":")[0], len(segs)))
for i, c in enumerate(segs[first:first + count], first):
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    mfma = sum(v for k, v in c.items() if k.startswith("v_mfma"))
    ds = sum(v for k, v in c.items() if k.startswith("ds_"))
    vmem = sum(v for k, v in c.items() if k.startswith(("global_", "buffer_", "scratch_")))
    salu = sum(v for k, v in c.items() if k.startswith("s_") and k != "s_nop")
    print("segment %2d: VALU %3d  MFMA %2d  LDS %2d  VMEM %2d  SALU %3d  s_nop %2d   top: %s" % (
        i, valu, mfma, ds, vmem, salu, c["s_nop"], ", ".join("%s x%d" % kv for kv in c.most_common(6))))
