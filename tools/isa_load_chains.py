#!/usr/bin/env python3
"""Per kernel of a disassembly (llvm-objdump -d of a gfx950 code object): vector-memory loads, the waits that drain them, and the
longest run of 'load(s) then a wait that leaves none in flight' — a compiler-serialised chain of memory round trips (the
tridiagonalisation's load prologue was 105 of them: tools/r4_trace_c4.sh, icp_tridiag.hpp).
usage: isa_load_chains.py <file.s> [min_chain]"""
import re, sys
path = sys.argv[1]
min_chain = int(sys.argv[2]) if len(sys.argv) > 2 else 6
name = None
stats = {}
cur = None
for line in open(path, errors="replace"):
    m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
    if m:
        name = m.group(1)
        cur = stats.setdefault(name, {"loads": 0, "waits0": 0, "chain": 0, "best": 0, "pending": 0, "insns": 0})
        continue
    if cur is None: continue
    t = line.strip().split("//")[0].strip()
    if not t: continue
    cur["insns"] += 1
    op = t.split()[0]
    if op.startswith(("global_load", "buffer_load", "flat_load", "scratch_load")):
        cur["loads"] += 1
        cur["pending"] += 1
    elif op == "s_waitcnt" and "vmcnt(0)" in t:
        if cur["pending"] > 0:
            cur["waits0"] += 1
            cur["chain"] += 1
            cur["best"] = max(cur["best"], cur["chain"])
        cur["pending"] = 0
    elif op.startswith(("s_cbranch", "s_branch", "s_barrier", "s_endpgm")):
        cur["chain"] = 0
rows = [(v["best"], k, v) for k, v in stats.items() if v["best"] >= min_chain]
rows.sort(reverse=True)
for best, k, v in rows:
    print("%4d-long chain | %5d loads, %4d full drains, %6d instructions | %s" % (best, v["loads"], v["waits0"], v["insns"], k[:150]))
