#!/usr/bin/env python3
"""Basic blocks of one kernel in a .s file with their instruction mix: python tools/asm_blocks.py file.s [min_instructions]
Prints, per block in file order: first line, label, loop depth, vector / scalar / LDS / memory / branch instruction counts."""
import re
import sys

path = sys.argv[1]
least = int(sys.argv[2]) if len(sys.argv) > 2 else 1
lines = open(path).read().split("\n")
blocks = []
cur = None
depth = 0
for i, ln in enumerate(lines, 1):
    t = ln.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", t) or re.match(r"^; %bb\.(\d+):", t)
    if m:
        d = re.search(r"Depth[= ](\d+)", ln)
        cur = {"line": i, "label": m.group(1), "depth": int(d.group(1)) if d else -1, "v": 0, "s": 0, "lds": 0, "mem": 0, "br": 0, "rl": 0}
        blocks.append(cur)
        continue
    if cur is None or not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("v_readlane") or op.startswith("v_writelane"):
        cur["rl"] += 1
        cur["v"] += 1
    elif op.startswith("v_"):
        cur["v"] += 1
    elif op.startswith("ds_"):
        cur["lds"] += 1
    elif op.startswith("global_") or op.startswith("flat_") or op.startswith("buffer_") or op.startswith("s_load") or op.startswith("scratch_"):
        cur["mem"] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur["br"] += 1
        cur["s"] += 1
    elif op.startswith("s_waitcnt") or op.startswith("s_nop"):
        pass
    elif op.startswith("s_"):
        cur["s"] += 1
tot = {"v": 0, "s": 0}
for b in blocks:
    n = b["v"] + b["s"] + b["lds"] + b["mem"]
    tot["v"] += b["v"]
    tot["s"] += b["s"]
    if n >= least:
        print("%6d %-12s d%-2d  V %4d (lane moves %3d)  S %4d  LDS %3d  MEM %3d  BR %2d" % (b["line"], b["label"], b["depth"], b["v"], b["rl"], b["s"], b["lds"], b["mem"], b["br"]))
print("total vector %d scalar %d" % (tot["v"], tot["s"]))
