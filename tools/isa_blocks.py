#!/usr/bin/env python3
"""Per-basic-block instruction mix of a gfx950 .s file (hipcc -save-temps): quick check of what a loop
body really issues.  usage: isa_blocks.py file.s [min_instructions]"""
import re
import sys
from collections import Counter

lines = open(sys.argv[1]).read().split("\n")
mn = int(sys.argv[2]) if len(sys.argv) > 2 else 40
blocks, cur, name = [], [], "entry"
for l in lines:
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append((name, cur)); name, cur = m.group(1), []
    elif l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;"):
        cur.append(l.strip().split()[0])
blocks.append((name, cur))
for n, b in blocks:
    if len(b) < mn or "s_endpgm" in b:
        continue
    c = Counter(b)
    valu = sum(v for k, v in c.items() if k.startswith("v_") and not k.startswith("v_mfma"))
    print(f"{n}: {len(b)} instr, valu {valu}, mfma {sum(v for k, v in c.items() if k.startswith('v_mfma'))}, "
          f"ds {sum(v for k, v in c.items() if k.startswith('ds_'))}, barrier {c.get('s_barrier', 0)}")
    print("    " + ", ".join(f"{k}:{v}" for k, v in c.most_common(8)))
