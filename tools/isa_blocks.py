#!/usr/bin/env python3
"""Per-basic-block instruction counts of one kernel in a hipcc -S listing (development aid).
    python tools/isa_blocks.py stream.s <kernel-symbol-substring> [min_valu]"""
import collections
import re
import sys

path, sym = sys.argv[1], sys.argv[2]
min_valu = int(sys.argv[3]) if len(sys.argv) > 3 else 0
lines = open(path).read().split("\n")
start = next(i for i, l in enumerate(lines) if re.match(r"^_Z\w*%s\w*:" % re.escape(sym), l))
end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith(".end_amdhsa_kernel") or lines[i].startswith(".Lfunc_end"))
blocks = collections.OrderedDict()
cur = "entry"
blocks[cur] = []
for l in lines[start + 1:end]:
    s = l.strip()
    m = re.match(r"^(\.LBB\d+_\d+):", s)
    if m:
        cur = m.group(1)
        blocks[cur] = []
        continue
    if not s or s[0] in ";." :
        continue
    blocks[cur].append(s.split(";")[0].strip())
tot = collections.Counter()
for b, ins in blocks.items():
    c = collections.Counter()
    for x in ins:
        op = x.split()[0]
        k = "valu" if op.startswith("v_") else "salu" if op.startswith("s_") else "lds" if op.startswith("ds_") else "vmem"
        c[k] += 1
    tot.update(c)
    if c["valu"] >= min_valu:
        br = [x.split()[-1] for x in ins if x.startswith(("s_cbranch", "s_branch"))]
        print(f"{b:12s} valu {c['valu']:4d} salu {c['salu']:4d} lds {c['lds']:3d} vmem {c['vmem']:3d}  -> {' '.join(br)}")
print(dict(tot))
