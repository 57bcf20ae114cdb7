#!/usr/bin/env python3
"""Scratch: what the device did during a traced CLI run (tools/trace_cli.sh): per kernel name count / total / mean, copies
by direction, and the busy time of the union of all kernel intervals against the span they cover.
    python tools/trace_summary.py <dir with cli_kernel_trace.csv / cli_memory_copy_trace.csv>"""
import csv
import glob
import os
import sys
from collections import defaultdict

d = sys.argv[1]


def rows(pat):
    out = []
    for f in glob.glob(os.path.join(d, "**", pat), recursive=True):
        out += list(csv.DictReader(open(f)))
    return out


k = rows("*kernel_trace.csv")
m = rows("*memory_copy_trace.csv")
agg = defaultdict(lambda: [0, 0])
iv = []
for r in k:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    n = r["Kernel_Name"].split("(")[0][:40]
    agg[n][0] += 1
    agg[n][1] += b - a
    iv.append((a, b, n))
for n, (c, t) in sorted(agg.items(), key=lambda x: -x[1][1]):
    print(f"{n:42s} {c:6d} calls {t / 1e6:10.2f} ms total {t / c / 1e3:10.1f} us mean")
if iv:
    iv.sort()
    t0, t1 = iv[0][0], max(b for _, b, _ in iv)
    busy, cur_a, cur_b = 0, iv[0][0], iv[0][1]
    for a, b, _ in iv[1:]:
        if a > cur_b:
            busy += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    busy += cur_b - cur_a
    print(f"kernels span {(t1 - t0) / 1e6:.1f} ms, some kernel running {busy / 1e6:.1f} ms ({100 * busy / (t1 - t0):.0f} %)")
    for name in ("k_tokens", "k_resolve"):
        sel = [(a, b) for a, b, n in iv if n.startswith(name)]
        if sel:
            sel.sort()
            gaps = [sel[i + 1][0] - sel[i][0] for i in range(len(sel) - 1)]
            gaps.sort()
            print(f"{name}: start-to-start median {gaps[len(gaps) // 2] / 1e6:.2f} ms, span {(sel[-1][1] - sel[0][0]) / 1e6:.1f} ms")
# how the two decoder passes share the device: time with a pass-1 kernel running, a pass-2 kernel running, both, neither
def union(sel):
    sel = sorted(sel)
    out = []
    for a, b in sel:
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(u):
    return sum(b - a for a, b in u)


def inter(u, v):
    i = j = 0
    tot = 0
    while i < len(u) and j < len(v):
        a, b = max(u[i][0], v[j][0]), min(u[i][1], v[j][1])
        if b > a:
            tot += b - a
        if u[i][1] < v[j][1]:
            i += 1
        else:
            j += 1
    return tot


tk = union([(a, b) for a, b, n in iv if n.startswith("k_tokens")])
rs = union([(a, b) for a, b, n in iv if n.startswith("k_resolve")])
if tk and rs:
    lo, hi = min(tk[0][0], rs[0][0]), max(tk[-1][1], rs[-1][1])
    both = inter(tk, rs)
    print(f"decoder window {(hi - lo) / 1e6:.1f} ms: pass 1 running {length(tk) / 1e6:.1f} ms, pass 2 running {length(rs) / 1e6:.1f} ms, both {both / 1e6:.1f} ms, "
          f"neither {((hi - lo) - length(tk) - length(rs) + both) / 1e6:.1f} ms")
    # concurrency of pass-1 kernels among themselves: sum of durations / union
    d1 = sum(b - a for a, b, n in iv if n.startswith("k_tokens"))
    d2 = sum(b - a for a, b, n in iv if n.startswith("k_resolve"))
    print(f"mean pass-1 kernels in flight while any runs: {d1 / max(length(tk), 1):.2f}; pass-2: {d2 / max(length(rs), 1):.2f}")
cp = defaultdict(lambda: [0, 0, 0])
for r in m:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    key = r.get("Direction", "?")
    cp[key][0] += 1
    cp[key][1] += b - a
    cp[key][2] += int(r.get("Size", 0) or 0) if "Size" in r else 0
for kx, (c, t, sz) in cp.items():
    print(f"copy {kx:22s} {c:6d} copies {t / 1e6:10.2f} ms total {sz / 1e9:8.2f} GB")
# the big host-to-device copies (the compressed chunks): how long the link was busy with them and at what rate each ran
big = []
for r in m:
    a, b = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    sz = int(r.get("Size", 0) or 0) if "Size" in r else 0
    if "HOST_TO_DEVICE" in r.get("Direction", "") and (sz >= (32 << 20) or (not sz and b - a > 2_000_000)):
        big.append((a, b, sz))
if big:
    u = union([(a, b) for a, b, _ in big])
    tot = sum(s for _, _, s in big)
    rates = sorted((s / (b - a)) for a, b, s in big if s)
    print(f"big H2D copies: {len(big)}, {tot / 1e9:.2f} GB, link busy with them {length(u) / 1e6:.1f} ms of a span of {(u[-1][1] - u[0][0]) / 1e6:.1f} ms"
          + (f", per-copy rate median {rates[len(rates) // 2]:.1f} GB/s (min {rates[0]:.1f}, max {rates[-1]:.1f})" if rates else ""))
    if tk:
        print(f"pass 1 running {length(tk) / 1e6:.1f} ms; copies and pass 1 together {inter(u, tk) / 1e6:.1f} ms")
