#!/usr/bin/env python3
"""What runs between the last PGD iteration of an attack and the first one of the next (the parameter update: forward, loss, backward with
weight gradients, SGD, filter re-arrangement), from the second section of a trace_sequence.py output: launches and time per kernel name.
Usage: update_breakdown.py <trace_sequence.txt>"""
import collections
import re
import sys

lines = open(sys.argv[1]).read().split("\n")
i = [k for k, l in enumerate(lines) if l.startswith("last iteration")][0]
names = []
for l in lines[i + 1:]:
    m = re.match(r"\s+([\d.]+) us\s+gap\s+([\d.]+)\s+(.*)", l)
    if m:
        names.append((float(m.group(1)), float(m.group(2)), m.group(3)))
a = [k for k, (d, g, n) in enumerate(names) if n.startswith("chain_bwd")][0] + 1
upd = names[a:]
print("after the attack: %d launches, busy %.1f us, gaps %.1f us" % (len(upd), sum(d for d, g, n in upd), sum(g for d, g, n in upd)))
agg = collections.defaultdict(lambda: [0, 0.0])
for d, g, n in upd:
    k = re.sub(r"<.*", "", n)[:70]
    agg[k][0] += 1
    agg[k][1] += d
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:40]:
    print("%4d %8.1f us  %s" % (c, t, k))
