#!/usr/bin/env python3
"""Per-launch durations of the kernels of a few mid-run planner rounds, from a rocprofv3 kernel-trace CSV.
A round starts at round_begin_kernel.  Prints, for rounds at 25 / 50 / 75 / 95 % of the trace, every launch with its
duration and the gap to the previous kernel's end, then the mean duration of the step-wise steer launches by step index."""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
starts = [i for i, r in enumerate(rows) if "round_begin_kernel" in r[2]]
print(f"{len(rows)} kernel launches, {len(starts)} rounds")


def short(n):
    n = n.replace("rkh::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:60]


for frac in (0.25, 0.5, 0.75, 0.95):
    if not starts:
        break
    k = int(frac * (len(starts) - 1))
    a = starts[k]
    b = starts[k + 1] if k + 1 < len(starts) else len(rows)
    print(f"--- round {k} ({frac:.0%} of the run): {b - a} launches, {(rows[b - 1][1] - rows[a][0]) / 1e3:.1f} us")
    prev_end = rows[a][0]
    for s, e, n in rows[a:b]:
        print(f"  {short(n):60s} {(e - s) / 1e3:9.1f} us   gap {(s - prev_end) / 1e3:7.1f}")
        prev_end = e
# step-wise steer launches: mean duration by position within the round
by_pos = defaultdict(list)
for k in range(len(starts)):
    a = starts[k]
    b = starts[k + 1] if k + 1 < len(starts) else len(rows)
    pos = 0
    for s, e, n in rows[a:b]:
        if "propagate_pair_step_kernel" in n:
            by_pos[pos].append((e - s) / 1e3)
            pos += 1
if by_pos:
    print("step-wise steer launches, mean us by step (second half of the run):")
    tot = 0.0
    for pos in sorted(by_pos):
        v = by_pos[pos][len(by_pos[pos]) // 2:]
        m = sum(v) / max(1, len(v))
        tot += m
        print(f"  step {pos:2d}: {m:8.1f} us  ({len(v)} launches)")
    print(f"  sum {tot:.1f} us")
