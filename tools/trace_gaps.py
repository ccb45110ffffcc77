"""Per-kernel durations and the idle gaps between consecutive kernels from a rocprofv3
--kernel-trace CSV (one stream): python tools/trace_gaps.py <kernel_trace.csv> [skip_first_n]"""
import csv
import sys
from collections import defaultdict

rows = []
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0][:40]))
rows.sort()
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows = rows[skip:]
dur = defaultdict(list)
gap_after = defaultdict(list)
for i, (s, e, n) in enumerate(rows):
    dur[n].append(e - s)
    if i + 1 < len(rows):
        gap_after[n].append(rows[i + 1][0] - e)
print("%-42s %8s %10s %12s" % ("kernel", "count", "avg_us", "gap_after_us"))
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    d = dur[n]
    g = gap_after[n]
    gs = sorted(g)
    print("%-42s %8d %10.2f %12.2f (median %.2f)" % (n, len(d), sum(d) / len(d) / 1e3,
                                                    (sum(g) / max(len(g), 1)) / 1e3, gs[len(gs) // 2] / 1e3 if gs else 0))
span = rows[-1][1] - rows[0][0]
print("span %.1f us, kernels busy %.1f us (%.1f %%)" % (span / 1e3, sum(sum(v) for v in dur.values()) / 1e3,
                                                        100.0 * sum(sum(v) for v in dur.values()) / span))
