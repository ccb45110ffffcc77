"""Reads a rocprofv3 kernel_trace.csv and reports per-kernel time, union busy time and overlap."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].split("(")[0]
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if not n.startswith("k_") and "k_search" not in n:
        continue
    per[n][0] += 1; per[n][1] += e - s
    ev.append((s, e, n, r.get("Queue_Id"), r.get("Stream_Id")))
ev.sort()
# restrict to the iteration kernels of the last ~60% of the trace
t0, t1 = ev[0][0], max(e for s, e, *_ in ev)
busy = 0; cur_s, cur_e = None, None
for s, e, *_ in ev:
    if cur_e is None or s > cur_e:
        if cur_e is not None: busy += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
busy += cur_e - cur_s
tot = sum(v[1] for v in per.values())
print("span %.2f ms, union busy %.2f ms, sum of kernel durations %.2f ms (overlap factor %.2f)" % ((t1 - t0) / 1e6, busy / 1e6, tot / 1e6, tot / busy))
for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1]):
    print("  %-28s calls %6d  avg %8.1f us  total %8.2f ms" % (n, c, d / c / 1e3, d / 1e6))
print("queues:", collections.Counter(q for *_, q, st in ev), "streams:", collections.Counter(st for *_, q, st in ev))
