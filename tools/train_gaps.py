"""Busy time against wall time of the trainer's launches from a rocprofv3 --kernel-trace csv:
   python tools/train_gaps.py <kernel_trace.csv> [first launches to skip]
per stream (queue): the sum of kernel durations, the span, and the idle time between consecutive kernels."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[skip:]
t0, t1 = int(rows[0]["Start_Timestamp"]), max(int(r["End_Timestamp"]) for r in rows)
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
# union of busy intervals (two streams overlap)
iv = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows)
u, cur_s, cur_e = 0, iv[0][0], iv[0][1]
for s, e in iv[1:]:
    if s > cur_e:
        u += cur_e - cur_s
        cur_s, cur_e = s, e
    else:
        cur_e = max(cur_e, e)
u += cur_e - cur_s
print("launches %d  span %.1f us  sum of durations %.1f us  union busy %.1f us (%.1f %% of the span)  idle %.1f us" %
      (len(rows), (t1 - t0) / 1e3, busy / 1e3, u / 1e3, 100.0 * u / (t1 - t0), (t1 - t0 - u) / 1e3))
per = collections.defaultdict(lambda: [0, 0])
for r in rows:
    n = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:50]
    per[n][0] += 1
    per[n][1] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
for n, (c, d) in sorted(per.items(), key=lambda kv: -kv[1][1])[:20]:
    print("  %-52s %5d x %7.2f us" % (n, c, d / c / 1e3))
