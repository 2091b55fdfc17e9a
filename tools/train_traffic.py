"""HBM traffic of the trainer's kernels from two rocprofv3 PMC passes + a kernel-trace pass (all with NF_TRAIN_SERIAL=1):

    python tools/train_traffic.py <fetch counter_collection.csv> <write counter_collection.csv> <kernel_stats.csv>

Per kernel name: launches, average duration, FETCH_SIZE / WRITE_SIZE per launch (KB -> MB; gfx950 correction of
tools/make_traffic.py: fetches x2 for wide coalesced reads, writes as reported) and the resulting HBM rate."""
import csv, sys
from collections import defaultdict


def short(name):
    return name.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]


def per_kernel(path, counter):
    tot, cnt = defaultdict(float), defaultdict(int)
    with open(path) as f:
        for row in csv.DictReader(f):
            if row["Counter_Name"] == counter:
                k = short(row["Kernel_Name"])
                tot[k] += float(row["Counter_Value"])
                cnt[k] += 1
    return {k: tot[k] / cnt[k] for k in tot}


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
rows = list(csv.DictReader(open(sys.argv[3])))
print("%-46s %6s %9s %10s %10s %9s" % ("kernel", "calls", "avg us", "fetch MB", "write MB", "TB/s"))
for r in rows[:16]:
    k = short(r["Name"])
    us = float(r["AverageNs"]) / 1e3
    f_mb = 2.0 * fetch.get(k, 0.0) * 1024 / 1e6
    w_mb = write.get(k, 0.0) * 1024 / 1e6
    print("%-46s %6s %9.1f %10.1f %10.1f %9.2f" % (k, r["Calls"], us, f_mb, w_mb, (f_mb + w_mb) / us if us else 0.0))
