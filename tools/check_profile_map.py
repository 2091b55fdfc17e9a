"""Every `kernel_ms` of a bench line against the rocprofv3 kernel-stats csv that backs it (profiles/README.md's table):
    python tools/check_profile_map.py profiles/r05_bench.json profiles r05
prints one row per roofline block: bench kernel_ms (HIP events), the csv's AverageNs of the dominant kernel, the ratio."""
import csv, json, os, re, sys

bench, pdir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
d = json.loads(open(bench).read().strip().splitlines()[-1])


def dominant(path):
    rows = list(csv.DictReader(open(path)))
    rows = [r for r in rows if "nf_" in r["Name"] and "synth" not in r["Name"]]
    r = max(rows, key=lambda r: float(r["TotalDurationNs"]))
    m = re.search(r"(nf_\w+(?:<[^>]*>)?)", r["Name"])
    return (m.group(1) if m else r["Name"][:60]).replace(" ", ""), float(r["AverageNs"]) * 1e-6, int(r["Calls"])


m = [("headline (configs[1])", d["roofline"]["kernel_ms"], "kernel_stats.csv"),
     ("sampling (configs[2])", d["sampling"]["roofline"]["kernel_ms"], "kernel_stats_sampling.csv"),
     ("fp16_cnn_64x64 (configs[4])", d["fp16_cnn_64x64"]["roofline"]["kernel_ms"], "kernel_stats_fp16_cnn_64x64.csv")]
for key in ("w16", "w32", "w32_fp16", "w128", "w512", "w512_fp16"):
    if key in d.get("wide_cnn", {}):
        m.append(("wide_cnn." + key, d["wide_cnn"][key]["kernel_ms"], "kernel_stats_%s.csv" % key))
worst = 0.0
print("| bench section | kernel_ms (HIP events) | csv | kernel | csv average ms (launches) | csv / bench |")
print("|---|---|---|---|---|---|")
for name, ms, f in m:
    path = os.path.join(pdir, ("%s_%s" % (tag, f)) if tag else f)
    if not os.path.exists(path):
        print("| %s | %.4f | %s | MISSING | | |" % (name, ms, os.path.basename(path)))
        continue
    kn, avg, calls = dominant(path)
    worst = max(worst, abs(avg / ms - 1.0))
    print("| %s | %.4f | `%s` | `%s` | %.4f (%d) | %.3f |" % (name, ms, os.path.basename(path), kn, avg, calls, avg / ms))
print("worst deviation: %.1f %%" % (100 * worst))
