"""Per-launch averages of rocprofv3 PMC passes for one kernel.

    python tools/pmc_report.py <dir-with-*_counter_collection.csv (searched recursively)> <kernel-substring> [min-grid]

Every pass is its own `rocprofv3 --pmc <counters> --kernel-trace --output-format csv -d <dir>/<pass> -- <cmd>` run
(never combined with other trace domains).  Prints one `COUNTER,value` line per counter (mean over the matching
dispatches) plus the derived figures DESIGN.md / profiles/ quote: MFMA-busy share, VALU per MFMA, LDS activity."""
import csv
import glob
import os
import sys


def collect(root, kernel_sub, min_grid=0):
    acc = {}
    dur = []
    for path in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        with open(path) as f:
            for row in csv.DictReader(f):
                if kernel_sub not in row["Kernel_Name"] or int(row["Grid_Size"]) < min_grid:
                    continue
                acc.setdefault(row["Counter_Name"], []).append(float(row["Counter_Value"]))
                if row.get("End_Timestamp") and row.get("Start_Timestamp"):
                    dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) * 1e-3)
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}, dur


def main():
    root, kernel_sub = sys.argv[1], sys.argv[2]
    min_grid = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    mean, cnt, dur = collect(root, kernel_sub, min_grid)
    if not mean:
        raise SystemExit("no rows for %r under %s" % (kernel_sub, root))
    print("# kernel substring %r, min grid %d; dispatches per counter: %s" % (kernel_sub, min_grid, sorted(set(cnt.values()))))
    for k in sorted(mean):
        print("%s,%.6g" % (k, mean[k]))
    g = mean.get
    if dur:
        dur.sort()
        print("# kernel time under the profiler: median %.1f us" % dur[len(dur) // 2])
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("GRBM_GUI_ACTIVE"):
        # GRBM_GUI_ACTIVE counts per XCD (8) -> cycles = /8; 1024 SIMDs; BUSY_CYCLES is summed over SIMDs in units of 4 cycles? report raw ratio too
        cyc = g("GRBM_GUI_ACTIVE") / 8.0
        print("# shader cycles per launch (GRBM_GUI_ACTIVE/8): %.4g" % cyc)
        print("# MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x cycles): %.1f %%" % (100.0 * g("SQ_VALU_MFMA_BUSY_CYCLES") / (1024.0 * cyc)))
    if g("SQ_INSTS_VALU") and g("SQ_INSTS_MFMA"):
        print("# non-MFMA VALU instructions per MFMA instruction: %.3f" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")))
    if g("SQ_LDS_IDX_ACTIVE") and g("GRBM_GUI_ACTIVE"):
        print("# LDS active = SQ_LDS_IDX_ACTIVE / (256 CUs x cycles): %.1f %%" % (100.0 * g("SQ_LDS_IDX_ACTIVE") / (256.0 * g("GRBM_GUI_ACTIVE") / 8.0)))
    if g("SQ_LDS_BANK_CONFLICT") and g("SQ_LDS_IDX_ACTIVE"):
        print("# LDS bank-conflict cycles / LDS active: %.1f %%" % (100.0 * g("SQ_LDS_BANK_CONFLICT") / g("SQ_LDS_IDX_ACTIVE")))
    if g("SQ_WAVE_CYCLES"):
        for name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            if g(name):
                print("# %s / SQ_WAVE_CYCLES: %.1f %%" % (name, 100.0 * g(name) / g("SQ_WAVE_CYCLES")))


if __name__ == "__main__":
    main()
