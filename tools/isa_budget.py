"""Per-class instruction budget of a kernel's loop from its gfx950 assembly (no GPU needed).

    hipcc -O3 -std=c++17 --offload-arch=gfx950 -munsafe-fp-atomics --cuda-device-only -S csrc/nf_kernels.hip -o /tmp/k.s
    python tools/isa_budget.py /tmp/k.s 'nf_flow_kernel<4, 1024, 4, false, true, true, 2, false>' [first_label last_label] [--skip=L1,L2] [--fwd] [--list]

Without labels: the blocks between the first and the last s_barrier-carrying innermost loop of the kernel (the coupling loop of
the fused kernels).  `--skip` drops basic blocks (e.g. the other direction's affine block) from the count.  Issue weights are the
ones tools/probes/valu_rates.hip measured relative to a VGPR v_fmac: plain VALU 1, VOP3 with SGPR / packed
fp32 1.6, transcendentals 2.8, 2-pass MFMA (4x4x1 f32, 4x4x4 f16) 2.85, 16x16x32 f16 one slot."""
import re
import subprocess
import sys
from collections import OrderedDict

CLASSES = OrderedDict([
    ("mfma 16x16x32 f16", r"^v_mfma_f32_16x16x32"),
    ("mfma 32x32 f16", r"^v_mfma_f32_32x32x16"),
    ("mfma 4x4x4 f16", r"^v_mfma_f32_4x4x4"),
    ("mfma 4x4x1 f32", r"^v_mfma_f32_4x4x1"),
    ("mfma other", r"^v_mfma"),
    ("transcendental", r"^v_(exp|log|rcp|rsq|sqrt|sin|cos)_"),
    ("convert", r"^v_cvt_"),
    ("relu (max)", r"^v_(pk_)?max"),
    ("packed fp32", r"^v_pk_(fma|mul|add)_f32"),
    ("fp32 arithmetic", r"^v_(fma|fmac|mul|add|sub|mac|mad)_f32|^v_pk_(fma|mul|add)_f16"),
    ("move / select", r"^v_(mov|cndmask|accvgpr|perm|swap|readlane|readfirstlane|writelane|bfe|bfi|and|or|xor|lshl|lshr|ashr|not)"),
    ("integer / address", r"^v_(add|sub|mul|mad|lshl_add|add_lshl|lshl_or|and_or|or3|add3|cmp|min|max)_[ui]|^v_(add|sub)_co|^v_cmp"),
    ("lds read", r"^ds_read"),
    ("lds write", r"^ds_write"),
    ("global / buffer", r"^(global|buffer|flat|scratch)_"),
    ("barrier", r"^s_barrier"),
    ("waitcnt", r"^s_waitcnt"),
    ("nop", r"^s_nop"),
    ("scalar load", r"^s_(load|buffer_load)"),
    ("branch", r"^s_(cbranch|branch)"),
    ("salu", r"^s_"),
    ("other valu", r"^v_"),
])
WEIGHT = {"mfma 16x16x32 f16": 1.0, "mfma 32x32 f16": 1.0, "mfma 4x4x4 f16": 2.85, "mfma 4x4x1 f32": 2.85, "mfma other": 2.85, "transcendental": 2.8,
          "convert": 1.0, "relu (max)": 1.0, "packed fp32": 1.6, "fp32 arithmetic": 1.0, "move / select": 1.0, "integer / address": 1.0, "other valu": 1.0}


def kernel_body(path, name):
    lines = open(path).read().splitlines()
    start = None
    for i, l in enumerate(lines):
        m = re.match(r"^(_Z\w+):", l)
        if m:
            dem = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
            dem = dem.replace("(anonymous namespace)::", "")
            if name in dem:
                start = i
                break
    if start is None:
        sys.exit("kernel not found: " + name)
    end = next(i for i in range(start, len(lines)) if lines[i].strip().startswith("s_endpgm"))
    return lines[start:end + 1]


HEADER = {}   # block label -> label of the innermost loop header it belongs to


def blocks(body):
    out, cur, label = OrderedDict(), [], "entry"
    for l in body:
        m = re.match(r"^(\.LBB\d+_\d+):", l) or re.match(r"^; %(bb\.\d+):", l)
        if m:
            out[label] = cur
            cur, label = [], m.group(1)
            h = re.search(r"in Loop: Header=(BB\d+_\d+)", l)
            if h:
                HEADER[label] = ".L" + h.group(1)
            elif "Loop Header" in l:
                HEADER[label] = label
            continue
        s = l.strip()
        if not s or s.startswith(";") or s.startswith("."):
            continue
        cur.append(s.split(";")[0].strip())
    out[label] = cur
    return out


def classify(ins):
    op = ins.split()[0]
    for c, pat in CLASSES.items():
        if re.match(pat, op):
            return c
    return "other"


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    skip = set()
    for a in sys.argv[1:]:
        if a.startswith("--skip="):
            skip = set(a[7:].split(","))
    path, name = args[0], args[1]
    bl = blocks(kernel_body(path, name))
    labels = list(bl.keys())
    if len(args) >= 4:
        i0, i1 = labels.index(args[2]), labels.index(args[3])
    else:
        with_bar = [i for i, k in enumerate(labels) if any(x.startswith("s_barrier") for x in bl[k])]
        # the coupling loop: the block that holds two barriers (publish z0 | publish relu(h2))
        two = [i for i in with_bar if sum(x.startswith("s_barrier") for x in bl[labels[i]]) >= 2]
        hdr = HEADER.get(labels[two[-1] if two else with_bar[-1]], labels[two[-1] if two else with_bar[-1]])   # (a peeled copy comes first)
        members = [i for i, k in enumerate(labels) if HEADER.get(k) == hdr]
        i0, i1 = min(members), max(members)
    counts = OrderedDict((c, 0) for c in list(CLASSES) + ["other"])
    listing = []
    if "--fwd" in sys.argv:   # drop the sampling direction's affine block: the one that forms (z - shift) with negated packed adds
        skip |= {k for k in labels[i0:i1 + 1] if any(x.startswith("v_pk_add_f32") and "neg_lo" in x for x in bl[k])}
    for k in labels[i0:i1 + 1]:
        if k in skip:
            continue
        for ins in bl[k]:
            counts[classify(ins)] += 1
            listing.append((k, ins))
    if "--list" in sys.argv:
        for k, ins in listing:
            print("%-12s %s" % (k, ins))
    print("blocks %s .. %s%s" % (labels[i0], labels[i1], (" (skipped: %s)" % ",".join(sorted(skip))) if skip else ""))
    tot_valu = 0.0
    n_mfma = sum(v for c, v in counts.items() if c.startswith("mfma"))
    print("%-22s %6s %9s" % ("class", "count", "issue"))
    for c, v in counts.items():
        if not v:
            continue
        w = WEIGHT.get(c)
        tot_valu += v * w if w else 0.0
        print("%-22s %6d %9s" % (c, v, ("%.1f" % (v * w)) if w else "-"))
    n_valu = sum(v for c, v in counts.items() if c in WEIGHT and not c.startswith("mfma"))
    print("MFMA %d, other VALU %d (%.2f per MFMA), VALU-port issue units %.1f" % (n_mfma, n_valu, n_valu / max(n_mfma, 1), tot_valu))


if __name__ == "__main__":
    main()
