"""Register / occupancy table of the kernels of one .hip source:  python tools/kernel_resources.py nf_train.hip k_mm_ [extra hipcc flags]
(hipcc -Rpass-analysis=kernel-resource-usage, cross-compiled for gfx950; no GPU needed)."""
import re, subprocess, sys, os
src, pat = sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ""
csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "noise_flow_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-munsafe-fp-atomics",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/tmp/_kr.o"] + sys.argv[3:]
out = subprocess.run(cmd, cwd=csrc, capture_output=True, text=True).stderr
cur = None
rows = {}
for line in out.splitlines():
    m = re.search(r"remark: (?:\s*)Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[bytes/\w+\])?: (\d+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = int(m.group(2))
print("%-5s %-5s %-4s %-6s %-5s  %s" % ("VGPR", "AGPR", "occ", "spill", "SGPR", "kernel"))
for k, r in rows.items():
    if pat in k:
        print("%-5d %-5d %-4d %-6d %-5d  %s" % (r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("Occupancy", -1), r.get("VGPRs Spill", -1), r.get("TotalSGPRs", -1),
                                                  re.sub(r"\(anonymous namespace\)::", "", k)[:110]))
