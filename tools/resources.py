#!/usr/bin/env python3
"""Register / spill / occupancy table of every kernel of pt_kernels.hip (fast arithmetic build unless --exact).
    python tools/resources.py [--exact] [extra hipcc -D flags]"""
import re, subprocess, sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
exact = "--exact" in sys.argv
extra = [a for a in sys.argv[1:] if a != "--exact"]
# the three translation units of the library's build, each with its options (pathtrace_amd/csrc/Makefile)
base = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-fno-fast-math",
        f"-DPT_MATH_EXACT={1 if exact else 0}", "-Rpass-analysis=kernel-resource-usage", "-c", "pt_kernels.hip", "-o", "/dev/null"]
units = [["-fno-slp-vectorize", "-DPT_TU=1"], ["-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-DPT_TU=2"], ["-DPT_TU=3"]]
out = ""
for u in units:
    out += subprocess.run(base + u + extra, cwd=os.path.join(ROOT, "pathtrace_amd", "csrc"), stderr=subprocess.PIPE, stdout=subprocess.PIPE, text=True).stderr
rows, cur = [], None
for line in out.splitlines():
    m = re.search(r"remark: +(.*?) \[-Rpass", line)
    if not m:
        if "error" in line:
            print(line)
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        cur = {"name": t.split(":", 1)[1].strip()}
        rows.append(cur)
    elif cur is not None and ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
def short(n):
    d = subprocess.run(["c++filt", n], stdout=subprocess.PIPE, text=True).stdout.strip()
    d = re.sub(r"^void ptk_\w+_impl::", "", d)
    return re.sub(r"\(.*$", "", d)
print(f"{'kernel':58s} {'VGPR':>5s} {'spill':>5s} {'scratch':>7s} {'SGPR':>5s} {'occ':>3s} {'LDS':>6s}")
for r in rows:
    print(f"{short(r['name']):58s} {r.get('VGPRs','?'):>5s} {r.get('VGPRs Spill','?'):>5s} {r.get('ScratchSize [bytes/lane]','?'):>7s} "
          f"{r.get('TotalSGPRs','?'):>5s} {r.get('Occupancy [waves/SIMD]','?'):>3s} {r.get('LDS Size [bytes/block]','?'):>6s}")
