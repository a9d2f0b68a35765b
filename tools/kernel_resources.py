#!/usr/bin/env python3
"""Register / LDS / occupancy table of every kernel of one HIP source (hipcc -Rpass-analysis=kernel-resource-usage); no GPU needed.
  python tools/kernel_resources.py summertts_amd/csrc/resblock_bf3.hip [extra hipcc flags]"""
import re
import subprocess
import sys

src = sys.argv[1]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-mllvm", "-amdgpu-mfma-vgpr-form",
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[2:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = {}
rows = []
for ln in err.splitlines():
    m = re.search(r"remark: [^:]*:\d+:\d+:\s+(.*?) \[-Rpass", ln) or re.search(r"remark:\s+(.*?) \[-Rpass", ln)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith("Function Name:") or t.startswith("Name:"):
        if cur:
            rows.append(cur)
        cur = {"name": t.split(":", 1)[1].strip()}
    elif ":" in t:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
if cur:
    rows.append(cur)
dem = subprocess.run(["c++filt"] + [r["name"] for r in rows], capture_output=True, text=True).stdout.splitlines()
print(f"{'VGPR':>5} {'AGPR':>5} {'SGPR':>5} {'scratch':>8} {'occ':>4} {'LDS':>7}  kernel")
for r, d in zip(rows, dem):
    d = re.sub(r"\(.*", "", d).replace("void sts::", "")
    print(f"{r.get('VGPRs', '?'):>5} {r.get('AGPRs', '?'):>5} {r.get('TotalSGPRs', r.get('SGPRs', '?')):>5} {r.get('ScratchSize [bytes/lane]', '?'):>8} "
          f"{r.get('Occupancy [waves/SIMD]', '?'):>4} {r.get('LDS Size [bytes/block]', '?'):>7}  {d}")
