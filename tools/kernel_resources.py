#!/usr/bin/env python3
"""Registers, scratch and occupancy of every kernel in vr_kernels.hip as the compiler reports them (CPU only: hipcc
cross-compiles).  usage: python tools/kernel_resources.py [extra -D flags...]"""
import os
import re
import subprocess
import sys

CSRC = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "volumerenderer_amd", "csrc")
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fno-fast-math",
       "-fhip-fp32-correctly-rounded-divide-sqrt", "-fno-gpu-flush-denormals-to-zero", "-S", "--cuda-device-only", "-o", "/dev/null",
       "vr_kernels.hip", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[1:]
err = subprocess.run(cmd, cwd=CSRC, capture_output=True, text=True).stderr
rows, cur = [], None
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        name = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(anonymous namespace\)::", "", name)
        cur = {"name": re.sub(r"^void ", "", name).split("(")[0]}
        rows.append(cur)
        continue
    for key in ("TotalSGPRs", "VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
print("%-44s %5s %5s %5s %8s %5s" % ("kernel", "VGPR", "AGPR", "SGPR", "scratch", "occ"))
for r in rows:
    print("%-44s %5s %5s %5s %8s %5s" % (r["name"][:44], r.get("VGPRs"), r.get("AGPRs"), r.get("TotalSGPRs"), r.get("ScratchSize"), r.get("Occupancy")))
if not rows:
    sys.stderr.write(err)
    sys.exit(1)
