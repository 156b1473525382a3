#!/usr/bin/env python3
"""Per-kernel register / occupancy report of one HIP source (compile only: works without a GPU).
usage: python scripts/kernel_resources.py rails_amd/csrc/spmm.hip [name filter] [extra hipcc flags...]"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-I" + ROOT + "/include", "-I" + ROOT + "/rails_amd/include",
       "-I" + ROOT + "/rails_amd/csrc", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + sys.argv[3:]
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur, d = None, {}
for l in out.splitlines():
    m = re.search(r"Function Name: (\S+)", l)
    if m:
        cur, d = m.group(1), {}
        continue
    m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", l)
    if m and cur:
        d[m.group(1).strip()] = m.group(2)
        if m.group(1).strip().startswith("LDS Size"):
            name = subprocess.run(["c++filt", cur], capture_output=True, text=True).stdout.strip()
            name = name.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
            if flt in name:
                print("%-44s VGPR %4s AGPR %3s spill %3s waves/SIMD %s scratch %s" % (name[:44], d.get("VGPRs"), d.get("AGPRs"), d.get("VGPRs Spill"),
                                                                                   d.get("Occupancy"), d.get("ScratchSize")))
