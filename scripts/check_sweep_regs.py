#!/usr/bin/env python3
"""The sweep kernel's inline assembly owns v24-v255 (rails_amd/csrc/spmm_sweep.hip, register plan); amdgpu_num_vgpr(24)
keeps the compiler below them only as long as its own values fit.  This check compiles the kernels to assembly and fails
when compiler-generated code (outside the ;;#ASMSTART ... ;;#ASMEND regions) names a reserved register or touches scratch.
Run by __graft_entry__.build() and tests/test_sweep_plan.py."""
import os
import re
import subprocess
import sys
import tempfile

CSRC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "rails_amd", "csrc")
RESERVED = re.compile(r"\bv(\d+)\b|\bv\[(\d+):(\d+)\]")


def check(first_reserved=24):
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "k.s")
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I../include", "-I.", "-S",
                        "--cuda-device-only", "spmm_sweep.hip", "-o", out], cwd=CSRC, check=True, stderr=subprocess.DEVNULL)
        text = open(out).read().splitlines()
    bad, kernel, in_asm, kernels = [], None, False, 0
    for line in text:
        if re.match(r"^_ZN.*k_spmm_sweep.*:", line):
            kernel, kernels = line.split(":")[0], kernels + 1
        elif "s_endpgm" in line:
            kernel = None
        if ";;#ASMSTART" in line:
            in_asm = True
        elif ";;#ASMEND" in line:
            in_asm = False
        if kernel is None or in_asm:
            continue
        code = line.split(";")[0]
        if "scratch_" in code:
            bad.append(line)
        for m in RESERVED.finditer(code):
            hi = int(m.group(1)) if m.group(1) else int(m.group(3))
            if hi >= first_reserved:
                bad.append(line)
                break
    return kernels, bad


if __name__ == "__main__":
    kernels, bad = check()
    if not kernels or bad:
        print("sweep kernel: %d kernels; compiler code uses reserved registers or scratch:" % kernels)
        print("\n".join(bad[:10]))
        sys.exit(1)
    print("sweep kernel register plan: ok (%d kernels)" % kernels)
