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
        subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I../../include", "-I../include", "-I.", "-S",
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


def kernel_metadata(so_path=None):
    """name -> {vgpr_count, agpr_count, sgpr_count, scratch} of every kernel in the gfx950 code objects of the BUILT library: the
    clang offload bundles inside the .so are cut out and their AMDGPU metadata notes read with the toolchain's own llvm-readelf."""
    import struct

    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    readelf = os.path.join(os.path.dirname(os.path.realpath(hipcc)), "..", "lib", "llvm", "bin", "llvm-readelf")
    if not os.path.exists(readelf):
        readelf = "/opt/rocm/lib/llvm/bin/llvm-readelf"
    so_path = so_path or os.path.join(CSRC, "..", "lib", "librails_hip.so")
    data = open(so_path, "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out = {}
    pos = data.find(magic)
    with tempfile.TemporaryDirectory() as d:
        n_obj = 0
        while pos >= 0:
            (count,) = struct.unpack_from("<Q", data, pos + len(magic))
            q = pos + len(magic) + 8
            for _ in range(count):
                off, size, tlen = struct.unpack_from("<QQQ", data, q)
                triple = data[q + 24:q + 24 + tlen].decode()
                q += 24 + tlen
                if "gfx950" not in triple or size == 0:
                    continue
                path = os.path.join(d, "co%d.elf" % n_obj)
                n_obj += 1
                open(path, "wb").write(data[pos + off:pos + off + size])
                txt = subprocess.run([readelf, "--notes", path], capture_output=True, text=True).stdout
                cur = {}
                for line in txt.splitlines():
                    m = re.match(r"\s*-?\s*\.(agpr_count|vgpr_count|sgpr_count|private_segment_fixed_size|name):\s*(\S+)", line)
                    if not m:
                        continue
                    key, val = m.group(1), m.group(2)
                    if key == "agpr_count" and cur:  # (.agpr_count is the first key of a kernel's entry)
                        if "name" in cur:
                            out[cur["name"]] = cur
                        cur = {}
                    cur[key] = val if key == "name" else int(val)
                if "name" in cur:
                    out[cur["name"]] = cur
            pos = data.find(magic, pos + 1)
    return out


def check_built(first_reserved=24):
    """The shipped object, not a separate compile: every sweep kernel allocates the whole 256-register file (the assembly's share
    included), keeps no accumulation registers (a spill of the compiler's 24 into AGPRs would show as agpr_count > 0) and no scratch."""
    meta = kernel_metadata()
    sweep = {k: v for k, v in meta.items() if "k_spmm_sweep" in k}
    bad = []
    for name, v in sweep.items():
        if v.get("agpr_count", 0) != 0 or v.get("private_segment_fixed_size", 0) != 0 or v.get("vgpr_count", 0) > 256:
            bad.append("%s: vgpr %s agpr %s scratch %s" % (name, v.get("vgpr_count"), v.get("agpr_count"), v.get("private_segment_fixed_size")))
    return len(sweep), bad


if __name__ == "__main__":
    kernels, bad = check()
    nb, badb = check_built()
    if not nb or badb:
        print("sweep kernels of the built library (%d): register file / scratch not as planned:" % nb)
        print("\n".join(badb[:10]))
        sys.exit(1)
    print("built library: %d sweep kernels, no AGPRs, no scratch" % nb)
    if not kernels or bad:
        print("sweep kernel: %d kernels; compiler code uses reserved registers or scratch:" % kernels)
        print("\n".join(bad[:10]))
        sys.exit(1)
    print("sweep kernel register plan: ok (%d kernels)" % kernels)
