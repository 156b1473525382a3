#!/usr/bin/env python3
"""Disassemble one kernel of the built library (the gfx950 code objects inside rails_amd/lib/librails_hip.so):
    python scripts/disasm_kernel.py <substring of the mangled name> [out.s]"""
import os, re, struct, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"


def main():
    want = sys.argv[1]
    data = open(os.path.join(ROOT, "rails_amd", "lib", "librails_hip.so"), "rb").read()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    pos = data.find(magic)
    out = []
    with tempfile.TemporaryDirectory() as d:
        n = 0
        while pos >= 0:
            (count,) = struct.unpack_from("<Q", data, pos + len(magic))
            q = pos + len(magic) + 8
            for _ in range(count):
                off, size, tlen = struct.unpack_from("<QQQ", data, q)
                triple = data[q + 24:q + 24 + tlen].decode()
                q += 24 + tlen
                if "gfx950" not in triple or size == 0:
                    continue
                path = os.path.join(d, "co%d.elf" % n)
                n += 1
                open(path, "wb").write(data[pos + off:pos + off + size])
                txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", path], capture_output=True, text=True).stdout
                cur, keep = None, False
                for line in txt.splitlines():
                    m = re.match(r"^[0-9a-f]+ <(.+)>:", line)
                    if m:
                        keep = want in m.group(1)
                        if keep:
                            out.append("; ---- " + m.group(1))
                        continue
                    if keep:
                        out.append(line)
            pos = data.find(magic, pos + 1)
    text = "\n".join(out)
    if len(sys.argv) > 2:
        open(sys.argv[2], "w").write(text)
    else:
        print(text)


if __name__ == "__main__":
    main()
