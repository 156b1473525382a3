"""In-tree build of librails_hip.so for gfx950 (hipcc cross-compiles without a GPU)."""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))


def build(jobs=6, verbose=False):
    cmd = ["make", "-C", os.path.join(_HERE, "csrc"), "-j%d" % jobs]
    if not verbose:
        cmd.insert(1, "-s")
    subprocess.check_call(cmd)
    return os.path.join(_HERE, "lib", "librails_hip.so")


if __name__ == "__main__":
    print(build(verbose=True))
