#!/usr/bin/env python3
"""Pack the data set the reference's own test holds for its application problem (matlab/test/test_MOC.m:94-134: the MOC ocean
model of matlab/DataErik/, n = 8*8*4*6 = 1536 unknowns) into tests/golden/moc_erik.npz.

Run in the build container only (needs /root/reference):   python tests/golden/make_moc_fixture.py

The files are plain text columns of numbers and are read as numbers (numpy.loadtxt); nothing else is taken from the reference:
  Ap1.info  n nnz            Ap1.beg   n+1 row starts (1-based)      Ap1.jco  nnz column indices (1-based)
  Ap1.co    nnz values       Bp1.co    n diagonal entries of M      Frcp1.co n entries of the forcing F
"""
import os

import numpy as np

SRC = "/root/reference/matlab/DataErik"
DST = os.path.join(os.path.dirname(os.path.abspath(__file__)), "moc_erik.npz")


def main():
    n, nnz = (int(x) for x in np.loadtxt(os.path.join(SRC, "Ap1.info")))
    beg = np.loadtxt(os.path.join(SRC, "Ap1.beg")).astype(np.int64)
    jco = np.loadtxt(os.path.join(SRC, "Ap1.jco")).astype(np.int32)
    co = np.loadtxt(os.path.join(SRC, "Ap1.co"), dtype=np.float64)
    mdiag = np.loadtxt(os.path.join(SRC, "Bp1.co"), dtype=np.float64)
    frc = np.loadtxt(os.path.join(SRC, "Frcp1.co"), dtype=np.float64)
    assert beg.size == n + 1 and beg[0] == 1 and beg[-1] == nnz + 1 and jco.size == nnz == co.size and mdiag.size == n == frc.size
    np.savez_compressed(DST, n=n, beg=beg, jco=jco, co=co, mdiag=mdiag, frc=frc)
    print("wrote", DST, os.path.getsize(DST), "bytes")


if __name__ == "__main__":
    main()
