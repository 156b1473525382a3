#!/usr/bin/env python3
"""Where the host-side projected Lyapunov solve spends its time (rails_sb03md = dgees + 2 dgemm + dtrsyl + 2 dgemm): per phase at the
sizes of the C3 loop.  OPENBLAS_NUM_THREADS=1 python scripts/host_lyap_bench.py"""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import rails_amd  # noqa: E402
from scipy.linalg import lapack  # noqa: E402


def best(f, reps=7):
    ts = []
    for _ in range(reps):
        t = time.perf_counter()
        f()
        ts.append(time.perf_counter() - t)
    return 1e3 * min(ts)


lib = rails_amd.load()
dp = C.POINTER(C.c_double)
rng = np.random.default_rng(0)
for n in (128, 144, 160, 176, 192, 200, 256):
    A = np.asfortranarray(-3 * np.eye(n) + rng.standard_normal((n, n)) / np.sqrt(n))
    Bm = rng.standard_normal((n, 16))
    Cm = np.asfortranarray(-(Bm @ Bm.T))

    def sb():
        Ap, X = A.copy(order="F"), Cm.copy(order="F")
        scale, info = C.c_double(0), C.c_int(0)
        lib.rails_sb03md(b"C", b"X", b"N", b"T", n, Ap.ctypes.data_as(dp), n, X.ctypes.data_as(dp), n, C.byref(scale), C.byref(info))
    t_sb = best(sb)
    Sm = rng.standard_normal((n, n))
    As = np.asfortranarray(-(Sm @ Sm.T) / n - np.eye(n))

    def sbs():
        Ap, X = As.copy(order="F"), Cm.copy(order="F")
        scale, info = C.c_double(0), C.c_int(0)
        lib.rails_sb03md(b"C", b"X", b"N", b"T", n, Ap.ctypes.data_as(dp), n, X.ctypes.data_as(dp), n, C.byref(scale), C.byref(info))
    t_sym = best(sbs)
    t_gees = best(lambda: lapack.dgees(lambda r, i: 0, A, sort_t=0))
    t_hrd = best(lambda: lapack.dgehrd(A))
    S, sdim, wr, wi, U, work, info = lapack.dgees(lambda r, i: 0, A, sort_t=0)
    F = np.asfortranarray(U.T @ Cm @ U)
    t_syl = best(lambda: lapack.dtrsyl(S, S, F, trana="N", tranb="T"))
    t_gemm = best(lambda: U.T @ Cm @ U)
    print("n=%3d  rails_sb03md %6.2f ms (symmetric A: %5.2f) | dgees %6.2f (dgehrd %5.2f) | dtrsyl %5.2f | 2 dgemm %5.2f" % (n, t_sb, t_sym, t_gees, t_hrd, t_syl, t_gemm), flush=True)

# restart-side host numerics: eig(T) of the k x k solution and the pivoted QR of compress()
for n in (128, 200, 256):
    Sm = rng.standard_normal((n, n))
    T = np.asfortranarray(Sm + Sm.T)
    w = np.zeros(n)

    def ev():
        Tp, info = T.copy(order="F"), C.c_int(0)
        lib.rails_dsyev(b"V", b"U", n, Tp.ctypes.data_as(dp), n, w.ctypes.data_as(dp), C.byref(info))
    t_ev = best(ev)
    t_evr = best(lambda: lapack.dsyevr(T))
    t_evd = best(lambda: lapack.dsyevd(T))
    print("n=%3d  rails_dsyev %5.2f ms | scipy dsyevd %5.2f | dsyevr %5.2f" % (n, t_ev, t_evd, t_evr), flush=True)
for dim, ncols, rk in ((354, 300, 160), (347, 308, 292), (450, 330, 300)):
    Cc = np.asfortranarray(rng.standard_normal((dim, rk)) @ rng.standard_normal((rk, ncols)))  # rank rk of ncols columns
    Q = np.zeros((dim, min(dim, ncols)), order="F")

    def rb():
        Cp, rank, info = Cc.copy(order="F"), C.c_int(0), C.c_int(0)
        lib.rails_range_basis(dim, ncols, Cp.ctypes.data_as(dp), dim, C.c_double(1e-14), Q.ctypes.data_as(dp), dim, C.byref(rank), C.byref(info))
        return rank.value
    print("range_basis %d x %d (rank %d): %5.2f ms" % (dim, ncols, rb(), best(rb)), flush=True)
