"""Round 3 probe (host side of a restart): pivoted QR of the live coefficient columns (430 x 480, compress()) and eig(T) (200 x 200) on the
GPU box's host with 1, 2, 4, 8 BLAS threads (scipy's OpenBLAS through threadpoolctl) -- would selective threading of the two big calls pay?"""
import time
import numpy as np
import scipy.linalg as sl
from threadpoolctl import threadpool_limits
g = np.random.default_rng(0)
A = g.standard_normal((430, 280)) @ g.standard_normal((280, 480))
T = g.standard_normal((200, 200)); T = T + T.T
M = g.standard_normal((200, 200)) - 12 * np.eye(200)
F = g.standard_normal((200, 16))
for nt in (1, 2, 4, 8):
    with threadpool_limits(limits=nt):
        for name, fn in (("pivoted QR 430x480", lambda: sl.qr(A, pivoting=True, mode="economic")), ("eigh 200", lambda: sl.eigh(T)),
                         ("lu_factor 200 + 23 solves x16", lambda: [sl.lu_solve(sl.lu_factor(M), F) for _ in range(1)] + [sl.lu_solve(lu, F) for lu in [sl.lu_factor(M)] for _ in range(23)]),
                         ("dgemm 200x200x200", lambda: M @ M)):
            fn(); fn()
            ts = []
            for _ in range(15):
                t = time.perf_counter(); fn(); ts.append(time.perf_counter() - t)
            print("threads %d  %-32s median %.3f ms  min %.3f ms" % (nt, name, 1e3 * np.median(ts), 1e3 * min(ts)), flush=True)
