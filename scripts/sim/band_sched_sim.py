"""Design-time simulation for the LDS-window SpMM: how many lock-step trips a group of S slots (one row each)
needs when the window of X rows slides through LDS in steps.  Not product code."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from rails_amd import problems as P


def sched_group(cols_list, W0, nb, wstep, nsteps, lazy=True):
    """cols_list: list of sorted arrays of columns (one per slot).  Buffer at step k holds [W0 + k*wstep, W0 + k*wstep + nb).
    Returns total trips."""
    S = len(cols_list)
    ptr = [0] * S
    trips = 0
    units = 0
    for k in range(nsteps):
        lo = W0 + k * wstep
        hi = lo + nb
        lo_next = lo + wstep if k < nsteps - 1 else 1 << 60
        forced = 0
        avail = []
        for s in range(S):
            c = cols_list[s]
            a = np.searchsorted(c, hi, "left") - ptr[s]
            f = np.searchsorted(c, lo_next, "left") - ptr[s]
            avail.append(a)
            forced = max(forced, f)
        T = forced if lazy else max(avail)
        if T > 0:
            units += 1
            trips += T
            for s in range(S):
                ptr[s] += min(avail[s], T)
    assert all(ptr[s] == len(cols_list[s]) for s in range(S)), "unprocessed"
    return trips, units


def main():
    m = 1 << 16
    bw = 4096
    rowptr, col, val = P.banded_random(m, 27, bw, seed=0)
    col = col.reshape(m, 27).astype(np.int64)
    for R, S, nb, wstep, order in [(2048, 8, 1024, 256, "seq"), (2048, 8, 1024, 128, "seq"), (2048, 16, 1024, 256, "seq"), (2048, 8, 768, 256, "seq"),
                                    (2048, 8, 1024, 256, "sorted"), (2048, 16, 1024, 256, "sorted"), (2048, 8, 2048, 256, "seq"), (2048, 8, 1024, 512, "seq")]:
        r0 = 20480
        rows = np.arange(r0, r0 + R)
        W0 = r0 - bw
        nsteps = (R + 2 * bw + 1 - nb + wstep - 1) // wstep + 1
        C = col[rows]
        if order == "sorted":
            # sort rows by mean column offset relative to W0 (profile proxy)
            key = (C - W0).mean(1)
            rows_o = np.argsort(key)
        else:
            rows_o = np.arange(R)
        tot = 0
        units = 0
        ideal = 0
        for g in range(R // S):
            sel = rows_o[g * S:(g + 1) * S]
            t, u = sched_group([C[i] for i in sel], W0, nb, wstep, nsteps)
            tot += t
            units += u
            ideal += 27
        print(f"R={R} S={S} nb={nb} wstep={wstep} order={order}: trips/ideal = {tot/ideal:.3f} (eff {ideal/tot:.3f}), units per group {units/(R//S):.1f}, steps {nsteps}")


main()
