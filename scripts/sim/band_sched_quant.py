"""Design-time simulation: lock-step trips when the trips of a (step, group) unit come in multiples of Q."""
import sys
import numpy as np
sys.path.insert(0, "/root/repo")
from rails_amd import problems as P


def sched_group(cols_list, W0, nseg, seg, nsteps, Q, eager_fill=False):
    S = len(cols_list)
    ptr = [0] * S
    trips = 0
    units = 0
    for k in range(nsteps):
        lo = W0 + (k - nseg + 2) * seg
        hi = W0 + (k + 1) * seg
        lo_next = lo + seg if k < nsteps - 1 else 1 << 60
        forced = 0
        avail = []
        for s in range(S):
            c = cols_list[s]
            a = np.searchsorted(c, hi, "left") - ptr[s]
            f = np.searchsorted(c, lo_next, "left") - ptr[s]
            avail.append(a)
            forced = max(forced, f)
        T = -(-forced // Q) * Q
        if T > 0:
            units += 1
            trips += T
            for s in range(S):
                ptr[s] += min(avail[s], T)
    assert all(ptr[s] == len(cols_list[s]) for s in range(S))
    return trips, units


def main():
    m = 1 << 16
    bw = 4096
    rowptr, col, val = P.banded_random(m, 27, bw, seed=0)
    col = col.reshape(m, 27).astype(np.int64)
    R = 2816
    r0 = 20480
    rows = np.arange(r0, r0 + R)
    C = col[rows]
    W0 = r0 - bw
    for S, nseg, seg, Q in [(8, 5, 256, 1), (8, 5, 256, 2), (8, 5, 256, 4), (8, 6, 256, 4), (8, 9, 128, 4), (8, 4, 512, 4), (16, 5, 256, 4), (8, 10, 128, 1), (8, 3, 512, 4), (8, 3, 512, 1)]:
        nsteps = (R + 2 * bw) // seg + 2
        tot = units = ideal = 0
        for g in range(R // S):
            t, u = sched_group([C[i] for i in range(g * S, (g + 1) * S)], W0, nseg, seg, nsteps, Q)
            tot += t
            units += u
            ideal += 27
        print(f"S={S} nseg={nseg} seg={seg} Q={Q}: eff {ideal/tot:.3f}  units/group {units/(R//S):.1f}  trips/unit {tot/units:.2f}")


main()


def sched_group2(cols_list, W0, nseg, seg, nsteps, Q, full_eager):
    S = len(cols_list)
    ptr = [0] * S
    trips = 0
    units = 0
    for k in range(nsteps):
        lo = W0 + (k - nseg + 2) * seg
        hi = W0 + (k + 1) * seg
        lo_next = lo + seg if k < nsteps - 1 else 1 << 60
        while True:
            forced = 0
            avail = []
            left = 0
            for s in range(S):
                c = cols_list[s]
                a = np.searchsorted(c, hi, "left") - ptr[s]
                f = np.searchsorted(c, lo_next, "left") - ptr[s]
                avail.append(a)
                forced = max(forced, f)
                left += len(c) - ptr[s]
            T = 0
            if forced > 0:
                T = Q
            elif full_eager and sum(min(a, Q) for a in avail) >= full_eager * Q * S and left > 0:
                T = Q
            if T == 0:
                break
            units += 1
            trips += T
            for s in range(S):
                ptr[s] += min(avail[s], T)
    assert all(ptr[s] == len(cols_list[s]) for s in range(S))
    return trips, units


def main2():
    m = 1 << 16
    bw = 4096
    rowptr, col, val = P.banded_random(m, 27, bw, seed=0)
    col = col.reshape(m, 27).astype(np.int64)
    R = 2816
    r0 = 20480
    C = col[np.arange(r0, r0 + R)]
    W0 = r0 - bw
    for S, nseg, seg, Q, fe, order in [(8, 5, 256, 4, 0, "seq"), (8, 5, 256, 4, 1.0, "seq"), (8, 5, 256, 4, 0.9, "seq"), (8, 5, 256, 4, 0.8, "seq"), (8, 10, 128, 4, 0.9, "seq"), (8, 5, 256, 4, 0.9, "sorted"), (8, 5, 256, 4, 0, "sorted"), (8, 5, 256, 2, 0.9, "seq")]:
        nsteps = (R + 2 * bw) // seg + 2
        idx = np.arange(R)
        if order == "sorted":
            idx = np.argsort((C - W0).mean(1), kind="stable")
        tot = units = ideal = 0
        for g in range(R // S):
            t, u = sched_group2([C[i] for i in idx[g * S:(g + 1) * S]], W0, nseg, seg, nsteps, Q, fe)
            tot += t
            units += u
            ideal += 27
        print(f"S={S} nseg={nseg} seg={seg} Q={Q} full_eager={fe} {order}: eff {ideal/tot:.3f}  units/group {units/(R//S):.1f}")


main2()
