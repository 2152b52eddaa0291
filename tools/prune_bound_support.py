"""How much of k_support_lds's work could an EXACT lower-bound pruning remove?  (CPU study, nothing here ships.)

computeMatchingDisparity (libelas/src/elas.cpp:395-433) needs only the smallest key E<<16|d and the energy of the
second smallest.  A disparity whose partial SAD over the first k of the four descriptors already exceeds the row's
second-best energy can be dropped.  In k_support_lds a wave searches FOUR candidates at once (16 lanes each, 16
disparities per trip), so a trip's remaining work can only be skipped when all 64 lanes agree.

This script replays the kernel's wave / trip structure on the descriptors of the four urban crops (oracle DESC taps)
and counts the SAD instructions that could be skipped
  ideal      with the final second-best energy of every candidate known in advance (an upper bound for ANY scheme),
  running    with the second-best energy among the trips already evaluated, shared across the row after every trip
             for free (an upper bound for any scheme that learns the threshold as it goes; trips in the kernel's
             rising-address order),
for the forward search (every candidate) and the backward search (candidates with a forward match).
    python tools/prune_bound_support.py > profiles/r05_prune_bound_support.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H  # noqa: E402

KSB, KNW = 64, 8            # candidates per block, waves per block (k_support_lds<64, 512>)
STEP, DMAX = 5, 255


def energies(own, oth, v, us, right, W):
    """partial energies [4][ncand][256] (cumulative over the four descriptors), inf outside the valid range"""
    d = np.arange(DMAX + 1)
    out = np.zeros((4, len(us), DMAX + 1), np.int32)
    k = 0
    acc = np.zeros((len(us), DMAX + 1), np.int32)
    for dy in (-2, 2):
        for dx in (-2, 2):
            a = own[v + dy, us + dx].astype(np.int16)                       # [nc,16]
            xs = us[:, None] + dx + (d[None, :] if right else -d[None, :])  # [nc,256]
            ok = (xs >= 0) & (xs < W)
            b = oth[v + dy, np.clip(xs, 0, W - 1)].astype(np.int16)         # [nc,256,16]
            acc = acc + np.abs(a[:, None, :] - b).sum(axis=2)
            out[k] = acc
            k += 1
    return out


def search(own, oth, v, us, right, W, tex_thr):
    """per candidate: active flag, dmax, cumulative partial energies"""
    tex = np.abs(own[v, us].astype(np.int32) - 128).sum(axis=1)
    dmax = np.minimum(DMAX, (W - us - 5) if right else (us - 5))
    act = (us >= 5) & (us <= W - 6) & (tex >= tex_thr) & (dmax >= 10)
    E = energies(own, oth, v, us, right, W)
    return act, dmax, E


def tally(act, dmax, E, order_desc, groups, stats):
    """groups: list of arrays of 4 candidate indices (one wave).  Counts trips and what could be skipped."""
    big = np.int32(1 << 30)
    for g in groups:
        a = act[g]
        if not a.any():
            continue
        top = int(dmax[g][a].max())
        T = top // 16 + 1
        full = E[3][g].astype(np.int64)                                   # [4,256]
        dd = np.arange(DMAX + 1)[None, :]
        inr = a[:, None] & (dd <= dmax[g][:, None])
        fullm = np.where(inr, full, big)
        srt = np.sort(fullm, axis=1)
        e2_final = srt[:, 1]                                              # ideal threshold per candidate
        trips = list(range(T))
        if order_desc:
            trips = trips[::-1]
        run1 = np.full(4, big, np.int64)
        run2 = np.full(4, big, np.int64)
        for t in trips:
            sl = slice(16 * t, 16 * t + 16)
            m = inr[:, sl]
            if not m.any():
                continue
            stats["trips"] += 1
            for mode, thr in (("ideal", e2_final), ("running", run2)):
                done = False
                for k in (0, 1, 2):                                      # after k+1 descriptors
                    lb = np.where(m, E[k][g][:, sl], big)
                    if (lb > thr[:, None]).all():
                        stats[mode][k] += 1
                        done = True
                        break
                if not done:
                    stats[mode][3] += 1
            # running thresholds learn this trip's energies (all of them: an upper bound on what is known)
            vals = np.where(m, full[:, sl], big)
            allv = np.sort(np.concatenate([vals, run1[:, None], run2[:, None]], axis=1), axis=1)
            run1, run2 = allv[:, 0], allv[:, 1]


def main():
    prm = H.robotics()
    names = ["urban1_1242x375", "urban2_1242x375", "urban3_1242x375", "urban4_1242x375"]
    tot = {"fwd": {"trips": 0, "ideal": [0, 0, 0, 0], "running": [0, 0, 0, 0]},
           "bwd": {"trips": 0, "ideal": [0, 0, 0, 0], "running": [0, 0, 0, 0]}}
    for name in names:
        l, r = H.golden_pair(name)
        run = H.oracle_elas_run(prm, l, r)
        Hh, W = l.shape
        d1 = run[H.DESC1].reshape(Hh, W, 16)
        d2 = run[H.DESC2].reshape(Hh, W, 16)
        dcan = run[H.DCAN_RAW]
        Wc, Hc = W // STEP, Hh // STEP          # (lattice as the reference sizes it)
        for vc in range(1, Hc):
            v = vc * STEP
            if v < 5 or v > Hh - 6:
                continue
            ucs = np.arange(Wc)
            us = ucs * STEP
            act, dmax, E = search(d1, d2, v, us, False, W, prm.support_texture)
            act &= ucs > 0
            # waves of the forward rounds: block = 64 candidates; wave w, round rep: c = rep*32 + w + 8*grp
            groups = []
            for c0 in range(0, Wc, KSB):
                for rep in range(KSB // (4 * KNW)):
                    for w in range(KNW):
                        g = np.array([c0 + rep * 4 * KNW + w + KNW * grp for grp in range(4)])
                        g = g[g < Wc]
                        if len(g) < 4:
                            g = np.concatenate([g, np.full(4 - len(g), g[0] if len(g) else 0)])
                            a2 = act.copy()
                        groups.append(g)
            tally(act, dmax, E, True, groups, tot["fwd"])
            # forward results (for the backward candidates): smallest key, ratio test
            full = np.where((np.arange(DMAX + 1)[None, :] <= dmax[:, None]) & act[:, None], E[3], 1 << 30).astype(np.int64)
            key = full * 65536 + np.arange(DMAX + 1)[None, :]
            ks = np.sort(key, axis=1)
            e1, e2, dbest = ks[:, 0] >> 16, ks[:, 1] >> 16, ks[:, 0] & 0xFFFF
            good = act & (e1 < (1 << 29)) & (e2 < (1 << 29)) & \
                (e1.astype(np.float32) < np.float32(prm.support_threshold) * e2.astype(np.float32))
            todo = np.nonzero(good)[0]
            if len(todo) == 0:
                continue
            ub = us[todo] - dbest[todo]
            actb, dmaxb, Eb = search(d2, d1, v, ub, True, W, prm.support_texture)
            # compacted per block: consecutive todo entries of one block, four per wave
            groups = []
            for c0 in range(0, Wc, KSB):
                idx = np.nonzero((todo >= c0) & (todo < c0 + KSB))[0]
                for i in range(0, len(idx), 4):
                    g = idx[i:i + 4]
                    if len(g) < 4:
                        g = np.concatenate([g, np.full(4 - len(g), g[0])])
                    groups.append(g)
            tally(actb, dmaxb, Eb, False, groups, tot["bwd"])
        print("# %s done" % name, flush=True)
    print("exact lower-bound pruning in k_support_lds: what the wave-wide vote leaves (four urban crops, ROBOTICS)")
    for side in ("fwd", "bwd"):
        s = tot[side]
        print("%s search: %d wave-trips (16 v_sad_hi_u8 + 4 ds_read_b128 each)" % (side, s["trips"]))
        for mode in ("ideal", "running"):
            c = s[mode]
            saved = (12 * c[0] + 8 * c[1] + 4 * c[2]) / (16.0 * s["trips"])
            print("  %-8s skipped after 1/2/3 descriptors: %5.1f %% / %5.1f %% / %5.1f %%   never: %5.1f %%   -> SADs saved %5.1f %%"
                  % (mode, 100.0 * c[0] / s["trips"], 100.0 * c[1] / s["trips"], 100.0 * c[2] / s["trips"],
                     100.0 * c[3] / s["trips"], 100.0 * saved))
    print("(ideal = final second-best energy known in advance; running = second best of the trips evaluated so far, "
          "shared across the row after every trip at no cost.  A vote costs ~3 instructions per test point and breaks "
          "the four-reads-in-flight batching of a trip; a trip is ~24 VALU instructions.)")


if __name__ == "__main__":
    main()
