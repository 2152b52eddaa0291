#!/usr/bin/env python3
"""k_match_list's instruction split from the probe builds' counters (tools/gpu_match_split.sh -> match_split.txt).

    python tools/match_split_summary.py <raw before> <raw after>  > profiles/r06_match_split.txt

Every probe build is the product with one part of the kernel cut away (SVH_ML_PROBE in csrc/elas_kernels.hip); the
counters are rocprofv3 --pmc per-dispatch averages of one 4-pair launch (kernels serialised: one lane, groups of the
four 1242x375 urban crops).  Parts = differences of the cumulative builds."""
import re
import sys


def parse(path):
    probes, cur, events, build = {}, None, {}, ""
    for line in open(path):
        if line.startswith("#"):
            build = line[1:].strip()
        m = re.match(r"== probe (\d+)", line)
        if m:
            cur = int(m.group(1))
            probes[cur] = {}
            continue
        if line.startswith("== event"):
            cur = "ev"
            continue
        m = re.match(r"\s+(\w+)\s+([\d.]+)\s*$", line)
        if m and cur == "ev":
            events[m.group(1)] = int(float(m.group(2)))
        elif m and cur is not None:
            probes[cur][m.group(1)] = float(m.group(2))
    return build, probes, events


def split(p):
    v = lambda n: p[n]["SQ_INSTS_VALU"]
    full, lr = v(0), v(0) - v(6)
    parts = [
        ("staging (rows from the Sobel planes) + prologue", v(1)),
        ("own descriptor, texture test, live vote, result stores", v(2) - v(1) - lr),
        ("L/R pass (E12, fused)", lr),
        ("owner -> plane fetch, plan (cell record, d_plane, cell / band tests), votes", v(3) - v(2)),
        ("cell candidates (trips of four: 16 v_sad_hi_u8 + addresses + min3 each)", v(4) - v(3)),
        ("plane band (five candidates with their priors) + rank decode", v(5) - v(3)),
        ("checked redo of the cold wave-pixels", v(0) - v(8)),
    ]
    rest = full - sum(x for _, x in parts)
    parts.append(("not attributed (probe builds schedule differently)", rest))
    return full, parts


def main(before, after):
    b0, p0, e0 = parse(before)
    b1, p1, e1 = parse(after)
    f0, s0 = split(p0)
    f1, s1 = split(p1)
    wp = e1.get("wave_pixels", 60000)
    rows = 58219           # 4 pairs x 375 rows x 2 maps x 1242 / 64: the verdict's "wave-rows"
    print("k_match_list<true, 5>: where the wave-level VALU instructions of one 4-pair launch go (1242x375 urban crops, ROBOTICS)")
    print("before: %s   (round 5's kernel; profiles/r06_match_split_raw_baseline.txt)" % b0)
    print("after:  %s   (%s)" % (b1, after))
    print("method: SQ_INSTS_VALU of the product (probe 0) and of eight cut-down builds (tools/Makefile `probe`, SVH_ML_PROBE = 1..8),")
    print("        rocprofv3 --pmc per-dispatch averages, kernels serialised; parts = differences of the cumulative builds.")
    print()
    print("%-82s %10s %6s %7s   %10s %6s %7s   %6s" % ("part", "before", "%", "/pixel", "after", "%", "/pixel", "change"))
    for (n, a), (_, c) in zip(s0, s1):
        print("%-82s %10d %5.1f%% %7.1f   %10d %5.1f%% %7.1f   %+5.0f%%" % (n, a, 100 * a / f0, a / rows, c, 100 * c / f1, c / rows,
                                                                       100 * (c - a) / a if a else 0))
    print("%-82s %10d %5.1f%% %7.1f   %10d %5.1f%% %7.1f   %+5.0f%%" % ("whole kernel", f0, 100, f0 / rows, f1, 100, f1 / rows, 100 * (f1 - f0) / f0))
    print()
    print("(/pixel = wave-level instructions per wave-row of 64 pixels: %d rows; the review's figure for round 5 was 275)" % rows)
    sad = 16 * (e1.get("trips_plain", 0) + e1.get("trips_excl", 0) + e1.get("trips_edge", 0)) + 20 * e1.get("fast_wp", 0) + 4 * wp
    print()
    print("other counters of the launch      before        after")
    for k in ("SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_WAVE_CYCLES"):
        print("   %-28s %12d %12d" % (k, p0[0].get(k, 0), p1[0].get(k, 0)))
    print()
    print("events of the launch (probe 7: counters in the kernel; the same four pairs before and after -- the kernel's control flow")
    print("is data, not code: wave-pixels, live ones, hot / cold split and trips did not move, except for the edge form)")
    for k in e1:
        if k.startswith("lane_trips"):
            continue
        print("   %-24s %10s" % (k, e1[k]) + ("   (before: %s)" % e0[k] if k in e0 and e0[k] != e1[k] and not k.startswith("live_lanes") else ""))
    print()
    print("SAD instructions the launch needs: %d (16 per trip of four candidates, 20 per band, 4 per texture test) = %.1f %% of "
          "the kernel after, %.1f %% before" % (sad, 100.0 * sad / f1, 100.0 * sad / f0))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
