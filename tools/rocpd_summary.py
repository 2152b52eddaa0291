#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd database (the default output of
`rocprofv3 --kernel-trace --stats`) as a per-kernel table: calls, total, average,
min, max (us), share of GPU kernel time, VGPR/SGPR/LDS.  Used to turn the
gpurun_out/ scratch databases into the small text summaries kept in profiles/."""
import sqlite3
import sys


def main(path):
    db = sqlite3.connect(path)
    rows = db.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(grid_x*grid_y*grid_z), "
        "max(workgroup_x*workgroup_y*workgroup_z) from kernels group by name order by sum(duration) desc"
    ).fetchall()
    total = sum(r[2] for r in rows) or 1
    print("%-58s %7s %11s %9s %9s %9s %6s %5s %5s %6s %9s" % (
        "kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct", "vgpr", "sgpr", "lds", "threads"))
    for name, n, tot, avg, mn, mx, vg, sg, lds, grid, wg in rows:
        import re as _re; m = _re.search(r"(k_\w+)(<[^>]*>)?", name); short = (m.group(0) if m else name)[:58]
        print("%-58s %7d %11.1f %9.2f %9.2f %9.2f %6.2f %5s %5s %6s %9s" % (
            short, n, tot / 1e3, avg / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total, vg, sg, lds, grid))
    # fraction of the span during which at least one kernel was running (sweep over intervals)
    iv = db.execute("select start, end from kernels order by start").fetchall()
    busy, cur_s, cur_e = 0, None, None
    for s, e in iv:
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                busy += cur_e - cur_s
            cur_s, cur_e = s, e
        elif e > cur_e:
            cur_e = e
    if cur_e is not None:
        busy += cur_e - cur_s
    # steady state: the same sweep over the middle of the run only (from the 30th to the 95th
    # percentile of the library's own kernel launches -- before that the lanes are still being
    # allocated and warmed, after it the pipeline drains)
    lib = db.execute("select start, end from kernels where name like '%k\\_%' escape '\\' order by start").fetchall()
    if len(lib) > 100:
        t0, t1 = lib[int(0.30 * len(lib))][0], lib[int(0.95 * len(lib))][0]
        b2, cs, ce, tot2 = 0, None, None, 0
        for s_, e_ in iv:
            s2, e2 = max(s_, t0), min(e_, t1)
            if e2 <= s2:
                continue
            tot2 += e2 - s2
            if ce is None or s2 > ce:
                if ce is not None:
                    b2 += ce - cs
                cs, ce = s2, e2
            elif e2 > ce:
                ce = e2
        if ce is not None:
            b2 += ce - cs
        print("\nsteady state (%.1f ms window inside the run): some kernel running during %.1f %% of it, "
              "%.2f kernels in flight on average" % ((t1 - t0) / 1e6, 100.0 * b2 / (t1 - t0), tot2 / (t1 - t0)))
    span = db.execute("select min(start), max(end) from kernels").fetchone()
    if span and span[0] is not None:
        print("\nkernel time %.3f ms over a %.3f ms span (sum/span = %.2f: >1 means kernels from "
              "different lanes overlap); some kernel running during %.1f %% of the span"
              % (total / 1e6, (span[1] - span[0]) / 1e6, total / max(span[1] - span[0], 1),
                 100.0 * busy / max(span[1] - span[0], 1)))


if __name__ == "__main__":
    main(sys.argv[1])
