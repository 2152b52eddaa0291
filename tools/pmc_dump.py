#!/usr/bin/env python3
"""Print the per-kernel average of every counter in one or more rocprofv3 --pmc rocpd
databases (one line per kernel; counters summed over their instances per dispatch).

    python tools/pmc_dump.py pass1.db [pass2.db ...] [--only k_match,k_support]
"""
import re
import sqlite3
import sys


def main(argv):
    only = None
    paths = []
    for a in argv:
        if a.startswith("--only"):
            only = set(a.split("=", 1)[1].split(","))
        else:
            paths.append(a)
    table = {}
    for p in paths:
        db = sqlite3.connect(p)
        q = ("select name, counter_name, count(distinct dispatch_id), sum(counter_value) "
             "from pmc_events group by name, counter_name")
        for name, ctr, n, tot in db.execute(q):
            m = re.search(r"(k_\w+)", name)
            if not m:
                continue
            k = m.group(1)
            if only and k not in only and k.replace("_lds", "") not in only:
                continue
            table.setdefault(k, {})[ctr] = tot / max(n, 1)
    for k in sorted(table):
        print(k)
        for c in sorted(table[k]):
            print("   %-28s %16.1f" % (c, table[k][c]))


if __name__ == "__main__":
    main(sys.argv[1:])
