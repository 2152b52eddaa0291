#!/usr/bin/env python3
"""Per-kernel HBM traffic from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE are
collected in SEPARATE runs: they do not fit one pass on gfx950, MI355X_MICROARCH.md
"rocprofv3 PMC slots").  Units: both counters are KiB per dispatch.  gfx950 correction
(same guide, "HBM"): FETCH_SIZE tallies 128-byte read requests at 64 bytes, i.e. reports
half the bytes of a wide coalesced stream -> read bytes = 2 * FETCH_SIZE * 1024 for the
16-byte-per-lane kernels here; byte-wide image loads (k_descriptor, k_filters) are not
covered by that calibration and are flagged.  WRITE_SIZE is taken as is.

    python tools/pmc_summary.py FETCH.db WRITE.db pairs_per_launch > profiles/rNN_pmc_traffic.json
"""
import json
import re
import os
import sqlite3
import sys


def build_stamp():
    """svh_version() of the libsvhip.so these counters were taken on (carries the source hash):
    bench.py quotes the file only while it matches the library it has loaded"""
    import ctypes
    so = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "stereo-vision_amd", "libsvhip.so")
    lib = ctypes.CDLL(so)
    lib.svh_version.restype = ctypes.c_char_p
    return lib.svh_version().decode()


ALIAS = {"k_support_lds": "k_support", "k_match_keyed": "k_match", "k_match_list": "k_match"}   # symbol -> bench.py profile name


def per_kernel(path, counter):
    """average per launch over the launches with the largest grid of each kernel (the full
    G-pair groups; bench.py's single-pair latency calls are excluded)"""
    db = sqlite3.connect(path)
    grid = {did: gx * gy * gz for did, gx, gy, gz in
            db.execute("select dispatch_id, grid_x, grid_y, grid_z from kernels")}
    rows = {}
    for name, did, tot in db.execute(
            "select name, dispatch_id, sum(counter_value) from pmc_events where counter_name=? "
            "group by name, dispatch_id", (counter,)):
        m = re.search(r"(k_\w+)", name)
        if m:
            k = ALIAS.get(m.group(1), m.group(1))
            rows.setdefault(k, []).append((grid.get(did, 0), tot))
    out = {}
    for k, v in rows.items():
        gmax = max(g for g, _ in v)
        sel = [x for g, x in v if g == gmax]
        out[k] = (len(sel), sum(sel) / len(sel))
    return out


def main(fetch_db, write_db, pairs):
    f = per_kernel(fetch_db, "FETCH_SIZE")
    w = per_kernel(write_db, "WRITE_SIZE")
    res = {"build": build_stamp(), "pairs_per_launch": int(pairs), "unit": "bytes per launch",
           "note": "read = 2*FETCH_SIZE*1024 (gfx950 128-B requests tallied at 64 B), write = WRITE_SIZE*1024",
           "kernels": {}}
    for k in sorted(set(f) | set(w)):
        fk = f.get(k, (0, 0.0))
        wk = w.get(k, (0, 0.0))
        res["kernels"][k] = {
            "launches": fk[0], "fetch_size_kib": round(fk[1], 1), "write_size_kib": round(wk[1], 1),
            "read_bytes": int(2 * fk[1] * 1024), "write_bytes": int(wk[1] * 1024),
            "hbm_bytes": int(2 * fk[1] * 1024 + wk[1] * 1024),
            "read_calibrated": k not in ("k_descriptor", "k_filters"),
        }
    json.dump(res, sys.stdout, indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:4])
