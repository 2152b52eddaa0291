#!/usr/bin/env python3
"""Issue-pipe utilisation per kernel from two rocprofv3 --pmc rocpd databases (SQ counters;
tools/gpu_pmc.sh collects them in separate passes).  Only the launches with the largest grid
of each kernel are used (the full G-pair groups; single-pair latency calls are excluded).

    python tools/pmc_issue.py pass_a.db pass_b.db > profiles/rNN_pmc_issue.json

Derived per launch (MI355X: 256 CUs, 4 SIMDs per CU, 8 XCDs):
  cycles            GRBM_GUI_ACTIVE summed over the 8 XCD instances / 8
  valu_busy         SQ_INSTS_VALU * 4 / (1024 * cycles)   (MODEL: a wave64 VALU op of the classes these
                    kernels are made of holds its SIMD for 4 cycles -- profiles/r03_microbench_valu.txt:
                    v_sad_u8 / min / max / VOP3 / DPP / compares 4.2-4.5, add / and / or / xor / fp32 fma 2.4-2.6)
  valu_active       SQ_ACTIVE_INST_VALU * 4 / (1024 * cycles)   (MEASURED: quad-cycles waves spent
                    executing VALU instructions, over the SIMD-cycles of the launch)
  cyc_per_valu      SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU     (measured cycles per wave-level VALU op)
  sq_busy           SQ_BUSY_CYCLES / (8 XCDs x cycles) as reported (any wave resident)
  salu_cycles       SQ_INST_CYCLES_SALU * 4 / (1024 * cycles)
  salu_per_valu     SQ_INSTS_SALU / SQ_INSTS_VALU
  lds_busy          SQ_LDS_IDX_ACTIVE / (256 * cycles)
  waves_per_simd    SQ_WAVE_CYCLES * 4 / (1024 * cycles)  (SQ_WAVE_CYCLES counts quad-cycles)
  parked / stalled  SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES
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


def load(path, table):
    db = sqlite3.connect(path)
    grid = {}
    for name, did, gx, gy, gz in db.execute("select name, dispatch_id, grid_x, grid_y, grid_z from kernels"):
        grid[did] = (name, gx * gy * gz)
    q = ("select dispatch_id, counter_name, sum(counter_value), max(duration) from pmc_events "
         "group by dispatch_id, counter_name")
    for did, ctr, tot, dur in db.execute(q):
        name, g = grid.get(did, (None, 0))
        m = re.search(r"(k_\w+)", name or "")
        if not m:
            continue
        table.setdefault(m.group(1), {}).setdefault(ctr, []).append((g, tot, dur))


def main(paths):
    table = {}
    for p in paths:
        load(p, table)
    out = {"build": build_stamp(), "note": __doc__.split("Derived per launch")[1].strip().splitlines()[0], "kernels": {}}
    for k in sorted(table):
        c = {}
        dur = 0.0
        for ctr, rows in table[k].items():
            gmax = max(r[0] for r in rows)
            sel = [r for r in rows if r[0] == gmax]
            c[ctr] = sum(r[1] for r in sel) / len(sel)
            dur = max(dur, sum(r[2] for r in sel) / len(sel))
        if "GRBM_GUI_ACTIVE" not in c or "SQ_INSTS_VALU" not in c:
            continue
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        wc = c.get("SQ_WAVE_CYCLES", 0.0)
        out["kernels"][k] = {
            "launch_us_under_pmc": round(dur / 1e3, 2),
            "cycles": round(cyc),
            "waves": round(c.get("SQ_WAVES", 0)),
            "valu_wave_instr": round(c["SQ_INSTS_VALU"]),
            "salu_wave_instr": round(c.get("SQ_INSTS_SALU", 0)),
            "lds_wave_instr": round(c.get("SQ_INSTS_LDS", 0)),
            "valu_busy": round(c["SQ_INSTS_VALU"] * 4.0 / (1024.0 * cyc), 3),
            "valu_active": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / (1024.0 * cyc), 3) if "SQ_ACTIVE_INST_VALU" in c else None,
            "cyc_per_valu": round(c["SQ_ACTIVE_INST_VALU"] * 4.0 / max(c["SQ_INSTS_VALU"], 1), 2) if "SQ_ACTIVE_INST_VALU" in c else None,
            "sq_busy_cycles": round(c["SQ_BUSY_CYCLES"]) if "SQ_BUSY_CYCLES" in c else None,
            "salu_cycles": round(c["SQ_INST_CYCLES_SALU"] * 4.0 / (1024.0 * cyc), 3) if "SQ_INST_CYCLES_SALU" in c else None,
            "any_active": round(c["SQ_ACTIVE_INST_ANY"] / wc, 3) if (wc and "SQ_ACTIVE_INST_ANY" in c) else None,
            "salu_per_valu": round(c.get("SQ_INSTS_SALU", 0) / max(c["SQ_INSTS_VALU"], 1), 3),
            "lds_busy": round(c.get("SQ_LDS_IDX_ACTIVE", 0) / (256.0 * cyc), 3),
            "lds_bank_conflict_cycles": round(c.get("SQ_LDS_BANK_CONFLICT", 0)),
            "waves_per_simd": round(wc * 4.0 / (1024.0 * cyc), 2) if wc else None,
            "parked": round(c.get("SQ_WAIT_ANY", 0) / wc, 3) if wc else None,
            "stalled": round(c.get("SQ_WAIT_INST_ANY", 0) / wc, 3) if wc else None,
        }
    json.dump(out, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main(sys.argv[1:])
