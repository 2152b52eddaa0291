#!/usr/bin/env python3
"""Device-wide counters of an OVERLAPPED run (bench.py --devcount, tools/devcount.cpp) -> one stamped summary.

    python tools/devcount_summary.py gpurun_out/devcount_issue.json gpurun_out/devcount_wait.json [more ...] \
        > profiles/rNN_devcount.json

Each input is a bench line with a `devcount` object (one counter set per run, collected over the timed region of an
ordinary pipelined run: nothing is serialised).  Derived (MI355X: 256 CUs x 4 SIMDs, 8 XCDs):
  cycles           GRBM_GUI_ACTIVE summed over the 8 XCD instances / 8     (device-active cycles of the region)
  clock_GHz        cycles / region seconds
  valu_active      SQ_ACTIVE_INST_VALU * 4 / (1024 * cycles)   MEASURED share of SIMD-cycles spent executing VALU
  valu_model       SQ_INSTS_VALU * 4 / (1024 * cycles)         the 4-cycles-per-wave64-instruction model
  cyc_per_valu     SQ_ACTIVE_INST_VALU * 4 / SQ_INSTS_VALU
  valu_per_pair    SQ_INSTS_VALU / pairs
  waves_per_simd   SQ_WAVE_CYCLES * 4 / (1024 * cycles)
  parked / stalled SQ_WAIT_ANY / SQ_WAVE_CYCLES, SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (same run for both)
  lds_busy         SQ_LDS_IDX_ACTIVE / (256 * cycles);  lds_conflict_share = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  salu_busy        SQ_INST_CYCLES_SALU * 4 / (1024 * cycles)
  hbm              FETCH_SIZE / WRITE_SIZE (KiB units of the derived counters; read = 2 * FETCH_SIZE * 1024 on gfx950 as
                   in tools/pmc_summary.py), per pair and as a rate over the region
"""
import json
import sys


def main(paths):
    out = {"build": None, "runs": [], "what": "rocprofiler-sdk device counting service over the timed region of bench.py "
                                              "(pipelined steady state, kernels overlap as in every run)"}
    C = {}
    for p in paths:
        try:
            d = json.loads(open(p).read().strip().splitlines()[-1])
        except (OSError, ValueError, IndexError):
            continue
        dc = d.get("devcount")
        if not dc:
            continue
        build = d["config"]["build"]
        if out["build"] is None:
            out["build"] = build
        if build != out["build"]:
            raise SystemExit("devcount_summary: %s was taken on another build" % p)
        c = dc["counters"]
        cyc = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
        run = {"file": p.split("/")[-1], "pairs_per_s": d["value"], "pairs": dc["pairs"], "region_s": dc["region_s"],
               "cycles": cyc, "clock_GHz": cyc / dc["region_s"] / 1e9 if dc["region_s"] else None, "counters": c}
        out["runs"].append(run)
        for k, v in c.items():
            C.setdefault(k, (v, cyc, dc["pairs"], dc["region_s"]))
    def get(k):
        return C.get(k, (None, None, None, None))
    der = {}
    v, cyc, pairs, sec = get("SQ_INSTS_VALU")
    if v:
        der["valu_model"] = round(v * 4 / (1024 * cyc), 4)
        der["valu_per_pair"] = round(v / pairs)
        a = get("SQ_ACTIVE_INST_VALU")[0]
        if a:
            der["valu_active"] = round(a * 4 / (1024 * cyc), 4)
            der["cyc_per_valu"] = round(a * 4 / v, 3)
    w, cyc, _, _ = get("SQ_WAVE_CYCLES")
    if w:
        der["waves_per_simd"] = round(w * 4 / (1024 * cyc), 3)
    b, cyc, _, _ = get("SQ_BUSY_CYCLES")
    if b:
        der["sq_busy"] = round(b / (32 * cyc), 4)
    l, cyc, _, _ = get("SQ_ACTIVE_INST_LDS")
    if l:
        der["lds_inst_active"] = round(l * 4 / (1024 * cyc), 4)
    x, cyc, _, _ = get("SQ_LDS_IDX_ACTIVE")
    if x:
        der["lds_busy"] = round(x / (256 * cyc), 4)
        bc = get("SQ_LDS_BANK_CONFLICT")[0]
        if bc is not None:
            der["lds_conflict_share"] = round(bc / x, 4)
    s, cyc, _, _ = get("SQ_INST_CYCLES_SALU")
    if s:
        der["salu_busy"] = round(s * 4 / (1024 * cyc), 4)
    # parked / stalled need SQ_WAVE_CYCLES of THEIR run: take the ratio against the occupancy of the issue run
    wa = get("SQ_WAIT_ANY")[0]
    wi = get("SQ_WAIT_INST_ANY")[0]
    if wa and w:
        der["parked"] = round(wa / w, 4)
    if wi and w:
        der["stalled"] = round(wi / w, 4)
    f, _, pairs_f, sec_f = get("FETCH_SIZE")
    wr, _, pairs_w, sec_w = get("WRITE_SIZE")
    if f is not None and wr is not None and pairs_f and pairs_w:
        rd_b, wr_b = 2.0 * f * 1024.0, wr * 1024.0
        der["hbm_bytes_per_pair"] = round(rd_b / pairs_f + wr_b / pairs_w)
        der["hbm_GBps"] = round((rd_b / sec_f + wr_b / sec_w) / 1e9, 1)
        der["hbm_frac_of_8TBps"] = round(der["hbm_GBps"] / 8000.0, 4)
    out["derived"] = der
    out["reading"] = ("valu_active is measured busy time of the vector ALUs under overlap; 1 - valu_active is what no "
                      "wave was ready to fill (parked = share of resident-wave time waiting on memory / LDS / barriers)")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main(sys.argv[1:])
