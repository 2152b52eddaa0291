#!/usr/bin/env python3
"""The round's measurement table of DESIGN.md section 7, regenerated from profiles/<round>_*:
    python tools/design_numbers.py r06 > /tmp/table.md     (the text between the markers in DESIGN.md)"""
import json
import os
import sys

R = sys.argv[1] if len(sys.argv) > 1 else "r06"
P = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def L(name):
    try:
        txt = open(os.path.join(P, "%s_%s.json" % (R, name))).read().strip()
        return json.loads([l for l in txt.splitlines() if l.startswith("{")][-1])
    except (OSError, ValueError, IndexError):
        return None


def v(d, *keys, default=None):
    for k in keys:
        if d is None:
            return default
        d = d.get(k) if isinstance(d, dict) else None
    return default if d is None else d


b = L("bench_line")
rf = b["roofline"]
dc = json.load(open(os.path.join(P, R + "_devcount.json")))["derived"]
dh = json.load(open(os.path.join(P, R + "_devcount_hd1080.json")))["derived"]
rows = []
rows.append(("`value`", "**%d pairs/s** (%.1f ms per step of 6 144 pairs, `outputs_match_golden: %s`, build `%s`); soak line %d, RCCL branch with one rank %d, stream API %d, C++ driver `svh_shard` with its RCCL communicator and no environment variable %d; round 3's matcher on this build (`SVH_MATCH_LIST=0`) %d"
             % (b["value"], b["ms_per_step"], str(b.get("outputs_match_golden")).lower(), b["config"]["build"].split()[-1],
                v(L("bench_line_soak60"), "value", default=0), v(L("bench_line_1rank_nccl"), "value", default=0),
                v(L("bench_line_kitti_stream"), "value", default=0), v(L("shard_driver_1rank_rccl"), "value", default=0),
                v(L("bench_line_keyed_matcher"), "value", default=0))))
cb = b["cpu_baseline"]
rows.append(("`cpu_baseline`", "reference `Elas::process` (`oracle/_ref`): **%.1f pairs/s** on one %s core (%.1f ms/pair); %d processes %.0f pairs/s"
             % (cb["value"], cb["host"].replace("AMD ", "").replace(" 64-Core Processor", ""), cb["ms_per_pair"],
                v(cb, "nproc_workers", "workers", default=0), v(cb, "nproc_workers", "value", default=0))))
rows.append(("`roofline` (dominant kernel `%s`)" % rf["kernel"],
             "isolated 32-pair launch **%.1f us** (HIP events, one group alone on the device); `frac` = **%.3f** of the 8 TB/s peak on the design's %.0f MB per launch (`achieved` %.0f GB/s), `traffic` %s MB from the counters (`frac_on_measured_traffic` %s); `bound` = `%s`; isolated kernels of a group sum to %.0f us"
             % (rf["avg_launch_us"], rf["frac"], rf["alg_bytes_per_launch"] / 1e6, rf["achieved"],
                ("%.0f" % (rf["traffic"] / 1e6)) if rf.get("traffic") else "n/a", rf.get("frac_on_measured_traffic"), rf["bound"], rf.get("isolated_sum_us", 0))))
rows.append(("`roofline.overlapped_counters` (`%s_devcount.json`: device-wide counters over the timed region, nothing serialised)" % R,
             "VALU active **%.3f** of all SIMD-cycles (%.2f M instructions per pair at %.2f cycles), %.2f waves per SIMD, parked %.2f, stalled %.2f, LDS busy %.2f (conflict share %.2f), SALU %.2f, HBM **%.0f GB/s = %.2f of the peak** (%.1f MB per pair)"
             % (dc["valu_active"], dc["valu_per_pair"] / 1e6, dc["cyc_per_valu"], dc["waves_per_simd"], dc["parked"], dc["stalled"], dc["lds_busy"],
                dc["lds_conflict_share"], dc["salu_busy"], dc["hbm_GBps"], dc["hbm_frac_of_8TBps"], dc["hbm_bytes_per_pair"] / 1e6)))
hd = L("bench_line_hd1080")
rows.append(("`--workload hd1080` (configs[3])", "**%d pairs/s** through the stream, 16 pairs per launch, `outputs_match_oracle: %s`; 8 per step %d, 256 per step %d, batch entry %d; device-wide: some wave resident %.1f %% of the time, VALU active %.2f, %.2f waves per SIMD, parked %.2f, HBM %.2f of the peak; isolated kernels of a 16-pair group sum to %.0f us"
             % (hd["value"], str(hd.get("outputs_match_oracle")).lower(), v(L("bench_line_hd1080_b8"), "value", default=0), v(L("bench_line_hd1080_x256"), "value", default=0),
                v(L("bench_line_hd1080_batch_api"), "value", default=0), 100 * dh["sq_busy"], dh["valu_active"], dh["waves_per_simd"], dh["parked"], dh["hbm_frac_of_8TBps"],
                v(hd, "roofline", "isolated_sum_us", default=0))))
rows.append(("`--workload sequence` (configs[2] substitute)", "**%d pairs/s** through the stream, %d through the batch entry"
             % (v(L("bench_line_sequence"), "value", default=0), v(L("bench_line_sequence_batch_api"), "value", default=0))))
hb, hs = b.get("throughput_host_buffers", {}), b.get("throughput_host_buffers_stream", {})
rows.append(("host buffers in and out (PCIe-inclusive, never `value`)", "batch entry **%d pairs/s** (%.1f GB/s over PCIe); as a stream (`svh_elas_stream_push_n` / `pop_n`, order preserved) **%d pairs/s** = %.2f of it (%d for one 430-frame sequence on its own, ramp-up and drain included), maps == device path: %s; single `svh_elas_process` %.2f ms"
             % (hb.get("value", 0), hb.get("pcie_GBps", 0), hs.get("value", 0), hs.get("vs_batch_entry", 0), v(hs, "per_sequence_of_430", "value", default=0), str(hs.get("maps_equal_device_path")).lower(), b.get("latency_ms_single_pair_host_buffers", 0))))
m, vo = b.get("matcher", {}), b.get("visual_odometry", {})
tl = m.get("timeline", {})
runs = {(r["calling_threads"], r["objects_per_call"]): r for r in v(vo, "lockstep", "runs", default=[]) if r.get("pipelined")}
rows.append(("Matcher / VO legs", "Matcher %.2f ms per stereo frame (reference %.1f), `matcher_matches_reference: %s`; timeline: device %.3f ms = %.2f of the call, host-only steps %.3f ms (dense vote %.3f), waiting for the device %.3f; VO %.2f ms (reference %.1f), `vo_matches_reference: %s`; lockstep pipelined 1 x 16 %d, 2 x 16 %d frames/s at %.1f host cores, 4 x 8 %d"
             % (m.get("frame_ms", 0), v(m, "cpu_reference", "frame_ms", default=0), str(m.get("matcher_matches_reference")).lower(), tl.get("device_ms", 0),
                tl.get("device_busy_fraction_of_the_call", 0), tl.get("host_only_ms", 0), v(tl, "steps_ms", "matchFeatures: dense outlier vote (host)", default=0),
                tl.get("host_waiting_for_the_device_ms", 0), vo.get("frame_ms", 0), v(vo, "cpu_reference", "frame_ms", default=0), str(vo.get("vo_matches_reference")).lower(),
                v(runs.get((1, 16)), "frames_per_s", default=0), v(runs.get((2, 16)), "frames_per_s", default=0), v(runs.get((2, 16)), "host_cores_used", default=0),
                v(runs.get((4, 8)), "frames_per_s", default=0))))
rows.append(("multi-rank dry runs on the one GPU", "2 ranks gloo %d; 8 ranks gloo: kitti %d, sequence %d, 1920x1080 %d; `svh_shard` 4 ranks %d, 8 ranks on the 430-frame sequence (strong) %d; `placement` / `one_rank_per_device` reported in every line"
             % (v(L("bench_line_2ranks_gloo_1gpu"), "value", default=0), v(L("bench_line_8ranks_gloo_1gpu_kitti"), "value", default=0),
                v(L("bench_line_8ranks_gloo_1gpu_sequence"), "value", default=0), v(L("bench_line_8ranks_gloo_1gpu_hd1080"), "value", default=0),
                v(L("shard_driver_4ranks_pipes_1gpu"), "value", default=0), v(L("shard_driver_8ranks_sequence_strong"), "value", default=0))))
print("| | |\n|---|---|")
for a, c in rows:
    print("| %s | %s |" % (a, c))
