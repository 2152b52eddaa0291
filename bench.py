#!/usr/bin/env python3
"""Headline benchmark: ELAS stereo pairs/s on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--batch B] [--lanes L]

Workload (BASELINE.json configs[1]): the four KITTI-size 1242x375 crops of
the reference's own urban{1..4} images (tests/golden/, offset x=51 y=8), tiled; full ELAS ROBOTICS
parameters, D1+D2 with L/R check, subsampling off.  A "step" is one pass of the hot path over one
batch of B pairs that are already resident in HBM; disparity maps are written to HBM.

Multi-GPU: pairs are independent, so ranks shard them with no data-path collective ("weak"
scaling: B pairs per rank and step); only a small per-rank record is gathered over RCCL.  The
driver starts one process per GPU (torch.distributed.run).  From a plain shell `--gpus N` with N>1
and no WORLD_SIZE in the environment makes this script start the N ranks itself (same launcher,
127.0.0.1 rendezvous).  `--dist-backend gloo` lets several ranks share one GPU, which is how the
N>1 path is exercised on a 1-GPU box.

Besides the contract fields the JSON line carries
  roofline                 achieved algorithmic GB/s of the dominant kernel, from HIP events
                           recorded on the kernels' own streams (svh_profile_*), vs 8 TB/s
  cpu_baseline             the reference Elas::process (oracle/_ref) on this host: 1 thread, and
                           `nproc_workers` = one independent process per available core
  throughput_host_buffers  the reference's ownership contract (host pointers in and out,
                           SURVEY 8b) through svh_elas_process_batch on pinned buffers --
                           PCIe-inclusive, never `value`
  value_synthetic          the same step on seeded synthetic pairs (round-1 headline workload)
  ranks                    per rank: pairs/s, host cores used, workers, host-core ceiling
torch is used only for device memory, pinned memory, the barrier and the result gather.
"""
import argparse
import ctypes as C
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))

W, H = 1242, 375        # --workload kitti (BASELINE.json configs[1]); hd1080 rebinds these
N_PIX = W * H
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
ALG_BYTES_PER_PIXEL_PAIR = 174.8   # staged model, whole Elas::process, per pixel
URBAN = ["urban%d_1242x375" % i for i in (1, 2, 3, 4)]

# algorithmic bytes per pair and kernel launch (SURVEY 8d staged model, N = W*H)
ALG_BYTES_PER_PIXEL = {
    "k_descriptor": 34.0,   # 2 x (1 B in + 16 B out)
    "k_support": 12.8,      # rows v+-2 of a 5-row lattice, both images
    "k_match": 72.0,        # 2 x (16 own + 16 other + 4 out)
    "k_lr": 16.0,
    "k_seg_tile": 8.0, "k_seg_border": 8.0, "k_seg_sum": 8.0, "k_seg_mask": 8.0,
    "k_gap_rows": 8.0, "k_gap_cols": 8.0, "k_gap_tile": 16.0,     # tile kernels: rows + columns in one
    "k_mean_h": 8.0, "k_mean_v": 8.0, "k_mean_tile": 16.0,
    "k_owner": 8.0, "k_owner_fix": 8.0,
}


# Compulsory HBM bytes per pixel of the kernels AS DESIGNED since round 4 (descriptors assembled on the fly from
# the two Sobel planes: nothing ever stores the 16-byte descriptors): what a launch must move even with perfect
# caching.  `roofline.achieved / frac` are quoted on THESE bytes -- a kernel cannot move more than it has to -- and the
# SURVEY 8(d) staged-model figure (bytes a stage-by-stage pipeline with stored descriptors would move) keeps its own
# key, `staged_model_8d`.
#   k_match       4 N  du + dv planes of both images (each row is fetched by the blocks of 5 + 3 neighbouring image
#                      rows: L2 traffic, not compulsory) + 8 N owner words of both maps + 8 N D1 + D2 after the fused
#                      L/R check + 0.6 N candidate records and triangle planes
#   k_support     4 N  the planes; the 37 KB lattice out is noise
#   k_descriptor  6 N  (k_sobel_planes) 2 N image bytes in, 4 N plane bytes out
#   the post-filters move what the staged model says (their inputs and outputs are the maps themselves)
ALG_BYTES_DESIGN_PER_PIXEL = dict(ALG_BYTES_PER_PIXEL, k_match=20.6, k_support=4.0, k_descriptor=6.0)


def make_inputs(batch, seed0=1000):
    import helpers as Hh
    I1 = np.empty((batch, H, W), np.uint8)
    I2 = np.empty((batch, H, W), np.uint8)
    for i in range(batch):
        I1[i], I2[i] = Hh.synth_pair(W, H, seed0 + i, dmax=96 if W < 1500 else 200, planes=8)
    return I1, I2


def urban_inputs():
    """the four KITTI-size crops of the reference's urban images (committed under tests/golden)"""
    import helpers as Hh
    pairs = [Hh.golden_pair(n) for n in URBAN]
    return np.stack([p[0] for p in pairs]), np.stack([p[1] for p in pairs])


def _cpu_run_fn(params, I1, I2):
    """(kind, run(i)) for the CPU leg: the reference itself when oracle/_ref is present"""
    import helpers as Hh
    D1 = np.zeros((H, W), np.float32)
    D2 = np.zeros((H, W), np.float32)
    dims = (C.c_int32 * 3)(W, H, W)
    if Hh.have_ref_elas():
        lib = C.CDLL(Hh.ref_elas_path())   # no ref_init(1): plain allocator, fair timing

        def run(i):
            lib.ref_elas_process(C.byref(params), I1[i].ctypes.data_as(C.c_void_p),
                                 I2[i].ctypes.data_as(C.c_void_p), D1.ctypes.data_as(C.c_void_p),
                                 D2.ctypes.data_as(C.c_void_p), dims)
        return "reference", run
    import svhip as S
    lib = Hh.oracle()
    tri = C.cast(S.lib().svh_delaunay, C.c_void_p)   # timing leg only: Triangle is not restated

    def run(i):
        lib.orc_elas_process(C.byref(params), I1[i].ctypes.data_as(C.c_void_p),
                             I2[i].ctypes.data_as(C.c_void_p), D1.ctypes.data_as(C.c_void_p),
                             D2.ctypes.data_as(C.c_void_p), dims, tri)
    return "port", run


def _cpu_loop(run, n_unique, budget_s, cap=400):
    run(0)  # warm
    n_done, t_used, i = 0, 0.0, 0
    while t_used < budget_s and n_done < cap:
        t = time.perf_counter()
        run(i % n_unique)
        t_used += time.perf_counter() - t
        n_done += 1
        i += 1
    return n_done, t_used


def cpu_worker(budget_s):
    """hidden mode (--cpu-worker S): one of the `nproc` independent CPU workers; no torch"""
    import helpers as Hh
    I1, I2 = urban_inputs()
    kind, run = _cpu_run_fn(Hh.robotics(), I1, I2)
    n, t = _cpu_loop(run, len(I1), budget_s)
    print(json.dumps({"n": n, "t": t, "kind": kind}), flush=True)


def cpu_baseline(I1, I2, params, what, budget_s=5.0, workers=0):
    """reference (or port) on host cores on a bounded sample of the bench's own pairs: 1 thread,
    and -- SURVEY 8(d): the only parallelism the reference admits -- one independent worker
    process per available core, one pair each"""
    kind, run = _cpu_run_fn(params, I1, I2)
    n_done, t_used = _cpu_loop(run, len(I1), budget_s)
    out = {"value": n_done / t_used, "unit": "pairs/s", "cores": 1, "kind": kind,
           "ms_per_pair": 1e3 * t_used / n_done,
           "sample": "%d x Elas::process on %s, 1 thread, %s"
                     % (n_done, what, "oracle/_ref (reference compiled -O3 -msse3)" if kind == "reference"
                        else "oracle/ scalar port"),
           "host": _cpu_model(), "host_cores": os.cpu_count(), "host_cpu_quota": _cpu_quota()}
    if workers > 1:
        t0 = time.perf_counter()
        procs = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--cpu-worker", str(budget_s)],
                                  stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
                 for _ in range(workers)]
        recs = []
        for p in procs:
            o, _ = p.communicate(timeout=60 + 20 * budget_s)
            try:
                recs.append(json.loads(o.strip().splitlines()[-1]))
            except (ValueError, IndexError):
                pass
        if recs:
            # every worker loops for the same budget; throughput = pairs finished / slowest loop
            tot = sum(r["n"] for r in recs)
            out["nproc_workers"] = {
                "value": tot / max(r["t"] for r in recs), "unit": "pairs/s", "workers": len(recs),
                "cores": len(recs), "pairs": tot, "wall_s": round(time.perf_counter() - t0, 2),
                "sample": "%d independent processes (one per available core), each Elas::process on the four "
                          "urban crops for %.0f s" % (len(recs), budget_s)}
    return out


def matcher_bench(iters=40):
    """secondary metric (BASELINE.json configs[4]): libviso2 Matcher on the reference's quad
    (1344x391): ms per stereo frame for pushBack + matchFeatures(2), device vs reference CPU"""
    import helpers as Hh
    im = [Hh.read_pgm(os.path.join(Hh.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    prm = Hh.matcher_defaults()

    def run(m, n):
        m.push_back(im[0], im[1])
        t_push = t_match = 0.0
        nm = 0
        for i in range(n):
            a, b = (im[2], im[3]) if i % 2 == 0 else (im[0], im[1])
            t = time.perf_counter()
            m.push_back(a, b)
            t_push += time.perf_counter() - t
            t = time.perf_counter()
            m.match(2, staged=False)
            t_match += time.perf_counter() - t
            nm = len(m.matches())
        return 1e3 * t_push / n, 1e3 * t_match / n, nm

    def helper_stats():
        st = (C.c_int64 * 4)()
        try:
            dev.lib.svh_host_helper_stats(st)
        except AttributeError:   # (an older library under SVH_LIB)
            pass
        return list(st)

    dev = Hh.ProductMatcher(prm)
    dev.lib.svh_matcher_set_taps(C.c_void_p(dev.h), 0)   # timing: no intermediate stage copies
    run(dev, 10)
    hs0 = helper_stats()
    cpu0, wall0 = time.process_time(), time.perf_counter()
    push, match, nm = run(dev, iters)
    host_cores = (time.process_time() - cpu0) / (time.perf_counter() - wall0)   # caller + polling helper threads
    hs1 = helper_stats()
    out = {"workload": "quad match on libviso2/img I1p/I2p/I1c/I2c 1344x391, default parameters",
           "pushBack_ms": push, "matchFeatures_ms": match, "frame_ms": push + match,
           "frames_per_s": 1e3 / (push + match), "matches": nm, "host_cores_used": round(host_cores, 2),
           "host_helper_threads": dict(zip(("tasks", "l3_moves", "taken_by_a_polling_helper", "taken_by_a_sleeping_helper"),
                                           [b - a for a, b in zip(hs0, hs1)]), per_frames=iters)}
    # round 6: the call's timeline from the library's own clocks (svh_matcher_get_timing: host wall-clock steps, and
    # the device time of the three device phases from HIP events on the object's streams), a second pass with the
    # collection on; and the leg's roofline: SURVEY 8(d)'s 17 N bytes per stereo frame over the device time
    try:
        dev.lib.svh_matcher_set_timing(1)
        dev.lib.svh_matcher_get_timing.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]
        names, ms = (C.c_char_p * 10)(), (C.c_double * 10)()
        dev.lib.svh_matcher_get_timing(C.c_void_p(dev.h), names, ms, 10, 1)
        tpush, tmatch, _ = run(dev, iters)
        k = dev.lib.svh_matcher_get_timing(C.c_void_p(dev.h), names, ms, 10, 1)
        dev.lib.svh_matcher_set_timing(0)
        tl = {names[i].decode(): round(ms[i], 4) for i in range(k)}
        host = [ms[i] for i in range(7)]
        dev_ms = ms[7] + ms[8] + ms[9]
        # (the frame the fractions refer to is the one measured above with the clocks OFF: reading eight HIP events per
        # frame costs this process -- torch's runtime, 20 hardware queues, dozens of streams -- up to a millisecond that
        # is no part of a frame; the steps themselves are timed inside the library and do not contain it)
        frame = push + match
        n_px = im[0].shape[0] * im[0].shape[1]
        out["timeline"] = {
            "frame_ms": round(frame, 4), "frame_ms_with_the_clocks_on": round(tpush + tmatch, 4), "steps_ms": tl,
            "device_ms": round(dev_ms, 4), "device_busy_fraction_of_the_call": round(dev_ms / frame, 3),
            "host_only_ms": round(host[0] + host[3] + host[4] + host[6], 4),
            "host_waiting_for_the_device_ms": round(host[1] + host[2] + host[5], 4),
            "outside_the_library_ms": round(frame - sum(host), 4),
            "note": "one object, one frame after the other (Matcher::pushBack then matchFeatures(2), as viso_stereo.cpp:41-68 "
                    "calls them): the dense outlier vote (a Delaunay triangulation of ~2.9 k matches, matcher.cpp:1383-1570) "
                    "and the prior statistics are host steps of the reference that SURVEY 8(a) leaves on the host; nothing "
                    "of the NEXT frame exists while they run (the caller hands it over afterwards), so for one object they "
                    "are idle time of the device (the triangulation runs on eight threads of the host: helper threads on the "
                    "caller's L3, awake while the dense matching is on the device).  K objects in lockstep fill it: "
                    "visual_odometry.lockstep"}
        out["roofline"] = {
            "bound": "latency", "unit": "GB/s", "algorithmic_bytes_per_frame": 17 * n_px,
            "achieved": round(17 * n_px / (dev_ms * 1e-3) / 1e9, 1), "peak": 8000.0,
            "frac": round(17 * n_px / (dev_ms * 1e-3) / 1e9 / 8000.0, 4),
            "device_ms_per_frame": round(dev_ms, 4),
            "note": "SURVEY 8(d): 17 N bytes per stereo frame (2 images x (4 N full-resolution Sobel + half image + 18 N/4)) "
                    "over the summed device time of the call's three device phases (HIP events); a frame is ~25 short "
                    "launches on one stream -- 8.9 MB cannot fill the machine, the phases are launch- and "
                    "latency-bound, and the call as a whole is bound by its host steps (timeline)"}
    except Exception as ex:      # (an older library under SVH_LIB)
        out["timeline"] = {"error": str(ex)}
    if Hh.have_ref_viso():
        ref = Hh.RefMatcher(prm)
        ref.lib.ref_init(0)
        rp, rm, rn = run(ref, 12)
        out["cpu_reference"] = {"pushBack_ms": rp, "matchFeatures_ms": rm, "frame_ms": rp + rm,
                                "matches": rn, "cores": 1, "kind": "reference"}
        # self-check of this leg (after the timing): the same three frames through a fresh object on each side,
        # the match lists -- indices i1p, i2p, i1c, i2c and coordinates -- byte for byte
        def lists(m):
            seq = []
            m.push_back(im[0], im[1])
            for a, b in ((im[2], im[3]), (im[0], im[1])):
                m.push_back(a, b)
                m.match(2, staged=False)
                seq.append(m.matches().tobytes())
            return seq
        d2 = Hh.ProductMatcher(prm)
        d2.lib.svh_matcher_set_taps(C.c_void_p(d2.h), 0)
        r2 = Hh.RefMatcher(prm)
        r2.lib.ref_init(0)
        out["matcher_matches_reference"] = lists(d2) == lists(r2)
        out["self_check"] = "match lists of two quad matches (fresh objects, same frames) == the reference's Matcher " \
                            "(oracle/_ref), byte for byte: indices and coordinates"
    return out


def vo_bench(iters=40):
    """VisualOdometryStereo::process (pushBack + quad match + bucketing +
    RANSAC/Gauss-Newton motion estimate) per stereo frame on the reference's quad, device vs the
    reference on one host core; plus estimateMotion alone on the bucketed matches"""
    import helpers as Hh
    im = [Hh.read_pgm(os.path.join(Hh.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    prm = Hh.vo_defaults()

    def run(vo, n):
        vo.process(im[0], im[1])
        t = time.perf_counter()
        ok = 0
        for i in range(n):
            a, b = (im[2], im[3]) if i % 2 == 0 else (im[0], im[1])
            ok += vo.process(a, b) == 1
        frame = 1e3 * (time.perf_counter() - t) / n
        m = vo.matches()
        t = time.perf_counter()
        for _ in range(n):
            vo.estimate_motion(m)
        est = 1e3 * (time.perf_counter() - t) / n
        return frame, est, ok, len(m), len(vo.inliers())

    dev = Hh.ProductVo(prm)
    t_warm = time.perf_counter()
    while time.perf_counter() - t_warm < 0.4:   # the CPU reference leg before this one let the GPU clock down
        run(dev, 20)
    del dev     # (one live object: the library waits for its stream by spinning, the single-sequence mode)
    frame, est, ok, nm, ni = run(Hh.ProductVo(prm), iters)
    out = {"workload": "VisualOdometryStereo::process on libviso2/img quad 1344x391, default parameters, "
                       "calibration of demo.cpp", "frame_ms": frame, "estimateMotion_ms": est,
           "frames_ok": ok, "matches": nm, "inliers": ni}
    if Hh.have_ref_viso():
        rf, re_, rok, rnm, rni = run(Hh.RefVo(prm), 10)
        out["cpu_reference"] = {"frame_ms": rf, "estimateMotion_ms": re_, "frames_ok": rok, "matches": rnm,
                                "inliers": rni, "cores": 1, "kind": "reference",
                                "note": "timed over 10 frames (the device leg over %d): the counts above are those of "
                                        "each side's LAST frame at a different position of the libc rand() stream; the "
                                        "like-for-like comparison is vo_matches_reference" % iters}
        # Self-check of this leg (after the timing): the SAME six frames through a fresh object on each side.  Both
        # constructors call srand(0) (viso.cpp:36), so bucketing and RANSAC draw the same numbers: bucketed matches
        # byte for byte, inlier indices identical, return values equal, motion within 1e-9.
        def seq(cls):
            vo = cls(prm)
            res = []
            for i in range(6):
                a, b = (im[0], im[1]) if i % 2 == 0 else (im[2], im[3])
                ok = vo.process(a, b)
                res.append((ok, vo.matches().tobytes(), vo.inliers().tolist(), vo.motion().copy()))
            return res
        sd, sr = seq(Hh.ProductVo), seq(Hh.RefVo)
        same = all(x[0] == y[0] and x[1] == y[1] and x[2] == y[2] and float(np.abs(x[3] - y[3]).max()) < 1e-9
                   for x, y in zip(sd, sr))
        out["vo_matches_reference"] = bool(same)
        out["self_check"] = {"frames": 6, "inliers_last_frame": len(sd[-1][2]), "inliers_last_frame_reference": len(sr[-1][2]),
                             "what": "return value, bucketed matches (bytes), inlier indices and Tr (1e-9) of six frames, "
                                     "fresh objects on both sides (srand(0) in both constructors), == the reference's "
                                     "VisualOdometryStereo (oracle/_ref)"}
    return out


def vo_replicas_bench(ks=(1, 4, 16), frames=40):
    """the Matcher / visual odometry chain frames of ONE sequence (ring buffer, Tr_delta), so
    it does not shard; what scales is the number of independent sequences.  K VisualOdometryStereo
    objects (own streams, own ring buffers), one host thread each, all on this GPU: aggregate stereo
    frames/s.  (libc rand() is shared by the threads, as it would be for K reference objects in one
    process: the RANSAC sample streams interleave, throughput is what is measured here.)"""
    import threading
    import helpers as Hh
    im = [Hh.read_pgm(os.path.join(Hh.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    prm = Hh.vo_defaults()
    out = []
    for K in ks:
        vos = [Hh.ProductVo(prm) for _ in range(K)]
        for vo in vos:                      # first frames: allocations, bootstrap
            vo.process(im[0], im[1])
            vo.process(im[2], im[3])
        ok = [0] * K
        go = threading.Barrier(K + 1)

        def worker(j):
            vo = vos[j]
            go.wait()
            for i in range(frames):
                a, b = (im[0], im[1]) if i % 2 == 0 else (im[2], im[3])
                ok[j] += vo.process(a, b) == 1
        th = [threading.Thread(target=worker, args=(j,)) for j in range(K)]
        for t in th:
            t.start()
        import resource
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        out.append({"replicas": K, "frames_per_s": K * frames / dt, "frame_ms_per_replica": 1e3 * dt / frames,
                    "frames_ok": int(sum(ok)), "frames": K * frames, "host_cores_used": round(cpu / dt, 2)})
        del vos
    return {"workload": "K independent VisualOdometryStereo objects on one GPU, libviso2/img quad 1344x391 "
                        "alternating, one host thread per object", "runs": out}


def vo_lockstep_bench(shapes=((1, 4), (1, 16), (1, 32), (2, 8), (2, 16), (4, 8)), frames=40, private_rand=True,
                      pipelined=(False, True)):
    """the same sequences driven in LOCKSTEP: svh_vo_process_batch, one launch per kernel over the K objects of a
    call (blockIdx.z = object), host steps (row packing, outlier votes, prior statistics) on the library's
    helper threads.  shapes = (host threads, objects per call): with two or more calling threads one group's
    host steps overlap the other's device steps.  pipelined: svh_vo_prefetch_batch + svh_vo_process_next_batch, the
    next frame's packing / upload / feature extraction overlap this frame's matching and motion estimate.  private_rand: every object draws from its own generator
    (glibc's srand(0) sequence) instead of the process-wide rand(), whose lock the calling threads contend for
    and whose draw order forces the bucketing of a call's objects to run one after the other.  Aggregate stereo frames/s; per object the results equal
    svh_vo_process calls (tests/test_batch_gpu.py)."""
    import resource
    import threading
    import helpers as Hh
    im = [Hh.read_pgm(os.path.join(Hh.GOLDEN, "viso_%s.pgm" % k)) for k in ("I1p", "I2p", "I1c", "I2c")]
    prm = Hh.vo_defaults()
    out = []
    for T, K, pipe in [(T, K, p) for p in pipelined for (T, K) in shapes]:
        groups = []
        for g in range(T):
            vos = [Hh.ProductVo(prm, private_rand=0 if private_rand else None) for _ in range(K)]
            # object k sees the quad shifted by 3k columns: different sequences
            seq = [[np.roll(a, 3 * (g * K + k), axis=1) for a in im] for k in range(K)]
            even = ([s[0] for s in seq], [s[1] for s in seq])
            odd = ([s[2] for s in seq], [s[3] for s in seq])
            for i in range(4):              # allocations, bootstrap (run one by one inside the entry), warm-up
                Hh.product_vo_process_batch(vos, *(even if i % 2 == 0 else odd))
            groups.append((vos, even, odd))
        good = [0] * T
        go = threading.Barrier(T + 1)

        errs = []

        def worker(g):
            try:
                work(g)
            except Exception as e:   # noqa: BLE001  (a failed call must not look like a fast one)
                errs.append(repr(e))

        def work(g):
            vos, even, odd = groups[g]
            go.wait()
            if pipe:   # frame i + 1 is handed over while frame i is matched
                shape = even[0][0].shape
                Hh.product_vo_prefetch_batch(vos, *even)
                for i in range(frames):
                    nxt = (odd if i % 2 == 0 else even) if i + 1 < frames else (None, None)
                    n, _ = Hh.product_vo_process_next_batch(vos, nxt[0], nxt[1], shape)
                    good[g] += n
                return
            for i in range(frames):
                n, _ = Hh.product_vo_process_batch(vos, *(even if i % 2 == 0 else odd))
                good[g] += n
        th = [threading.Thread(target=worker, args=(g,)) for g in range(T)]
        for t in th:
            t.start()
        ru0 = resource.getrusage(resource.RUSAGE_SELF)
        go.wait()
        t0 = time.perf_counter()
        for t in th:
            t.join()
        dt = time.perf_counter() - t0
        ru1 = resource.getrusage(resource.RUSAGE_SELF)
        cpu = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
        if errs:
            out.append({"calling_threads": T, "objects_per_call": K, "pipelined": pipe, "frames_per_s": None,
                        "error": errs[0]})
            del groups
            continue
        out.append({"calling_threads": T, "objects_per_call": K, "pipelined": pipe, "frames_per_s": T * K * frames / dt,
                    "call_ms": 1e3 * dt / frames, "frames_ok": int(sum(good)), "frames": T * K * frames,
                    "host_cores_used": round(cpu / dt, 2)})
        del groups
    return {"workload": "VisualOdometryStereo objects in lockstep (svh_vo_process_batch), libviso2/img quad "
                        "1344x391 alternating, shifted 3k columns for object k",
            "random_numbers": "private per-object streams with glibc's srand(0) sequence (svh_vo_set_private_rand)"
                              if private_rand else "libc rand(), drawn in object order", "runs": out}


def vo_replicas_processes(procs=4, per_proc=4, frames=60):
    """the same K = procs x per_proc sequences as `procs` PROCESSES with `per_proc` objects each (every
    process has its own HIP runtime: what serialises K objects of one process is the runtime's launch
    path, ~45 launches and copies per frame)"""
    script = os.path.join(ROOT, "tools", "gpu_legs.py")
    # (the runtime's default of 4 hardware queues per process: this bench's own 16 times several
    # processes oversubscribes the device's queue slots -- 1.9 k instead of 4.8 k frames/s)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="4")
    ps = [subprocess.Popen([sys.executable, script, "replicas%d" % per_proc], stdout=subprocess.PIPE,
                           stderr=subprocess.DEVNULL, text=True, env=env) for _ in range(procs)]
    tot, cores, ok = 0.0, 0.0, 0
    for pr in ps:
        out, _ = pr.communicate(timeout=600)
        try:
            r = json.loads(out.strip().splitlines()[-1])["vo_replicas"]["runs"][0]
            tot += r["frames_per_s"]
            cores += r["host_cores_used"]
            ok += 1
        except (ValueError, KeyError, IndexError):
            pass
    return {"processes": procs, "replicas_per_process": per_proc, "processes_ok": ok, "frames_per_s": tot,
            "host_cores_used": round(cores, 2),
            "note": "sum of the processes' own rates (they start together and run the same number of frames)"}


def map_bench(iters=30):
    """3-D reprojection + map fusion of one 1242x375 frame (createCurrentMap +
    addDisparityMapToReconstruction), device vs the CPU restatement (oracle, "port": the reference's
    own code sits in Qt classes that cannot be built here)"""
    import test_map as TM
    from svhip import mapper
    import helpers as Hh
    (f, cu, cv, base), frames = TM.synth_frames(W, H, 6, seed=3)

    def run(mk, n):
        m = mk()
        d, img, Ht, g = frames[0]
        m.add(d, img, Ht, g)
        t = time.perf_counter()
        for i in range(n):
            d, img, Ht, g = frames[1 + i % 5]
            m.add(d, img, Ht, g)
        ms = 1e3 * (time.perf_counter() - t) / n
        return ms, len(m.points(0)), len(m.points(1))

    run(lambda: mapper.Mapper(f, cu, cv, base, 20), 3)
    ms, n0, n1 = run(lambda: mapper.Mapper(f, cu, cv, base, 20), iters)
    out = {"workload": "map fusion of synthetic %dx%d frames (host disparity map in, point lists on the device)" % (W, H),
           "frame_ms": ms, "points_prev": n0, "points_curr": n1}
    L = TM.oracle_map(Hh.oracle())
    cms, c0, c1 = run(lambda: TM.OracleMapper(L, TM.MapParams(f, cu, cv, base, 20)), 5)
    out["cpu_port"] = {"frame_ms": cms, "cores": 1, "kind": "port",
                       "note": "oracle/map_oracle.cpp, 5 frames (the device leg runs %d; parity is tests/test_map.py)" % iters}
    return out


def settings_bench(S, torch, dev, batch=6144, steps=3, only=None):
    """pairs/s for the parameter sets the APPLICATION uses, next to the headline's plain ROBOTICS preset:
    stereomapper/stereothread.cpp:76-114 (ROBOTICS + support_texture = 30, postprocess_only_left, adaptive mean),
    the same with the GUI's subsampling checkbox (maindialog.cpp:473 -> param.subsampling), and the MIDDLEBURY
    preset (libelas/src/main.cpp, elas.h:118-146) on the reference's `cones` pair.  Device-resident batches through
    the same entry as the headline, steps of the headline's size (a call returns when its last pair is done: with
    1 536 pairs per call the ramp and the drain of the six workers cost 7 %); the first maps of each leg are compared
    with the reference's own output (tests/golden where a golden exists, oracle/_ref otherwise)."""
    import helpers as Hh
    legs = []
    u1, u2 = urban_inputs()
    cl, cr = Hh.golden_pair("cones_640x480")
    cones = (cl[None].repeat(4, 0), cr[None].repeat(4, 0))
    for name, prm, (a1, a2), gold in (
            ("stereomapper (ROBOTICS, support_texture 30)", Hh.robotics(support_texture=30), (u1, u2), (1, "urban2_stereomapper")),
            ("stereomapper + subsampling", Hh.robotics(support_texture=30, subsampling=1), (u1, u2), None),
            ("MIDDLEBURY on cones 640x480", Hh.middlebury(), cones, (0, "cones_middlebury"))):
        if only is not None and only.lower() not in name.lower():
            continue
        h, w = a1.shape[1:]
        dh, dw = (h // 2, w // 2) if prm.subsampling else (h, w)
        idx = torch.arange(batch, device=dev) % len(a1)
        dI1 = torch.from_numpy(np.ascontiguousarray(a1)).to(dev)[idx].contiguous()
        dI2 = torch.from_numpy(np.ascontiguousarray(a2)).to(dev)[idx].contiguous()
        dD1 = torch.empty((batch, dh, dw), dtype=torch.float32, device=dev)
        dD2 = torch.empty((batch, dh, dw), dtype=torch.float32, device=dev)
        e = S.Elas(prm)

        def step():
            st = e.process_batch_device(batch, dI1.data_ptr(), dI2.data_ptr(), w * h, dD1.data_ptr(), dD2.data_ptr(),
                                        dw * dh * 4, w, h, w)
            assert all(x == 0 for x in st), [x for x in st if x][:4]
        for _ in range(3):      # untimed: lane pool of the new shape, and the clocks the host-bound legs before let down
            step()
        torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        # outputs against the reference
        if gold is not None:
            k, npz = gold
            z = np.load(os.path.join(Hh.GOLDEN, npz + ".npz"))
            same = bool(np.array_equal(dD1[k].cpu().numpy().ravel(), z["d1"]) and
                        np.array_equal(dD2[k].cpu().numpy().ravel()[: z["d2"].size], z["d2"]))
            how = "tests/golden/%s.npz" % npz
        elif Hh.have_ref_elas():
            R1, R2 = Hh.ref_elas_process(prm, a1[1], a2[1])
            same = bool(np.array_equal(dD1[1].cpu().numpy(), R1) and np.array_equal(dD2[1].cpu().numpy(), R2))
            how = "oracle/_ref Elas::process on the same pair"
        else:
            same, how = None, "no checker on this box"
        legs.append({"setting": name, "image": "%dx%d" % (w, h), "pairs_per_s": steps * batch / dt,
                     "pixels_per_s": steps * batch * w * h / dt, "pairs_per_step": batch,
                     "outputs_match_reference": same, "checked_against": how,
                     "d1_valid_fraction": round(float((dD1[:16] >= 0).float().mean().item()), 4)})
        del dI1, dI2, dD1, dD2, e
    return legs


def _cpu_quota():
    """CPU cores this process may use (cgroup v2 cpu.max), None when unlimited / unknown"""
    try:
        q, per = open("/sys/fs/cgroup/cpu.max").read().split()
        return None if q == "max" else round(int(q) / int(per), 1)
    except (OSError, ValueError):
        return None


def _cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def read_profile(S):
    lib = S.lib()
    lib.svh_profile_get.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double),
                                    C.POINTER(C.c_int64)]
    n = lib.svh_profile_get(-1, None, None, None)
    out = {}
    for i in range(n):
        name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
        lib.svh_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt))
        out[name.value.decode()] = (ms.value, cnt.value)
    return out


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def self_spawn(argv, n):
    """`python bench.py --gpus N` from a plain shell: start the N ranks with the same launcher the
    driver uses (one process per GPU, rendezvous on 127.0.0.1)"""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver (RCCL)
    return subprocess.call(cmd, env=env)


def load_pmc(build, suffix=""):
    """the committed rocprofv3 --pmc summaries, only if they were taken on the library that is
    loaded now (tools/pmc_*.py stamp them with svh_version(), which carries the source hash)"""
    import glob
    out = {"traffic": None, "issue": None, "devcount": None, "notes": []}
    # (suffix "_hd1080": the passes of tools/gpu_pmc_all.sh _hd1080 --workload hd1080)
    for key, pat in (("traffic", "*_pmc_traffic%s.json" % suffix), ("issue", "*_pmc_issue%s.json" % suffix),
                     ("devcount", "*_devcount.json")):
        files = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
        if not files:
            continue
        try:
            d = json.load(open(files[-1]))
        except (OSError, ValueError):
            continue
        if d.get("build") == build:
            out[key] = d
            out[key + "_file"] = os.path.basename(files[-1])
        else:
            out["notes"].append("%s was measured on build '%s', the loaded library is '%s': not quoted"
                                % (os.path.basename(files[-1]), d.get("build", "unstamped"), build))
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", choices=("kitti", "hd1080", "sequence"), default="kitti",
                    help="kitti = BASELINE.json configs[1] (the headline: 1242x375); hd1080 = "
                         "configs[3] / SURVEY 8(d) config 4 (synthetic 1920x1080, disp_max 255); "
                         "sequence = configs[2]: a 430-frame 1242x375 sequence streamed once per step, "
                         "frames sharded contiguously over the GPUs (strong scaling).  drive_0029 is "
                         "not available offline: the frames cycle the four committed KITTI-size crops "
                         "of the reference's urban images (SURVEY 8d config 3 substitute)")
    ap.add_argument("--data", choices=("urban", "synthetic"), default="urban",
                    help="kitti workload: the four urban crops tiled (default) or seeded synthetic pairs")
    ap.add_argument("--batch", type=int, default=0,
                    help="pairs per step and GPU (0 = 6144 for kitti: a step of ~0.2 s, so that the "
                         "driver's 20 steps run for seconds; hd1080: configs[3]'s batch of 64 pairs shared "
                         "by the GPUs -- 64 on one, 8 each on eight)")
    ap.add_argument("--unique", type=int, default=64,
                    help="different synthetic pairs generated per rank; the batch tiles them")
    ap.add_argument("--lanes", type=int, default=0,
                    help="double-buffered pipeline workers per GPU (0 = auto: 1.5 per available core, <= 24)")
    ap.add_argument("--group", type=int, default=0,
                    help="pairs per kernel launch, 1..32 (0 = 32 for kitti; hd1080: 16 for steps of 32 pairs "
                         "or more, else 1 on the host stage and 2 on the device stage)")
    ap.add_argument("--spinup", type=float, default=1.0,
                    help="seconds of untimed steps before the warmup (GPU clocks, lane buffers)")
    ap.add_argument("--profile-in-timed-region", type=int, default=1,
                    help="1: HIP-event kernel timing is on during the timed steps (roofline comes "
                         "from exactly those launches); 0: a separate pass after them")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the default) or gloo (CPU collectives: lets several ranks "
                         "share one GPU when the multi-rank path is exercised on a 1-GPU box)")
    ap.add_argument("--soak", type=int, default=0,
                    help="after the timed region: this many extra steps, the output maps zero-filled before "
                         "each and EVERY pair of the batch verified after it (urban crops only; a race / "
                         "stale-buffer detector, reported as golden_check.soak)")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (and run the record gather and the barriers) even "
                         "with one rank: one execution of the RCCL path on a 1-GPU box")
    ap.add_argument("--stage", choices=("auto", "host", "device"), default="auto",
                    help="where lattice filters + Delaunay run (svh_elas_set_stage): auto = device for batches")
    ap.add_argument("--api", choices=("auto", "batch", "stream"), default="auto",
                    help="batch: one svh_elas_process_batch_device call per step (drains at its end); stream: "
                         "svh_elas_stream_* -- a producer thread pushes the steps' pairs, this thread pops "
                         "them, the lanes stay full across steps.  auto = stream for the workloads whose "
                         "step is shorter than the pipeline is deep (sequence, hd1080), batch for kitti")
    ap.add_argument("--depth", type=int, default=0, help="stream: pairs in flight (0: lanes x 2 groups)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-bind", action="store_true", help="several ranks: do not bind a rank's threads to its GPU's NUMA node")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the secondary legs (synthetic, host buffers, latency, Matcher, VO, map)")
    ap.add_argument("--cpu-budget", type=float, default=5.0, help="seconds per CPU baseline leg")
    ap.add_argument("--cpu-worker", type=float, default=0.0, help=argparse.SUPPRESS)
    ap.add_argument("--devcount", default="",
                    help="comma-separated hardware counters collected DEVICE-WIDE over the timed region by "
                         "tools/libdevcount.so (rocprofiler-sdk device counting service: the kernels are NOT "
                         "serialised, unlike rocprofv3 --pmc); one set per run; result under `devcount`")
    ap.add_argument("--kitti-dir", default="",
                    help="--workload sequence on a real KITTI raw drive directory (image_00/, image_01/ "
                         "with data/ and timestamps.txt) instead of the 430-frame substitute")
    args = ap.parse_args()
    if args.cpu_worker > 0:
        return cpu_worker(args.cpu_worker)
    # dmabuf IPC only on this driver: RCCL / device tensors shared across processes need it, whoever
    # started the ranks (self_spawn below, the driver's torch.distributed.run, a plain shell)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(self_spawn(sys.argv[1:], args.gpus))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks" % (args.gpus, world))
    global W, H, N_PIX
    if args.workload == "hd1080":
        W, H = 1920, 1080
        N_PIX = W * H
        args.data = "synthetic"
    kitti_frames = None
    if args.workload == "sequence" and args.kitti_dir:
        pass   # the drive is read below, after torch has brought up the HIP runtime
    elif args.workload == "sequence":
        from svhip import shard as _sh
        lo_, hi_ = _sh.shard_range(430, rank, world)
        args.batch = hi_ - lo_
        args.seq_first = lo_
    if args.batch <= 0:
        # hd1080 = BASELINE.json configs[3]: "batch=64 sharded 8 per GPU" -- the batch of 64 pairs is the
        # workload, one GPU takes all of it, eight take 8 each
        args.batch = 6144 if args.workload == "kitti" else max(8, 64 // max(world, 1))
    if args.group <= 0:
        # hd1080: 8 pairs per step, one pair per lane; kitti: the stages between the matching phases
        # (k_lattice, k_delaunay) are latency-bound single-workgroup jobs -- 32 pairs share one launch
        # (hd1080 on the device stage: pairs of a group share the launches of its latency-bound stage
        # kernels -- 8 pairs per step as 4 lanes x 2 pairs: 2.2 k pairs/s against 1.7 k as 8 x 1)
        if args.workload != "hd1080":
            args.group = 32
        elif (args.batch >= 32 or args.api in ("auto", "stream")) and args.stage != "host":
            # a deep step / a stream: the library's automatic mode takes the device stage.  16 pairs per launch
            # (round 5 sweep on one box: 8 -> 6.55 k, 16 -> 7.14-7.28 k, 24 -> 7.27 k pairs/s; 32 with 6 workers
            # collapses to 4.5 k: 12 lanes x 32 pairs x 2 MP of buffers no longer overlap anything)
            args.group = 16
        else:
            args.group = 2 if args.stage == "device" else 1

    # The ROCm runtime multiplexes HIP streams onto GPU_MAX_HW_QUEUES hardware queues (default 4)
    # and kernels of one hardware queue run one after the other.  The 6 double-buffered workers use 12
    # streams, and two of a group's kernels (k_delaunay ~0.8 ms, k_lattice ~0.2 ms: one workgroup per
    # triangulation / pair) occupy their queue while using almost none of the machine: with 16 queues
    # every stream has its own (round 3: 28.2 k pairs/s at 8, 29.6 k at 16, 29.3-29.6 k at 24-32; round 2
    # measured 4 -> 8 at +4 %).  Must be set before the runtime starts; an explicit setting of the caller wins.
    # Round 6: libsvhip no longer edits the environment when it is loaded; a program asks for the count through
    # svh_init (include/svh.h) BEFORE anything starts the HIP runtime -- the C++ callers (apps/svh_shard.cpp,
    # tests/cxx/elas_dropin.cpp via the implicit initialisation) do exactly that.  This script cannot: the library must be
    # loaded AFTER torch (torch ships its own copy of the HIP runtime, and the copy that is loaded first serves both), so
    # it does what svh_init documents for such hosts -- it sets GPU_MAX_HW_QUEUES itself, as the host program, before the
    # runtime starts; `library_init` in the line then reports "caller_set".
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")
    devcount_so = os.path.join(ROOT, "tools", "libdevcount.so")
    if args.devcount:
        if not os.path.exists(devcount_so):
            raise SystemExit("bench.py --devcount: build tools/libdevcount.so first (make -C tools)")
        os.environ["ROCP_TOOL_LIBRARIES"] = devcount_so     # read when the HIP runtime starts
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    ndev = torch.cuda.device_count()
    if world > ndev and args.dist_backend == "nccl":
        raise SystemExit("bench.py: %d ranks but %d GPU(s): RCCL needs one GPU per rank "
                         "(--dist-backend gloo lets ranks share a GPU)" % (world, ndev))
    device_index = local_rank % ndev
    # preflight (SCALE readiness): who shares a device, and how many hardware queues the ranks on it ask for.  The
    # ROCm runtime gives every process GPU_MAX_HW_QUEUES queues on a device; ranks that share one (gloo dry runs on a
    # 1-GPU box) multiply that.  RCCL with two ranks on one device is refused above.
    ranks_on_my_device = len([r for r in range(world) if r % ndev == device_index]) if world > 1 else 1
    hwq = int(os.environ.get("GPU_MAX_HW_QUEUES", "4"))
    preflight = {"ranks": world, "gpus_visible": ndev, "ranks_on_this_device": ranks_on_my_device,
                 "hw_queues_per_rank": hwq, "hw_queues_requested_on_this_device": hwq * ranks_on_my_device}
    if rank == 0 and world > 1:
        print("bench.py preflight: %d ranks on %d visible GPU(s), backend %s: rank 0 -> device %d shared by %d rank(s); "
              "GPU_MAX_HW_QUEUES=%d per rank = %d hardware queues requested on that device"
              % (world, ndev, args.dist_backend, device_index, ranks_on_my_device, hwq, hwq * ranks_on_my_device),
              file=sys.stderr, flush=True)
    torch.cuda.set_device(device_index)
    dev = torch.device("cuda", device_index)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where collectives run
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:
            if world > 1:
                raise SystemExit("bench.py: WORLD_SIZE>1 without MASTER_PORT (use --gpus N from a plain shell, "
                                 "or torch.distributed.run)")
            os.environ["MASTER_PORT"] = str(_free_port())
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import svhip as S
    import helpers as Hh
    library_init = S.init()          # (the explicit form of what the first svh_* call would do; the queues: see above)
    S.lib().svh_set_device(device_index)
    # 8-GPU readiness (round 6): with several ranks, each one binds its threads (the engine's workers inherit the mask) to
    # the CPUs of its GPU's NUMA node, at most its share of the host cores; the PCI address travels in the rank's record
    # and rank 0 checks "one rank per device, distinct addresses" whenever there are enough devices
    topo = S.device_topology(device_index)
    cpus_bound = 0
    if world > 1 and not args.no_bind:
        share = max(1, int(len(os.sched_getaffinity(0)) // max(1, min(world, 8))))
        cpus_bound = S.bind_host_to_device(device_index, share)
    S.set_stage({"auto": -1, "host": 0, "device": 1}[args.stage])
    build = S.lib().svh_version().decode()
    if args.workload == "sequence" and args.kitti_dir:
        from svhip import kitti as _kitti
        k1, k2, lo_, total_ = _kitti.load_shard(args.kitti_dir, rank, world)
        kitti_frames = (k1, k2, total_)
        args.batch = k1.shape[0]
        args.seq_first = lo_
        H, W = k1.shape[1:]
        N_PIX = W * H
    # workers per GPU: each is double-buffered and sleeps while it waits, so ~1.5 per available
    # core keeps the cores busy with the host stage
    avail = _cpu_quota() or (os.cpu_count() or 8)
    cores_per_rank = avail / max(world, 1)
    # (with the device stage the workers only enqueue and sleep: 6 of them, 12 streams, already
    # saturate the device; more only stretches every kernel's in-run duration)
    api = args.api if args.api != "auto" else ("batch" if args.workload == "kitti" else "stream")
    # (a stream always asks for the device stage where the geometry allows it: its workers only enqueue and
    # sleep; so does a rank with fewer than 4 cores to itself -- the host stage of 1920x1080 pairs needs 5-7:
    # the first 8-rank run on a 16-core quota must not be a host-starvation artefact)
    dev_stage = args.stage != "host" and (args.workload != "hd1080" or args.stage == "device" or args.batch >= 32
                                          or api == "stream" or cores_per_rank < 4)
    if dev_stage and args.stage == "auto" and args.workload == "hd1080" and api == "batch" and args.batch < 32:
        S.set_stage(1)     # the library's automatic mode would take the host stage for a shallow batch of large images
        args.stage = "device (auto: %.1f cores per rank)" % cores_per_rank
    # device stage: 6 workers whatever the core count (they use ~0.03 cores each; 8 ranks on a 16-core
    # quota keep the depth the single-GPU number was measured with); host stage: by cores
    lanes = args.lanes or (6 if dev_stage else int(max(2, min(24, round(1.5 * cores_per_rank)))))
    S.set_lanes(lanes)
    group = S.set_group(args.group)

    B = args.batch
    params = Hh.robotics()           # Elas::parameters(ROBOTICS), elas.h:91-116

    def tile_to_device(a1, a2, first=0):
        """B pairs = the unique pairs cycled (each copy at its own HBM address; the library keeps
        nothing between pairs, so a repeated pair is full work)"""
        idx = torch.tensor([(first + k) % len(a1) for k in range(B)], device=dev)
        return (torch.from_numpy(a1).to(dev)[idx].contiguous(),
                torch.from_numpy(a2).to(dev)[idx].contiguous())

    if kitti_frames is not None:
        U = B
        dI1 = torch.from_numpy(kitti_frames[0]).to(dev).contiguous()
        dI2 = torch.from_numpy(kitti_frames[1]).to(dev).contiguous()
        I1, I2 = kitti_frames[0], kitti_frames[1]
        data = "KITTI raw drive " + os.path.basename(os.path.normpath(args.kitti_dir))
        what = "the drive's frames"
    elif args.workload == "sequence" or (args.workload == "kitti" and args.data == "urban"):
        # frame f = crop f % 4; in the sequence workload this rank owns [seq_first, seq_first + B)
        U = 4
        I1, I2 = urban_inputs()
        dI1, dI2 = tile_to_device(I1, I2, getattr(args, "seq_first", 0))
        data = "urban crops"
        what = "the four 1242x375 urban crops"
    else:
        U = min(B, args.unique)
        I1, I2 = make_inputs(U, seed0=1000 + 100000 * rank)
        dI1, dI2 = tile_to_device(I1, I2)
        data = "synthetic"
        what = "the bench's own %dx%d synthetic pairs" % (W, H)
    depth = args.depth or lanes * 2 * group
    # stream: several steps are in flight at once, each writes its own slice of a ring of output buffers
    nring = 1 if api == "batch" else (depth + B - 1) // B + 1
    dDr1 = torch.empty((nring * B, H, W), dtype=torch.float32, device=dev)
    dDr2 = torch.empty((nring * B, H, W), dtype=torch.float32, device=dev)
    dD1, dD2 = dDr1[:B], dDr2[:B]
    torch.cuda.synchronize()
    e = S.Elas(params)
    # (opened after the spin-up and the probe step: a stream holds its workers' lanes while it is open,
    # the batch entry would wait for them)
    stm_box = [None]
    stream_steps = [0]

    def run_stream(k):
        """k steps through the stream: a producer thread pushes k x B pairs, this thread pops them"""
        import threading
        if stm_box[0] is None:
            stm_box[0] = e.stream(W, H, W, depth)
        stm = stm_box[0]
        base = stream_steps[0]

        def produce():
            for j in range(k):
                sl = ((base + j) % nring) * B
                stm.push_device_n(B, dI1.data_ptr(), dI2.data_ptr(), W * H, dDr1[sl].data_ptr(),
                                  dDr2[sl].data_ptr(), W * H * 4)
            stm.flush()
        th = threading.Thread(target=produce)
        th.start()
        for j in range(k):
            st = stm.pop_n(B)
            assert len(st) == B and all(x == 0 for x in st), ("stream statuses", [x for x in st if x][:8], S.last_error())
        th.join()
        stream_steps[0] = base + k

    def step():
        st = e.process_batch_device(B, dI1.data_ptr(), dI2.data_ptr(), W * H, dD1.data_ptr(),
                                    dD2.data_ptr(), W * H * 4, W, H, W)
        assert all(s == 0 for s in st), ("non-zero statuses", [(i, s) for i, s in enumerate(st) if s != 0][:8],
                                         S.last_error())

    def barrier():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    t_spin = time.perf_counter()
    step()
    while time.perf_counter() - t_spin < args.spinup:   # untimed: clocks up, lanes allocated
        step()
    if api != "stream":
        for _ in range(args.warmup):
            step()
    # which kernel dominates?  one profiled, untimed step with every kernel bracketed
    in_region = bool(args.profile_in_timed_region) and rank == 0
    prof_all = {}
    if rank == 0:
        S.lib().svh_profile_only.argtypes = [C.c_char_p]
        S.lib().svh_profile_only(None)
        S.lib().svh_profile_reset()
        S.lib().svh_profile_enable(1)
        step()
        S.lib().svh_profile_enable(0)
        prof_all = read_profile(S)
    if in_region and prof_all:
        # timed region: HIP events only around the dominant kernel (2 records per group)
        # (among the kernels that fill the device: see the roofline block below)
        fill0 = [k for k in prof_all if k in ALG_BYTES_DESIGN_PER_PIXEL] or list(prof_all)   # (see the roofline block)
        dom0 = max(fill0, key=lambda k: prof_all[k][0])
        S.lib().svh_profile_only(dom0.encode())
        S.lib().svh_profile_reset()
        S.lib().svh_profile_enable(1)
    if api == "stream":
        run_stream(max(1, args.warmup))      # opens the stream; its workers and lanes stay up from here on
    barrier()
    import resource
    dc = dc_names = dc_first = None
    if args.devcount and rank == 0:
        dc = C.CDLL(devcount_so)
        dc.svh_devcount_start.argtypes = [C.c_char_p]
        dc.svh_devcount_sample.argtypes = [C.POINTER(C.c_double), C.c_int]
        dc_names = [x for x in args.devcount.split(",") if x]
        rc = dc.svh_devcount_start(",".join(dc_names).encode())
        if rc != 0:
            raise SystemExit("bench.py --devcount: svh_devcount_start -> %d" % rc)
        buf = (C.c_double * len(dc_names))()
        dc.svh_devcount_sample(buf, len(dc_names))
        dc_first = list(buf)
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    if api == "stream":
        run_stream(args.steps)       # returns when the last pair of the last step has been popped
        sl = ((stream_steps[0] - 1) % nring) * B
        dD1, dD2 = dDr1[sl:sl + B], dDr2[sl:sl + B]     # the maps of the last step (self-check below)
    else:
        for _ in range(args.steps):
            step()
    torch.cuda.synchronize()
    elapsed_local = time.perf_counter() - t0
    devcount = None
    if dc is not None:
        buf = (C.c_double * len(dc_names))()
        rc = dc.svh_devcount_sample(buf, len(dc_names))
        dc.svh_devcount_stop()
        devcount = {"counters": {n: buf[i] - dc_first[i] for i, n in enumerate(dc_names)},
                    "first_sample": {n: dc_first[i] for i, n in enumerate(dc_names)},
                    "region_s": elapsed_local, "pairs": B * args.steps, "rc": rc,
                    "how": "rocprofiler-sdk device counting service on the whole device (tools/devcount.cpp), started "
                           "before and sampled after the timed region; the workers' kernels overlap as in every run "
                           "(nothing is serialised); values are sums over the counter's instances"}
    if stm_box[0] is not None:
        stm_box[0].close()
        stm_box[0] = None
    barrier()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    cpu_s = (ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)
    host_cores_used = cpu_s / elapsed
    if in_region:
        S.lib().svh_profile_enable(0)
        S.lib().svh_profile_only(None)
    # the only collective on the path: a small per-rank result record (RCCL over xGMI, or gloo)
    from svhip import shard
    my_pairs = B * args.steps
    # hd1080 has no committed goldens (synthetic pairs): every rank compares the maps of its first two pairs,
    # as the last timed step left them, with the reference's Elas::process run here on the same inputs
    # (oracle/_ref, the checker -- after the timed region), every pixel
    oracle_bad = -1.0
    if args.workload == "hd1080" and data == "synthetic" and Hh.have_ref_elas():
        oracle_bad = 0.0
        for k in sorted(set((0, min(1, B - 1)))):
            R1, R2 = Hh.ref_elas_process(params, I1[k % U], I2[k % U])
            oracle_bad += float((dD1[k].cpu().numpy() != R1).sum() + (dD2[k].cpu().numpy() != R2).sum())
    rec = [float(my_pairs), float((dD1[0] >= 0).sum().item()), host_cores_used, float(lanes),
           elapsed_local, cpu_s / my_pairs, float(device_index), oracle_bad,
           float(shard.pack_bus_id(topo["pci_bus_id"])), float(topo["numa_node"]), float(cpus_bound)]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        recs = shard.gather_records(rec, dist, cdev, force=True)
    else:
        recs = np.asarray([rec])
    total_pairs = int(recs[:, 0].sum())
    ranks = [{"rank": r, "device": int(x[6]), "pairs": int(x[0]), "pairs_per_s": x[0] / x[4],
              "host_cores_used": round(x[2], 2), "workers": int(x[3]),
              "host_cpu_us_per_pair": round(1e6 * x[5], 1),
              # what this rank's share of the host cores could feed at that CPU cost per pair
              "host_core_ceiling_pairs_per_s": round(cores_per_rank / x[5]) if x[5] > 0 else None,
              "d1_valid_px_first_pair": int(x[1]),
              "oracle_mismatch_px_first_two_pairs": (int(x[7]) if x[7] >= 0 else None),
              "pci_bus_id": shard.unpack_bus_id(int(x[8])), "numa_node": int(x[9]), "cpus_bound": int(x[10])}
             for r, x in enumerate(recs)]
    placement = shard.check_placement([(q["device"], q["pci_bus_id"]) for q in ranks], ndev)
    if rank == 0 and world > 1 and not placement["ok"]:
        raise SystemExit("bench.py: " + placement["why"])

    # ---- roofline of the dominant kernel.  Three measurements, kept apart:
    #   in_run     HIP events around its launches DURING the timed steps: ~10 kernels of different workers
    #              share the device, so a launch's duration there is a residency, not a rate;
    #   isolated   the same launch (one group of pairs, one worker, nothing else on the device) timed with
    #              HIP events in a probe after the timed region, on this box and this build: THIS is
    #              `achieved` / `frac` (algorithmic bytes of the launch / its duration);
    #   counters   rocprofv3 --pmc passes committed under profiles/ (quoted only when their build stamp is
    #              the loaded library's): measured HBM bytes (`traffic`) and VALU instructions.
    # The machine-share model of round 3 stays as `modelled_share_normalised`, never as `frac`.
    roofline = None
    if rank == 0:
        prof = read_profile(S) if in_region else prof_all
        pmc = load_pmc(build, "_hd1080" if args.workload == "hd1080" else "")
        if prof:
            # (one workgroup per triangulation / pair: k_delaunay, k_lattice and k_stage_pack are latency chains that use
            # almost none of the machine -- they overlap with the other workers' kernels and are not what a roofline
            # describes; the dominant kernel is chosen among the ones that fill the device)
            # -- i.e. among the kernels bench.py has a byte model for: when several ranks share one GPU the in-run time of
            # every kernel is mostly waiting, and a grid kernel without a model once came out on top with frac > 1)
            fill = [k for k in prof if k in ALG_BYTES_DESIGN_PER_PIXEL] or list(prof)
            dom = max(fill, key=lambda k: prof[k][0])
            ms, cnt = prof[dom]
            avg_s = ms / cnt / 1e3
            gl = min(group, B)                       # pairs one launch covers
            sbytes = ALG_BYTES_PER_PIXEL.get(dom, 8.0) * N_PIX * gl              # SURVEY 8(d) staged model
            abytes = ALG_BYTES_DESIGN_PER_PIXEL.get(dom, 8.0) * N_PIX * gl       # compulsory bytes of the design
            in_run = abytes / avg_s / 1e9
            # isolated probe: ONE group on ONE worker -> the kernels of the group run one after the other
            S.set_lanes(1)
            S.lib().svh_profile_only(None)
            S.lib().svh_profile_reset()
            S.lib().svh_profile_enable(1)
            probe_reps = 5
            for _ in range(probe_reps + 1):
                stp = e.process_batch_device(gl, dI1.data_ptr(), dI2.data_ptr(), W * H, dD1.data_ptr(),
                                             dD2.data_ptr(), W * H * 4, W, H, W)
                if _ == 0:                            # (the first repetition re-sizes the lane pool)
                    S.lib().svh_profile_reset()
            torch.cuda.synchronize()
            S.lib().svh_profile_enable(0)
            prof_iso = read_profile(S)
            S.set_lanes(lanes)
            iso_us = {k: 1e3 * v[0] / v[1] for k, v in prof_iso.items()}
            iso_s = iso_us.get(dom, 1e6 * avg_s) / 1e6
            achieved = abytes / iso_s / 1e9
            tot_ms = sum(v[0] for v in prof_all.values()) or 1.0
            share = prof_all.get(dom, (ms, cnt))[0] / tot_ms
            region_bytes = ALG_BYTES_PER_PIXEL.get(dom, 8.0) * N_PIX * B * args.steps
            norm = region_bytes / (share * elapsed_local) / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "definition": "compulsory HBM bytes of one launch AS THE KERNEL IS DESIGNED (descriptors on the "
                                      "fly: %.1f B/pixel x %d pixels x %d pairs, bench.py ALG_BYTES_DESIGN_PER_PIXEL) / "
                                      "duration of that launch alone on the device, HIP events on its stream, average "
                                      "of %d launches in a probe after the timed region.  The SURVEY 8(d) staged-model "
                                      "figure is under staged_model_8d, counter bytes under traffic"
                                      % (ALG_BYTES_DESIGN_PER_PIXEL.get(dom, 8.0), N_PIX, gl, probe_reps),
                        "avg_launch_us": 1e6 * iso_s, "alg_bytes_per_launch": abytes, "alg_bytes_design": abytes,
                        "pairs_per_launch": gl,
                        "staged_model_8d": {"alg_bytes_per_launch": sbytes, "bytes_per_pixel": ALG_BYTES_PER_PIXEL.get(dom, 8.0),
                                            "achieved": sbytes / iso_s / 1e9, "frac": sbytes / iso_s / 1e9 / HBM_PEAK_GBS,
                                            "note": "round 4's headline definition (72 B/pixel for the dense matcher: stored "
                                                    "descriptors read per map); the kernel no longer moves these bytes"},
                        "isolated_kernels_us": {k: round(v, 2) for k, v in sorted(iso_us.items())},
                        "isolated_sum_us": round(sum(iso_us.values()), 1),
                        "in_run": {"achieved": in_run, "frac": in_run / HBM_PEAK_GBS, "avg_launch_us": 1e6 * avg_s,
                                   "launches_timed": int(cnt), "timed_region": bool(in_region),
                                   "note": "HIP events around each launch during the timed steps; the launches of "
                                           "the workers overlap (sum of kernel time / wall = %.1f in the probe "
                                           "step), so this is a residency, not a rate"
                                           % (tot_ms / 1e3 / max(elapsed_local / args.steps, 1e-9))},
                        "modelled_share_normalised": {
                            "achieved": norm, "frac": norm / HBM_PEAK_GBS, "machine_share": round(share, 4),
                            "definition": "MODEL (round 3's headline, kept for continuity): staged-model bytes of all "
                                          "launches in the timed region / (kernel's share of summed kernel time x wall)"},
                        "kernels_us_probe_step": {k: round(1e3 * v[0] / v[1], 2)
                                                  for k, v in sorted(prof_all.items())}}
            if pmc["notes"]:
                roofline["pmc_notes"] = pmc["notes"]
            tr, iss = pmc["traffic"], pmc["issue"]
            if pmc["devcount"] and args.workload == "kitti":
                # counters of the pipelined steady state itself (nothing serialised): what binds UNDER OVERLAP
                dcd = pmc["devcount"]["derived"]
                roofline["overlapped_counters"] = dict(dcd, source="profiles/%s: rocprofiler-sdk device counting service over "
                                                                   "the timed region of this same command (tools/devcount.cpp, "
                                                                   "bench.py --devcount), build-stamped" % pmc["devcount_file"],
                                                       pairs_per_s_of_that_run=round(pmc["devcount"]["runs"][0]["pairs_per_s"]))
            alias_back = {"k_support": "k_support_lds", "k_match": "k_match_list", "k_descriptor": "k_descriptor_stream"}
            alias = {v: k for k, v in alias_back.items()}
            alias["k_match_keyed"] = "k_match"
            # measured HBM traffic of that kernel (rocprofv3 --pmc passes, taken on this workload's image size)
            if tr:
                kk = tr["kernels"].get(dom)
                if kk:
                    roofline["traffic"] = kk["hbm_bytes"] * gl / tr["pairs_per_launch"]
                    roofline["frac_on_measured_traffic"] = roofline["traffic"] / iso_s / 1e9 / HBM_PEAK_GBS
                    roofline["traffic_source"] = "profiles/%s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate " \
                                                 "passes), read = 2*FETCH_SIZE*1024 (gfx950), scaled to pairs " \
                                                 "per launch" % pmc["traffic_file"]
            # Issue-slot view of the same launch: wave-level VALU instructions (SQ_INSTS_VALU of the PMC
            # pass, scaled to the pairs per launch) against 1024 SIMDs x 2.4 GHz / 4 cycles per wave64
            # instruction of the SAD / min / compare class (profiles/r03_microbench_valu.txt).
            if iss:
                kk = iss["kernels"].get(alias_back.get(dom, dom)) or iss["kernels"].get(dom)
                if kk:
                    peak = 1024 * 2.4e9 / 4
                    instr = kk["valu_wave_instr"] * gl / float(tr["pairs_per_launch"] if tr else 4)
                    vfrac = instr / iso_s / peak
                    roofline["valu_issue_dominant_kernel"] = {
                        "wave_instr_per_launch": round(instr), "peak_wave_instr_per_s": peak,
                        "frac_isolated": round(vfrac, 3),
                        "frac_isolated_pmc_launch": round(kk["valu_wave_instr"] / (kk["launch_us_under_pmc"] * 1e-6) / peak, 3),
                        "lds_bank_conflict_cycles": kk.get("lds_bank_conflict_cycles"),
                        "waves_per_simd": kk.get("waves_per_simd"), "parked": kk.get("parked"),
                        "note": "SQ_INSTS_VALU x 4 cycles over the isolated launch time (issue-slot model backed by "
                                "the opcode microbenchmark); parked = share of wave time waiting (SQ_WAIT_ANY)"}
                    # what binds this kernel: the larger of its two utilisations, if either is telling
                    hfrac = roofline.get("frac_on_measured_traffic") or roofline["frac"]
                    roofline["bound"] = "valu_issue" if vfrac > hfrac else "hbm"
                    roofline["bound_evidence"] = {"valu_issue_frac": round(vfrac, 3), "hbm_frac": round(hfrac, 3),
                                                  "note": "isolated launch; neither roof is reached: the rest is LDS "
                                                          "and memory latency that 6 waves per SIMD do not cover"}
            # whole-path view: staged-model bytes of all pairs / wall time
            e2e = ALG_BYTES_PER_PIXEL_PAIR * N_PIX * B * args.steps / elapsed_local / 1e9
            roofline["end_to_end"] = {"alg_bytes_per_pair": ALG_BYTES_PER_PIXEL_PAIR * N_PIX,
                                      "achieved": e2e, "frac": e2e / HBM_PEAK_GBS,
                                      "note": "this rank's pairs; staged model 174.8 B/pixel"}
            if tr:
                twice = ("k_adaptive_mean", "k_gap_local", "k_owner")   # two launches per group share a symbol
                per_pair = sum(v["hbm_bytes"] * (2 if k in twice else 1)
                               for k, v in tr["kernels"].items()) / tr["pairs_per_launch"]
                mt = per_pair * B * args.steps / elapsed_local / 1e9
                roofline["end_to_end"].update(measured_hbm_bytes_per_pair=per_pair, measured_achieved=mt,
                                              measured_frac=mt / HBM_PEAK_GBS)
            # the box's own copy bandwidth (device-to-device, read + write counted)
            src = torch.empty(1 << 29, dtype=torch.uint8, device=dev)
            dst = torch.empty_like(src)
            dst.copy_(src)
            torch.cuda.synchronize()
            tc = time.perf_counter()
            for _ in range(10):
                dst.copy_(src)
            torch.cuda.synchronize()
            copy_gbs = 10 * 2 * src.numel() / (time.perf_counter() - tc) / 1e9
            del src, dst
            roofline["measured_copy_GBps"] = copy_gbs
            roofline["frac_of_measured_copy"] = achieved / copy_gbs
            if tr and iss:
                # every kernel of the pipeline from the committed isolated counter passes:
                # measured HBM bytes per launch (FETCH/WRITE_SIZE passes) / its duration
                rows = []
                for sym, v in iss["kernels"].items():
                    tk = tr["kernels"].get(alias.get(sym, sym))
                    if tk and v["launch_us_under_pmc"] > 0:
                        gbs = tk["hbm_bytes"] / v["launch_us_under_pmc"] / 1e3
                        rows.append({"kernel": alias.get(sym, sym), "hbm_GBps": round(gbs, 1),
                                     "frac_of_peak": round(gbs / HBM_PEAK_GBS, 3),
                                     "valu_busy": v["valu_busy"]})
                rows.sort(key=lambda r: -r["hbm_GBps"])
                # issue-slot view of the same run: wave-level VALU instructions of one pair (PMC,
                # kernels serialised) x this run's pairs/s, against 1024 SIMDs issuing one VALU
                # instruction per 4 cycles at the nominal 2.4 GHz
                if True:
                    per_pair = sum(v["valu_wave_instr"] for v in iss["kernels"].values()) / float(tr["pairs_per_launch"])
                    rate = B * args.steps / elapsed_local
                    roofline["valu_issue"] = {"wave_instr_per_pair": round(per_pair),
                                              "frac_of_issue_slots": round(per_pair * rate * 4 / (1024 * 2.4e9), 3),
                                              "note": "SQ_INSTS_VALU per pair (profiles/%s) x measured pairs/s; 256 CUs "
                                                      "x 4 SIMDs, 4 cycles per wave64 VALU op of the classes these "
                                                      "kernels are made of (profiles/r03_microbench_valu.txt: "
                                                      "v_sad_u8 / min / max / VOP3 / DPP / compares 4.2-4.5 cycles, "
                                                      "add / and / or / xor / fp32 fma 2.4-2.6) -- a model, not a "
                                                      "busy counter" % pmc["issue_file"]}
                    if roofline.get("overlapped_counters"):
                        oc = roofline["overlapped_counters"]
                        roofline["valu_issue"]["measured_overlapped"] = {
                            "valu_active": oc.get("valu_active"), "valu_per_pair": oc.get("valu_per_pair"),
                            "waves_per_simd": oc.get("waves_per_simd"), "parked": oc.get("parked"),
                            "lds_busy": oc.get("lds_busy"), "hbm_frac_of_8TBps": oc.get("hbm_frac_of_8TBps"),
                            "note": "busy counters of the steady state (SQ_ACTIVE_INST_VALU x 4 / SIMD-cycles): the vector "
                                    "ALUs are the busiest resource of the pipeline; what is left is wave time parked on "
                                    "memory / LDS / barriers that the resident waves do not cover"}
                        roofline["bound"] = "valu_issue"
                        roofline["bound_evidence"] = dict(roofline.get("bound_evidence") or {},
                                                          overlapped_valu_active=oc.get("valu_active"),
                                                          overlapped_hbm_frac=oc.get("hbm_frac_of_8TBps"))
                roofline["isolated_kernels_hbm"] = {"source": "profiles/%s + %s (rocprofv3 --pmc, kernels "
                                                              "serialised, %s workload)"
                                                              % (pmc["traffic_file"], pmc["issue_file"], args.workload),
                                                    "top": rows[:5]}
    if use_dist:
        dist.barrier()

    # self-check of the timed region's outputs: the maps of the first unique pairs, as the last timed
    # step left them, against the reference's own results (tests/golden, made by make_goldens.py from
    # oracle/_ref with the bench's parameters) -- bit for bit
    golden_check = None
    if rank == 0 and args.workload in ("kitti", "sequence") and data == "urban crops":
        gold = ["urban1_robotics", "urban2_kitti", "urban3_kitti", "urban4_kitti"]
        first = getattr(args, "seq_first", 0)
        bad = []
        for k in range(min(B, 8)):
            z = np.load(os.path.join(Hh.GOLDEN, gold[(first + k) % 4] + ".npz"))
            g1, g2 = dD1[k].cpu().numpy().ravel(), dD2[k].cpu().numpy().ravel()
            n1 = int((g1 != z["d1"]).sum())
            n2 = int((g2[: z["d2"].size] != z["d2"]).sum())
            if n1 or n2:
                bad.append({"pair": k, "golden": gold[(first + k) % 4], "d1_mismatch_px": n1, "d2_mismatch_px": n2})
        # ... and every other copy of the batch against the copy of its crop among those (on the device):
        # any pair a race or a stale buffer had touched anywhere in the batch shows up here
        def copies_differing():
            bad_ = 0
            if B > 4:
                n4 = (B // 4) * 4
                for dD in (dD1, dD2):
                    ref4 = dD[:4].unsqueeze(0)
                    for lo in range(0, n4, 1024):    # bounded temporaries: 1024 maps per comparison
                        hi = min(lo + 1024, n4)
                        eq = (dD[lo:hi].view(-1, 4, H, W) == ref4).flatten(2).all(dim=2)
                        bad_ += int((~eq).sum().item())
            return bad_
        copies_bad = copies_differing()
        soak = None
        if args.soak > 0 and not use_dist:
            gz = [np.load(os.path.join(Hh.GOLDEN, gold[(first + k) % 4] + ".npz")) for k in range(min(B, 4))]
            g1 = [torch.from_numpy(z_["d1"].reshape(H, W)).to(dev) for z_ in gz]
            g2 = [torch.from_numpy(z_["d2"].reshape(H, W)).to(dev) for z_ in gz]
            soak_bad = 0
            for _ in range(args.soak):
                dD1.zero_()
                dD2.zero_()
                torch.cuda.synchronize()
                step()
                torch.cuda.synchronize()
                for k in range(min(B, 4)):
                    soak_bad += int(not torch.equal(dD1[k], g1[k])) + int(not torch.equal(dD2[k], g2[k]))
                soak_bad += copies_differing()
            soak = {"steps": args.soak, "pairs_verified": args.soak * B, "maps_differing": soak_bad}
        golden_check = {"pairs_checked": min(B, 8), "mismatches": bad,
                        "copies_checked": max(0, (B // 4) * 4 - 4), "copies_differing": copies_bad, "soak": soak,
                        "what": "D1 and D2 of the first pairs after the last timed step == reference "
                                "Elas::process on the same crops (tests/golden/*.npz), every pixel; every "
                                "other pair of the batch == the copy of its crop among the first four"}

    if rank == 0:
        valid = float((dD1[:min(B, 64)] >= 0).float().mean().item())
        seq_note = ("configs[2]: %d-frame %dx%d KITTI raw sequence, frames sharded over GPUs"
                    % (kitti_frames[2], W, H)) if kitti_frames is not None else \
            "configs[2] substitute: 430-frame 1242x375 sequence (the four urban crops cycled), frames " \
            "sharded over GPUs"
        out = {
            "metric": "stereo pairs/sec (ELAS %dx%d, ROBOTICS, D1+D2+LR)" % (W, H),
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "ranks_seen": int(len(recs)),
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_pair": 1e3 * elapsed / (args.steps * B),
            "higher_is_better": True,
            "scaling": "strong" if args.workload in ("sequence", "hd1080") else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": data,
            "config": {"workload": {"kitti": "configs[1]: KITTI-size 1242x375 pairs",
                                    "hd1080": "configs[3]: synthetic 1920x1080 pairs, disp_max 255",
                                    "sequence": seq_note}[args.workload] +
                                   ", ELAS ROBOTICS, D1+D2 + LR-check, subsampling=false, inputs "
                                   "and outputs resident in HBM",
                       "pairs_per_step_per_gpu": B, "unique_pairs_per_gpu": U, "lanes_per_gpu": lanes,
                       "pairs_per_launch": group, "host_cores_used": round(float(recs[:, 2].sum()), 1),
                       "host_cpu_quota": _cpu_quota(), "host_cores_per_rank": round(cores_per_rank, 2),
                       "dist_backend": args.dist_backend if world > 1 else None,
                       "gpus_visible": ndev, "build": build, "stage": args.stage, "api": api,
                       "stream_depth_pairs": depth if api == "stream" else None,
                       "hw_queues": hwq, "library_init": library_init, "preflight": preflight, "placement": placement,
                       "stage_groups_device_handed_back": list(S.stage_stats()),
                       "d1_valid_fraction": round(valid, 4)},
            "ranks": ranks,
            "roofline": roofline,
        }
        if devcount is not None:
            out["devcount"] = devcount
        if args.workload == "hd1080" and all(r_["oracle_mismatch_px_first_two_pairs"] is not None for r_ in ranks):
            out["outputs_match_oracle"] = all(r_["oracle_mismatch_px_first_two_pairs"] == 0 for r_ in ranks)
            out["oracle_check"] = "every rank: D1 and D2 of its first two pairs after the last timed step == the " \
                                  "reference's Elas::process (oracle/_ref) on the same synthetic inputs, every pixel"
        if golden_check is not None:
            out["outputs_match_golden"] = (len(golden_check["mismatches"]) == 0 and golden_check["copies_differing"] == 0
                                           and (golden_check["soak"] is None or golden_check["soak"]["maps_differing"] == 0))
            out["golden_check"] = golden_check
        if use_dist:
            out["config"]["dist_backend"] = args.dist_backend
            out["config"]["dist_initialised"] = True
        extras = world == 1 and not args.no_extras
        if extras and args.workload != "hd1080":
            # SURVEY 8(b) ownership contract: host pointers in, host pointers out, through the
            # batch entry (pinned buffers; `lanes` workers x `group` pairs in flight).  PCIe-inclusive.
            nb = min(B, 768)
            hI1 = torch.from_numpy(I1).repeat((nb + U - 1) // U, 1, 1)[:nb].contiguous().pin_memory()
            hI2 = torch.from_numpy(I2).repeat((nb + U - 1) // U, 1, 1)[:nb].contiguous().pin_memory()
            hD1 = torch.empty((nb, H, W), dtype=torch.float32).pin_memory()
            hD2 = torch.empty((nb, H, W), dtype=torch.float32).pin_memory()
            arr = C.c_void_p * nb
            a1 = arr(*[hI1[i].data_ptr() for i in range(nb)])
            a2 = arr(*[hI2[i].data_ptr() for i in range(nb)])
            d1 = arr(*[hD1[i].data_ptr() for i in range(nb)])
            d2 = arr(*[hD2[i].data_ptr() for i in range(nb)])
            st = (C.c_int32 * nb)()
            dims = (C.c_int32 * 3)(W, H, W)

            def host_step():
                rc = S.lib().svh_elas_process_batch(e._h, nb, a1, a2, d1, d2, dims, st)
                assert rc == 0, (rc, S.last_error())
            host_step()
            t = time.perf_counter()
            for _ in range(4):
                host_step()
            dt = time.perf_counter() - t
            same = bool(torch.equal(hD1[1].to(dev), dD1[1]) and torch.equal(hD2[1].to(dev), dD2[1]))
            out["throughput_host_buffers"] = {
                "value": 4 * nb / dt, "unit": "pairs/s", "pairs_per_call": nb, "calls": 4,
                "pairs_in_flight": lanes * group,
                "bytes_over_pcie_per_pair": 2 * N_PIX + 8 * N_PIX,
                "pcie_GBps": 4 * nb * 10 * N_PIX / dt / 1e9,
                "note": "svh_elas_process_batch: pinned host images in, pinned host D1+D2 out (the reference's "
                        "ownership contract, SURVEY 8b); PCIe-inclusive, never `value`"}
            out["throughput_host_buffers"]["maps_equal_device_path"] = same
            # ... and as a STREAM (round 6; BASELINE configs[2] is a sequence streamed frame by frame, the reference's
            # producer hands host frames over, readfromfilesthread.cpp:25-112): a 430-frame sequence in pinned host
            # memory through svh_elas_stream_push_n in rings of 43 frames, a consumer thread popping in order; maps
            # back in host memory.  Timed from the first push to the last pop, 3 passes.
            import threading as _th
            ns, ring = 430, 43
            ns = min(ns, nb)
            srm = e.stream(W, H)
            got = []

            def seq_pass(passes):
                # `passes` x the 430-frame sequence as ONE stream: the producer keeps pushing rings while the consumer pops
                del got[:]
                cons = _th.Thread(target=lambda: got.extend(srm.pop_n(passes * ns)))
                cons.start()
                for _ in range(passes):
                    for r0 in range(0, ns, ring):
                        k = min(ring, ns - r0)
                        srm.push_n_raw(k, (C.c_void_p * k)(*a1[r0:r0 + k]), (C.c_void_p * k)(*a2[r0:r0 + k]),
                                       (C.c_void_p * k)(*d1[r0:r0 + k]), (C.c_void_p * k)(*d2[r0:r0 + k]))
                srm.flush()
                cons.join()
            hD1.zero_(); hD2.zero_()
            seq_pass(1)
            t = time.perf_counter()
            for _ in range(3):
                seq_pass(1)              # one sequence at a time: each pass pays the stream's ramp-up and drain
            dts = time.perf_counter() - t
            ok1 = got == [0] * ns
            t = time.perf_counter()
            seq_pass(6)                  # the same frames as one long stream (2 580 frames): the steady state
            dtl = time.perf_counter() - t
            srm.close()
            same_s = bool(ok1 and got == [0] * (6 * ns) and all(
                torch.equal(hD1[i].to(dev), dD1[i]) and torch.equal(hD2[i].to(dev), dD2[i]) for i in (0, 1, ns // 2, ns - 1)))
            out["throughput_host_buffers_stream"] = {
                "value": 6 * ns / dtl, "unit": "pairs/s", "frames": 6 * ns, "ring": ring,
                "per_sequence_of_430": {"value": 3 * ns / dts, "passes": 3,
                                        "note": "first push to last pop of ONE 430-frame sequence (37 ms): ramp-up and drain included"},
                "vs_batch_entry": round(6 * ns / dtl / out["throughput_host_buffers"]["value"], 3),
                "pcie_GBps": 6 * ns * 10 * N_PIX / dtl / 1e9, "maps_equal_device_path": same_s,
                "note": "svh_elas_stream_push_n / pop_n: pinned host frames in, maps back to pinned host memory, submission "
                        "order preserved; a producer pushing rings of %d frames and a consumer popping, the %d-frame "
                        "sequence six times over as one stream; PCIe-inclusive, never `value`" % (ring, ns)}
            del hI1, hI2, hD1, hD2
        if extras and args.workload == "kitti" and args.data == "urban":
            # the round-1 headline workload for comparison: seeded synthetic pairs, same step
            s1, s2 = make_inputs(min(B, args.unique), seed0=1000)
            dI1, dI2 = tile_to_device(s1, s2)
            torch.cuda.synchronize()
            step()
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(3):
                step()
            torch.cuda.synchronize()
            out["value_synthetic"] = {"value": 3 * B / (time.perf_counter() - t), "unit": "pairs/s",
                                      "data": "%d seeded synthetic pairs, tiled" % len(s1),
                                      "d1_valid_fraction": round(float((dD1[:64] >= 0).float().mean().item()), 4)}
        if extras:
            # single-stream latency of the reference-style call: one pair, pageable HOST buffers in
            # and out (PCIe-inclusive; never part of `value`)
            e1 = S.Elas(params)
            D1h = np.zeros((H, W), np.float32)
            D2h = np.zeros((H, W), np.float32)
            for _ in range(3):
                e1.process(I1[0], I2[0], D1h, D2h)
            t = time.perf_counter()
            for i in range(20):
                e1.process(I1[i % U], I2[i % U], D1h, D2h)
            out["latency_ms_single_pair_host_buffers"] = 1e3 * (time.perf_counter() - t) / 20
            out["latency_stages_ms"] = {k: round(v, 3) for k, v in e1.last_timing()}
            out["matcher"] = matcher_bench()     # before the CPU leg: the GPU is still at its clocks
            out["visual_odometry"] = vo_bench()
            out["map_fusion"] = map_bench()
            out["visual_odometry"]["replicas"] = vo_replicas_bench()
            out["visual_odometry"]["replicas"]["as_processes"] = vo_replicas_processes()
            out["visual_odometry"]["lockstep"] = vo_lockstep_bench()
            if args.workload == "kitti":
                out["application_settings"] = settings_bench(S, torch, dev)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(I1, I2, params, what, budget_s=args.cpu_budget,
                                               workers=int(avail) if args.workload == "kitti" else 0)
        print(json.dumps(out), flush=True)
    if use_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
