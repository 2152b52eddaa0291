#!/usr/bin/env python3
"""Headline benchmark: ELAS stereo pairs/s on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--batch B] [--lanes L]

Workload (BASELINE.json configs[1]): KITTI-sized 1242x375 pairs, full ELAS
ROBOTICS parameters, D1+D2 with L/R check, subsampling off.  A "step" is one
pass of the hot path over one batch of B synthetic pairs that are already
resident in HBM; disparity maps are written to HBM.  For N>1 the driver starts
one process per GPU (torch.distributed.run); pairs are independent, so ranks
shard them with no data-path collective ("weak" scaling: B pairs per rank and
step) and only a tiny per-rank result record is gathered over RCCL.

Besides the contract fields the JSON line carries
  roofline      achieved algorithmic GB/s of the dominant kernel, from HIP events
                recorded on the kernels' own streams (svh_profile_*), vs 8 TB/s
  cpu_baseline  the reference Elas::process (oracle/_ref) timed on this host,
                1 thread, on a bounded sample of the same pairs.
torch is used only for device memory, the barrier and the result gather.
"""
import argparse
import ctypes as C
import json
import os
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


def make_inputs(batch, seed0=1000):
    import helpers as Hh
    I1 = np.empty((batch, H, W), np.uint8)
    I2 = np.empty((batch, H, W), np.uint8)
    for i in range(batch):
        I1[i], I2[i] = Hh.synth_pair(W, H, seed0 + i, dmax=96 if W < 1500 else 200, planes=8)
    return I1, I2


def cpu_baseline(I1, I2, params, budget_s=15.0):
    """reference (or port) on host cores, 1 thread, bounded sample of the same pairs"""
    import helpers as Hh
    n_done, t_used = 0, 0.0
    D1 = np.zeros((H, W), np.float32)
    D2 = np.zeros((H, W), np.float32)
    dims = (C.c_int32 * 3)(W, H, W)
    if Hh.have_ref_elas():
        lib = C.CDLL(Hh.ref_elas_path())   # no ref_init(1): plain allocator, fair timing
        kind = "reference"

        def run(i):
            lib.ref_elas_process(C.byref(params), I1[i].ctypes.data_as(C.c_void_p),
                                 I2[i].ctypes.data_as(C.c_void_p), D1.ctypes.data_as(C.c_void_p),
                                 D2.ctypes.data_as(C.c_void_p), dims)
    else:
        import svhip as S
        lib = Hh.oracle()
        kind = "port"
        tri = C.cast(S.lib().svh_delaunay, C.c_void_p)   # timing leg only: Triangle is not restated

        def run(i):
            lib.orc_elas_process(C.byref(params), I1[i].ctypes.data_as(C.c_void_p),
                                 I2[i].ctypes.data_as(C.c_void_p), D1.ctypes.data_as(C.c_void_p),
                                 D2.ctypes.data_as(C.c_void_p), dims, tri)
    run(0)  # warm
    i = 0
    while t_used < budget_s and n_done < 400:
        t = time.perf_counter()
        run(i % len(I1))
        t_used += time.perf_counter() - t
        n_done += 1
        i += 1
    return {"value": n_done / t_used, "unit": "pairs/s", "cores": 1, "kind": kind,
            "ms_per_pair": 1e3 * t_used / n_done,
            "sample": "%d x Elas::process on the bench's own %dx%d synthetic pairs, 1 thread, "
                      "%s" % (n_done, W, H, "oracle/_ref (reference compiled -O3 -msse3)" if kind == "reference"
                              else "oracle/ scalar port"),
            "host": _cpu_model(), "host_cores": os.cpu_count()}


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

    dev = Hh.ProductMatcher(prm)
    dev.lib.svh_matcher_set_taps(C.c_void_p(dev.h), 0)   # timing: no intermediate stage copies
    run(dev, 10)
    push, match, nm = run(dev, iters)
    out = {"workload": "quad match on libviso2/img I1p/I2p/I1c/I2c 1344x391, default parameters",
           "pushBack_ms": push, "matchFeatures_ms": match, "frame_ms": push + match,
           "frames_per_s": 1e3 / (push + match), "matches": nm}
    if Hh.have_ref_viso():
        ref = Hh.RefMatcher(prm)
        ref.lib.ref_init(0)
        rp, rm, rn = run(ref, 12)
        out["cpu_reference"] = {"pushBack_ms": rp, "matchFeatures_ms": rm, "frame_ms": rp + rm,
                                "matches": rn, "cores": 1, "kind": "reference"}
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
    run(dev, 5)
    frame, est, ok, nm, ni = run(Hh.ProductVo(prm), iters)
    out = {"workload": "VisualOdometryStereo::process on libviso2/img quad 1344x391, default parameters, "
                       "calibration of demo.cpp", "frame_ms": frame, "estimateMotion_ms": est,
           "frames_ok": ok, "matches": nm, "inliers": ni}
    if Hh.have_ref_viso():
        rf, re_, rok, rnm, rni = run(Hh.RefVo(prm), 10)
        out["cpu_reference"] = {"frame_ms": rf, "estimateMotion_ms": re_, "frames_ok": rok, "matches": rnm,
                                "inliers": rni, "cores": 1, "kind": "reference"}
    return out


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
                         "not available offline: the frames cycle the two committed KITTI-size crops "
                         "of the reference's urban images and two synthetic pairs")
    ap.add_argument("--batch", type=int, default=0,
                    help="pairs per step and GPU (0 = 768 for kitti, 8 for hd1080)")
    ap.add_argument("--unique", type=int, default=256,
                    help="different synthetic pairs generated per rank; the batch tiles them")
    ap.add_argument("--lanes", type=int, default=0,
                    help="double-buffered pipeline workers per GPU (0 = auto: 1.5 per available core, <= 24)")
    ap.add_argument("--group", type=int, default=0,
                    help="pairs per kernel launch, 1..16 (0 = 8 for kitti, 1 for hd1080)")
    ap.add_argument("--spinup", type=float, default=1.0,
                    help="seconds of untimed steps before the warmup (GPU clocks, lane buffers)")
    ap.add_argument("--profile-in-timed-region", type=int, default=1,
                    help="1: HIP-event kernel timing is on during the timed steps (roofline comes "
                         "from exactly those launches); 0: a separate pass after them")
    ap.add_argument("--dist-backend", default="nccl",
                    help="nccl (= RCCL, the default) or gloo (CPU collectives: lets several ranks "
                         "share one GPU when the multi-rank path is smoke-tested on a 1-GPU box)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--kitti-dir", default="",
                    help="--workload sequence on a real KITTI raw drive directory (image_00/, image_01/ "
                         "with data/ and timestamps.txt) instead of the 430-frame substitute")
    args = ap.parse_args()
    global W, H, N_PIX
    if args.workload == "hd1080":
        W, H = 1920, 1080
        N_PIX = W * H
    kitti_frames = None
    if args.workload == "sequence" and args.kitti_dir:
        pass   # the drive is read below, after torch has brought up the HIP runtime
    elif args.workload == "sequence":
        from svhip import shard as _sh
        lo_, hi_ = _sh.shard_range(430, int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")))
        args.batch = hi_ - lo_
        args.seq_first = lo_
    if args.batch <= 0:
        args.batch = 768 if args.workload == "kitti" else 8
    if args.group <= 0:
        # hd1080: 8 pairs per step, one pair per lane
        args.group = 1 if args.workload == "hd1080" else 8

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the product has no CPU path")
    if args.dist_backend == "gloo":
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    cdev = dev if args.dist_backend == "nccl" else torch.device("cpu")   # where collectives run
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    import svhip as S
    import helpers as Hh
    S.lib().svh_set_device(local_rank)
    if args.workload == "sequence" and args.kitti_dir:
        from svhip import kitti as _kitti
        k1, k2, lo_, total_ = _kitti.load_shard(args.kitti_dir, rank, world)
        kitti_frames = (k1, k2, total_)
        args.batch = k1.shape[0]
        args.seq_first = lo_
        H, W = k1.shape[1:]
        N_PIX = W * H
    # workers per GPU: each is double-buffered and sleeps while it waits, so ~1.5 per available
    # core keeps the cores busy with the host stage (lattice filters + Delaunay)
    avail = _cpu_quota() or (os.cpu_count() or 8)
    lanes = args.lanes or int(max(2, min(24, round(1.5 * avail / max(world, 1)))))
    S.set_lanes(lanes)
    group = S.set_group(args.group)

    B = args.batch
    params = Hh.robotics()           # Elas::parameters(ROBOTICS), elas.h:91-116
    # B pairs per step = `unique` different synthetic pairs, tiled (generation is the slow part;
    # the library keeps nothing between pairs, so a repeated pair is full work)
    if kitti_frames is not None:
        U = B
        dI1 = torch.from_numpy(kitti_frames[0]).to(dev).contiguous()
        dI2 = torch.from_numpy(kitti_frames[1]).to(dev).contiguous()
        I1, I2 = kitti_frames[0], kitti_frames[1]
    elif args.workload == "sequence":
        # frame f of the sequence = cycle[f % 4]; this rank owns frames [seq_first, seq_first + B)
        U = 4
        s1, s2 = make_inputs(2, seed0=4242)
        cyc = [Hh.golden_pair("urban1_1242x375"), Hh.golden_pair("urban2_1242x375"),
               (s1[0], s2[0]), (s1[1], s2[1])]
        I1 = np.stack([c[0] for c in cyc])
        I2 = np.stack([c[1] for c in cyc])
        idx = torch.tensor([(args.seq_first + k) % 4 for k in range(B)], device=dev)
        dI1 = torch.from_numpy(I1).to(dev)[idx].contiguous()
        dI2 = torch.from_numpy(I2).to(dev)[idx].contiguous()
    else:
        U = min(B, args.unique)
        I1, I2 = make_inputs(U, seed0=1000 + 100000 * rank)
        reps = (B + U - 1) // U
        dI1 = torch.from_numpy(I1).to(dev).repeat(reps, 1, 1)[:B].contiguous()
        dI2 = torch.from_numpy(I2).to(dev).repeat(reps, 1, 1)[:B].contiguous()
    dD1 = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    dD2 = torch.empty((B, H, W), dtype=torch.float32, device=dev)
    torch.cuda.synchronize()
    e = S.Elas(params)

    def step():
        st = e.process_batch_device(B, dI1.data_ptr(), dI2.data_ptr(), W * H, dD1.data_ptr(),
                                    dD2.data_ptr(), W * H * 4, W, H, W)
        assert all(s == 0 for s in st), st

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t_spin = time.perf_counter()
    while time.perf_counter() - t_spin < args.spinup:   # untimed: clocks up, lanes allocated
        step()
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
        dom0 = max(prof_all, key=lambda k: prof_all[k][0])
        S.lib().svh_profile_only(dom0.encode())
        S.lib().svh_profile_reset()
        S.lib().svh_profile_enable(1)
    barrier()
    import resource
    ru0 = resource.getrusage(resource.RUSAGE_SELF)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    ru1 = resource.getrusage(resource.RUSAGE_SELF)
    host_cores_used = ((ru1.ru_utime - ru0.ru_utime) + (ru1.ru_stime - ru0.ru_stime)) / elapsed
    if in_region:
        S.lib().svh_profile_enable(0)
        S.lib().svh_profile_only(None)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # the only collective on the path: a tiny per-rank result record over RCCL
        from svhip import shard
        recs = shard.gather_records([float(B * args.steps), float((dD1[0] >= 0).sum().item())],
                                    dist, cdev)
        total_pairs = int(recs[:, 0].sum())
    else:
        total_pairs = B * args.steps

    # ---- roofline of the dominant kernel from HIP events recorded on its own stream
    # during the timed steps (only that kernel is bracketed there: <1 % of value)
    roofline = None
    if rank == 0:
        prof = read_profile(S) if in_region else prof_all
        if prof:
            dom = max(prof, key=lambda k: prof[k][0])
            ms, cnt = prof[dom]
            avg_s = ms / cnt / 1e3
            # one launch covers a whole group of pairs
            abytes = ALG_BYTES_PER_PIXEL.get(dom, 8.0) * N_PIX * min(group, B)
            achieved = abytes / avg_s / 1e9
            roofline = {"bound": "hbm", "kernel": dom, "achieved": achieved, "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                        "avg_launch_us": 1e6 * avg_s, "alg_bytes_per_launch": abytes,
                        "timed_region": bool(in_region),
                        "kernels_us_probe_step": {k: round(1e3 * v[0] / v[1], 2)
                                                  for k, v in sorted(prof_all.items())}}
            # measured HBM traffic of that kernel (rocprofv3 --pmc passes, profiles/*_pmc_traffic.json)
            try:
                import glob
                pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]))
                kk = pmc["kernels"].get(dom)
                if kk and args.workload == "kitti":   # the PMC passes were taken on the KITTI workload
                    roofline["traffic"] = kk["hbm_bytes"] * min(group, B) / pmc["pairs_per_launch"]
                    roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), " \
                                                 "read = 2*FETCH_SIZE*1024 (gfx950), scaled to pairs per launch"
            except (OSError, IndexError, KeyError, ValueError):
                pass
            # whole-path view: staged-model bytes of all pairs / wall time
            e2e = ALG_BYTES_PER_PIXEL_PAIR * N_PIX * B * args.steps / elapsed / 1e9
            roofline["end_to_end"] = {"alg_bytes_per_pair": ALG_BYTES_PER_PIXEL_PAIR * N_PIX,
                                      "achieved": e2e, "frac": e2e / HBM_PEAK_GBS,
                                      "note": "this rank's pairs; staged model 174.8 B/pixel"}
            # measured HBM traffic of the whole pipeline (sum over the kernels of the PMC passes)
            try:
                import glob
                pmc = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]))
                twice = ("k_adaptive_mean", "k_gap_local", "k_owner")   # two launches per group share a symbol
                per_pair = sum(v["hbm_bytes"] * (2 if k in twice else 1)
                               for k, v in pmc["kernels"].items()) / pmc["pairs_per_launch"]
                if args.workload != "hd1080":
                    mt = per_pair * B * args.steps / elapsed / 1e9
                    roofline["end_to_end"].update(measured_hbm_bytes_per_pair=per_pair, measured_achieved=mt,
                                                  measured_frac=mt / HBM_PEAK_GBS)
            except (OSError, IndexError, KeyError, ValueError):
                pass
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
            # the streaming (HBM-bound) kernels, from the committed isolated measurements:
            # measured HBM bytes per 4-pair launch (FETCH/WRITE_SIZE passes) / its duration
            try:
                import glob
                tr = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_traffic.json")))[-1]))
                iss = json.load(open(sorted(glob.glob(os.path.join(ROOT, "profiles", "*_pmc_issue.json")))[-1]))
                alias = {"k_support_lds": "k_support", "k_match_keyed": "k_match"}
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
                if args.workload == "kitti":
                    per_pair = sum(v["valu_wave_instr"] for v in iss["kernels"].values()) / float(tr["pairs_per_launch"])
                    rate = total_pairs / elapsed / world
                    roofline["valu_issue"] = {"wave_instr_per_pair": round(per_pair),
                                              "frac_of_issue_slots": round(per_pair * rate * 4 / (1024 * 2.4e9), 3),
                                              "note": "SQ_INSTS_VALU per pair (profiles/*_pmc_issue.json) x measured "
                                                      "pairs/s; 256 CUs x 4 SIMDs, 4 cycles per wave64 VALU op"}
                roofline["isolated_kernels_hbm"] = {"source": "profiles/*_pmc_traffic.json + *_pmc_issue.json "
                                                              "(rocprofv3 --pmc, kernels serialised, KITTI workload)",
                                                    "top": rows[:5]}
            except (OSError, IndexError, KeyError, ValueError):
                pass
    if world > 1:
        dist.barrier()

    if rank == 0:
        valid = float((dD1 >= 0).float().mean().item())
        out = {
            "metric": "stereo pairs/sec (ELAS %dx%d, ROBOTICS, D1+D2+LR)" % (W, H),
            "value": total_pairs / elapsed,
            "unit": "pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps,
            "ms_per_pair": 1e3 * elapsed / (args.steps * B),
            "higher_is_better": True,
            "scaling": "strong" if args.workload == "sequence" else "weak",
            "vs_baseline": None,
            "dtype": "u8",
            "data": ("KITTI raw drive " + os.path.basename(os.path.normpath(args.kitti_dir))) if kitti_frames is not None
                    else ("synthetic" if args.workload != "sequence"
                          else "2 crops of the reference's urban images + 2 synthetic pairs"),
            "config": {"workload": {"kitti": "configs[1]: KITTI-size 1242x375 pairs",
                                    "hd1080": "configs[3]: synthetic 1920x1080 pairs, disp_max 255",
                                    "sequence": "configs[2] substitute: 430-frame 1242x375 sequence (two urban "
                                                "crops + two synthetic pairs, cycled), frames sharded over GPUs"
                                    }[args.workload].replace(
                                        "configs[2] substitute: 430-frame 1242x375 sequence (two urban crops + two "
                                        "synthetic pairs, cycled)",
                                        "configs[2]: %d-frame %dx%d KITTI raw sequence" % (
                                            kitti_frames[2] if kitti_frames is not None else 0, W, H)
                                        if kitti_frames is not None else
                                        "configs[2] substitute: 430-frame 1242x375 sequence (two urban crops + two "
                                        "synthetic pairs, cycled)") +
                                   ", ELAS ROBOTICS, D1+D2 + LR-check, subsampling=false, inputs "
                                   "and outputs resident in HBM",
                       "pairs_per_step_per_gpu": B, "unique_pairs_per_gpu": U, "lanes_per_gpu": lanes,
                       "pairs_per_launch": group, "host_cores_used": round(host_cores_used, 1),
                       "host_cpu_quota": _cpu_quota(),
                       "d1_valid_fraction": round(valid, 4)},
            "roofline": roofline,
        }
        if world == 1:
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
        if world == 1 and not args.no_cpu_baseline:
            out["matcher"] = matcher_bench()     # before the CPU leg: the GPU is still at its clocks
            out["visual_odometry"] = vo_bench()
            out["map_fusion"] = map_bench()
            out["cpu_baseline"] = cpu_baseline(I1, I2, params)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
