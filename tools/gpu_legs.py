"""The secondary legs of bench.py on their own (Matcher, visual odometry, map fusion), so that a
rocprofv3 kernel trace of this script shows their kernels only:
    rocprofv3 --kernel-trace --stats -- python tools/gpu_legs.py [matcher|vo|map|replicas[K]|lockstep[TxK]|settings[:name]]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "20")   # (as bench.py does before the runtime starts: a queue per worker stream)
import torch  # noqa: E402,F401  (HIP runtime of the torch wheel first)

torch.cuda.init()
import bench  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
out = {}
if which in ("matcher", "all"):
    out["matcher"] = bench.matcher_bench(iters=60)
if which in ("vo", "all"):
    out["visual_odometry"] = bench.vo_bench(iters=60)
if which in ("map", "all"):
    out["map_fusion"] = bench.map_bench(iters=40)
if which.startswith("replicas"):   # replicas, replicas16, ...
    ks = (int(which[8:]),) if which[8:] else (1, 4, 16)
    out["vo_replicas"] = bench.vo_replicas_bench(ks=ks, frames=60)
if which.startswith("lockstep"):   # lockstep, lockstep16, ...
    shapes = (tuple(int(x) for x in which[8:].split("x")),) if which[8:] else None   # lockstep2x8 = 2 threads x 8
    pl = os.environ.get("LOCKSTEP_PIPELINED")   # 0 / 1: only that form
    out["vo_lockstep"] = bench.vo_lockstep_bench(frames=60, private_rand=os.environ.get("LOCKSTEP_LIBC_RAND") is None,
                                                 pipelined=(False, True) if pl is None else (pl == "1",),
                                                 **({"shapes": shapes} if shapes else {}))
if which.startswith("settings"):   # settings, settings:middlebury, settings:subsampling, settings:texture   (one application_settings leg)
    import svhip as S
    out["application_settings"] = bench.settings_bench(S, torch, torch.device("cuda:0"), steps=int(os.environ.get("SETTINGS_STEPS", "3")),
                                                       only=which.split(":", 1)[1] if ":" in which else None)
print(json.dumps(out))
