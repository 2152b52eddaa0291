"""One-off extended fuzz of the device stage (not part of the test suite): seeded random points of
Elas::parameters x image shapes, product (svh_elas_set_stage(1), all stage taps) vs the oracle with
the real Triangle.  python tools/gpu_fuzz_stage.py [first_seed] [count] [dup|gap]
"dup": candidate_stepsize 2-3 with lr_threshold 3-4 and noisy pairs, so that coincident right-image
support points occur (k_delaunay's replay of Triangle's quicksort decides which one survives)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
import helpers as H  # noqa: E402
import svhip as S  # noqa: E402
from test_elas_gpu import product_run  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 100
dup_mode = len(sys.argv) > 3 and sys.argv[3] == "dup"
gap_mode = len(sys.argv) > 3 and sys.argv[3] == "gap"   # wide interpolation gaps (MIDDLEBURY's 5000), with / without add_corners
ndup = 0
shapes = [(320, 200), (401, 177), (512, 160), (288, 240), (640, 480), (1242, 375), (97, 61), (1000, 120)]
S.set_stage(int(os.environ.get("SVH_FUZZ_STAGE", "1")))   # 1 = device stage (default), 0 = host stage
bad = 0
few = 0
for seed in range(first, first + count):
    prm = H.fuzz_elas_params(seed)
    w, h = shapes[seed % len(shapes)]
    if dup_mode:
        import numpy as np
        prm = prm.copy(candidate_stepsize=2 + seed % 2, lr_threshold=3 + (seed // 2) % 2, subsampling=0,
                       incon_min_support=2 + seed % 3, support_threshold=0.95)
        w, h = [(240, 120), (320, 200), (401, 177), (512, 160)][seed % 4]
        l, r = H.synth_pair(w, h, seed, dmax=30, noise=6)
    else:
        if gap_mode:
            prm = prm.copy(ipol_gap_width=[17, 40, 300, 5000][seed % 4], add_corners=(seed // 4) % 2)
        l, r = H.synth_pair(w, h, seed, dmax=min(48, prm.disp_max - 8, w // 4))
    got = product_run(S, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
    if dup_mode and want.status == 0:
        sup = want[H.SUPPORT].reshape(-1, 3)
        key = (sup[:, 0] - sup[:, 2]).astype(np.int64) * 65536 + sup[:, 1]
        ndup += len(np.unique(key)) < len(key)
    if got.status != want.status:
        bad += 1
        print("seed", seed, "status", got.status, want.status)
        continue
    if want.status != 0:
        few += 1
        continue
    diff = [(n, c) for n, c in H.compare_runs(want, got) if c != 0]
    if diff:
        bad += 1
        print("seed", seed, (w, h), diff)
dev, back = S.stage_stats()
print("fuzz: %d points, %d with too few support points, %d mismatching; device-stage groups %d, handed back %d%s"
      % (count, few, bad, dev, back, ("; %d with coincident right-image points" % ndup) if dup_mode else ""))
sys.exit(1 if bad else 0)
