"""One-off extended fuzz of the device stage (not part of the test suite): seeded random points of
Elas::parameters x image shapes, product (svh_elas_set_stage(1), all stage taps) vs the oracle with
the real Triangle.  python tools/gpu_fuzz_stage.py [first_seed] [count]"""
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
shapes = [(320, 200), (401, 177), (512, 160), (288, 240), (640, 480), (1242, 375), (97, 61), (1000, 120)]
S.set_stage(1)
bad = 0
few = 0
for seed in range(first, first + count):
    prm = H.fuzz_elas_params(seed)
    w, h = shapes[seed % len(shapes)]
    l, r = H.synth_pair(w, h, seed, dmax=min(48, prm.disp_max - 8, w // 4))
    got = product_run(S, prm, l, r)
    want = H.oracle_elas_run(prm, l, r)
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
print("fuzz: %d points, %d with too few support points, %d mismatching; device-stage groups %d, handed back %d"
      % (count, few, bad, dev, back))
sys.exit(1 if bad else 0)
