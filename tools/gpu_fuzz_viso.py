"""One-off extended fuzz of the Matcher / visual-odometry path (not part of the test suite): seeded random points of
Matcher::parameters x method x crop x predicted motion (helpers.fuzz_matcher_case) and of the VisualOdometryStereo
parameters (helpers.fuzz_vo_params), product (svh_matcher_* / svh_vo_* on the GPU) vs the oracle with the real Triangle.
python tools/gpu_fuzz_viso.py [first_seed] [matcher_points] [vo_points]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
import helpers as H  # noqa: E402
from test_matcher_gpu import push_quad, quad  # noqa: E402
from test_vo_gpu import TOL, run  # noqa: E402
from test_vo_gpu import quad as vo_quad  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
nm = int(sys.argv[2]) if len(sys.argv) > 2 else 200
nv = int(sys.argv[3]) if len(sys.argv) > 3 else 60
assert H.have_ref_viso(), "the oracle needs the real Triangle (oracle/_ref) for removeOutliers"
bad = 0
nmatch = 0
for seed in range(first, first + nm):
    prm, method, crop, tr = H.fuzz_matcher_case(seed)
    im = {k: v[crop] for k, v in quad().items()}
    a, b = H.OracleMatcher(prm), H.ProductMatcher(prm)
    rc = []
    for m in (a, b):
        push_quad(m, im)
        rc.append(m.match(method, tr))
    if rc[0] != rc[1] or rc[0] != 0:
        bad += rc[0] != rc[1]
        print("matcher seed", seed, "rc", rc)
        continue
    diff = [x for x in H.compare_matchers(a, b, method) if x[1] != 0]
    if diff:
        bad += 1
        print("matcher seed", seed, method, diff)
print("matcher fuzz: %d points, %d mismatching" % (nm, bad))
badv = 0
for seed in range(first, first + nv):
    prm = H.fuzz_vo_params(seed)
    a = run(H.OracleVo(prm), vo_quad())
    b = run(H.ProductVo(prm), vo_quad())
    same = (a[0] == b[0] and a[1].tobytes() == b[1].tobytes() and np.array_equal(a[2], b[2]) and
            np.abs(a[3] - b[3]).max() < TOL and a[4] == b[4])
    if not same:
        badv += 1
        print("vo seed", seed, "differs")
print("vo fuzz: %d points, %d mismatching" % (nv, badv))
sys.exit(1 if bad or badv else 0)
