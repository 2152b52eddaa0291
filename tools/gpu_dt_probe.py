"""isolated kernel times of one group of 32 KITTI-size pairs (one lane, HIP events around every kernel):
   python tools/gpu_dt_probe.py [kernel-name-prefix ...]      environment switches are read by the library"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import svhip as S

want = sys.argv[1:] or ["k_"]
pairs = [H.golden_pair("urban%d_1242x375" % (1 + i % 4)) for i in range(32)]
I1 = np.stack([p[0] for p in pairs]); I2 = np.stack([p[1] for p in pairs])
S.set_stage(1); S.set_lanes(1); S.set_group(32)
e = S.Elas(H.robotics())
lib = S.lib()
lib.svh_profile_only.argtypes = [C.c_char_p]
lib.svh_profile_get.argtypes = [C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_int64)]
st, D1, D2 = e.process_batch(I1, I2)
assert all(s == 0 for s in st), st
lib.svh_profile_only(None); lib.svh_profile_reset(); lib.svh_profile_enable(1)
for rep in range(5):
    e.process_batch(I1, I2)
lib.svh_profile_enable(0)
n = lib.svh_profile_get(-1, None, None, None)
tot = 0.0
out = []
for i in range(n):
    name, ms, cnt = C.c_char_p(), C.c_double(), C.c_int64()
    lib.svh_profile_get(i, C.byref(name), C.byref(ms), C.byref(cnt))
    us = 1e3 * ms.value / max(cnt.value, 1)
    tot += us
    if any(name.value.decode().startswith(w) for w in want):
        out.append("%s %.1f" % (name.value.decode(), us))
print(" ".join(out), "| sum %.1f us" % tot, "| env", {k: v for k, v in os.environ.items() if k.startswith("SVH_")})
