"""phase time stamps of k_lattice / k_delaunay (slot 0 of the last group) for a workload:
   python tools/gpu_stage_stamps.py [kitti|hd1080]"""
import ctypes as C
import os
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H
import svhip as S

wl = sys.argv[1] if len(sys.argv) > 1 else "kitti"
if wl == "hd1080":
    pairs = [H.synth_pair(1920, 1080, 1000 + i, dmax=200, planes=8) for i in range(2)]
else:
    pairs = [H.golden_pair("urban%d_1242x375" % i) for i in (1, 2)]
S.set_stage(1)
e = S.Elas(H.robotics())
I1 = np.stack([p[0] for p in pairs]); I2 = np.stack([p[1] for p in pairs])
for rep in range(3):
    st, D1, D2 = e.process_batch(I1, I2)
out = (C.c_int64 * 32)()
S.lib().svh_debug_stage_stamps(out)
t = [x * 0.01 for x in out]   # 10 ns ticks -> us
names = {0: "lattice start", 1: "copy in", 2: "consistency", 3: "redundancy", 4: "list", 8: "delaunay start",
         9: "ranks", 10: "cut order", 30: "end"}
print(wl, "statuses", st, "stage groups (device, handed back)", S.stage_stats())
for k in range(1, 5):
    print("k_lattice  %-12s %9.1f us" % (names[k], t[k] - t[k - 1]))
print("k_delaunay %-12s %9.1f us" % (names[9], t[9] - t[8]))
print("k_delaunay %-12s %9.1f us" % (names[10], t[10] - t[9]))
prev = t[10]
for k in range(11, 30):
    if out[k] == 0 or t[k] < prev:
        break
    print("k_delaunay depth step %2d %9.1f us" % (k - 11, t[k] - prev))
    prev = t[k]
print("k_delaunay tail        %9.1f us   total %9.1f us" % (t[30] - prev, t[30] - t[8]))
