"""single svh_elas_process call: wall time and the library's own stage split (svh_elas_last_timing), 1242x375 crops.
usage: SVH_LIB=... python tools/gpu_single_latency.py [calls]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import helpers as H, svhip as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pairs = [H.golden_pair("urban%d_1242x375" % i) for i in (1, 2, 3, 4)]
e = S.Elas(H.robotics())
h, w = pairs[0][0].shape
D1 = np.zeros((h, w), np.float32); D2 = np.zeros((h, w), np.float32)
for i in range(20): e.process(pairs[i % 4][0], pairs[i % 4][1], D1, D2)
# three passes: the first one after other work on the box may still see the device in another power state (the second
# device phase of a call was measured at 0.39-0.42 ms instead of 0.21 right after minutes of full load)
for rep in range(3):
    acc = {}
    t = time.perf_counter()
    for i in range(n):
        e.process(pairs[i % 4][0], pairs[i % 4][1], D1, D2)
        for k, v in e.last_timing(): acc[k] = acc.get(k, 0.0) + v
    dt = (time.perf_counter() - t) / n
    print(S.lib().svh_version().decode(), "pass %d: single call %.3f ms" % (rep, 1e3 * dt), {k: round(v / n, 3) for k, v in acc.items()})
    time.sleep(2.0)
