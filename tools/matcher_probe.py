#!/usr/bin/env python3
"""Matcher timing on the reference quad: python tools/matcher_probe.py (SVH_MATCHER_TIMING=1 for stages)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
r = bench.matcher_bench(iters=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
import ctypes as C, svhip
st = (C.c_int64 * 4)()
try:
    svhip.lib().svh_host_helper_stats(st)
except AttributeError:   # an older build under SVH_LIB
    pass
r["host_helper_stats"] = {"tasks": st[0], "l3_moves": st[1], "to_polling": st[2], "to_sleeping": st[3]}
print(json.dumps(r))
import gc; gc.collect()
