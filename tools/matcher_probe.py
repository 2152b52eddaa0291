#!/usr/bin/env python3
"""Matcher timing on the reference quad: python tools/matcher_probe.py (SVH_MATCHER_TIMING=1 for stages)"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "stereo-vision_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench
r = bench.matcher_bench(iters=int(sys.argv[1]) if len(sys.argv) > 1 else 40)
print(json.dumps(r))
import gc; gc.collect()
