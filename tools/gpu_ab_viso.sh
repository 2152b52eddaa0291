#!/bin/bash
# A/B of two builds (libsvhip_A.so = the older one) on ONE box, interleaved: Matcher frame + dense vote, single Elas::process call
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
for lib in libsvhip_A.so libsvhip.so; do
export SVH_LIB=$GRAFT_REPO_ROOT/stereo-vision_amd/$lib
echo -n "$lib matcher: "; timeout 120 python tools/matcher_probe.py 300 | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['frame_ms'],4), 'vote', d['timeline']['steps_ms']['matchFeatures: dense outlier vote (host)'], d['matcher_matches_reference'])"
echo -n "$lib single: "; timeout 120 python tools/gpu_single_latency.py 400 | cut -c40-300
done; done
