#!/bin/bash
# A/B of two builds of the library: isolated per-kernel times (1 lane, groups of 4 urban pairs)
# and the pipelined throughput.   tools/gpu_ab.sh [extra bench args]
cd $GRAFT_REPO_ROOT
for lib in libsvhip_A.so libsvhip.so; do
for rep in 1 2; do
SVH_LIB=$GRAFT_REPO_ROOT/stereo-vision_amd/$lib timeout 120 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --batch 64 --lanes 1 --group 4 --stage host --profile-in-timed-region 0 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['kernels_us_probe_step']; print('$lib iso', round(d['value']), 'owner', k['k_owner'], k['k_owner_fix'], 'support', k['k_support'], 'match', k['k_match'], 'sum', round(sum(k.values()),1))"
done
SVH_LIB=$GRAFT_REPO_ROOT/stereo-vision_amd/$lib GPU_MAX_HW_QUEUES=8 timeout 300 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline $@ 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib pipelined Q8', round(d['value']))"
done
