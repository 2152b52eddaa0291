cd $GRAFT_REPO_ROOT
for rep in 1 2; do for lib in libsvhip_A.so libsvhip.so; do
SVH_LIB=$GRAFT_REPO_ROOT/stereo-vision_amd/$lib timeout 280 python bench.py --no-cpu-baseline --steps 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); v=d['visual_odometry']
print('$lib vo', round(v['frame_ms'],3), 'lockstep', [(r['calling_threads'],r['objects_per_call'],r['pipelined'],round(r['frames_per_s']),r['host_cores_used']) for r in v['lockstep']['runs']])"
done; done
