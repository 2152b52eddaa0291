ulimit -c 0
cd $GRAFT_REPO_ROOT
cd tests && timeout 900 python -m pytest -x -q -m gpu test_elas_gpu.py test_stage_gpu.py 2>&1 | tail -4; cd ..
for v in 0 1 0 1; do
SVH_DESC_PK=$v timeout 300 python bench.py --steps 10 --warmup 3 --no-extras --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['roofline']['isolated_kernels_us']; print('pk=$v', round(d['value']), 'golden', d.get('outputs_match_golden'), 'desc iso us', k['k_descriptor'], 'match', k['k_match'])"
done
