#!/bin/bash
# A/B of an environment switch on the pipelined bench: tools/gpu_ab_env.sh VAR "v0 v1 ..." reps [bench.py args...]
# -> gpurun_out/ab_env_VAR.txt (pairs/s per run, interleaved)
var=$1; vals=$2; reps=$3; shift 3
mkdir -p gpurun_out
out=gpurun_out/ab_env_$var.txt
echo "# bench.py --no-extras --no-cpu-baseline $* ; $var in {$vals}, interleaved x $reps" >> $out
for rep in $(seq $reps); do
  for v in $vals; do
    r=$(env $var=$v timeout 300 python bench.py --no-extras --no-cpu-baseline "$@" 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(round(d['value']), d.get('outputs_match_golden'), d.get('outputs_match_oracle'), round(d['roofline']['isolated_kernels_us'].get('k_delaunay',0),1))")
    echo "rep $rep $var=$v: $r" >> $out
  done
done
cat $out
