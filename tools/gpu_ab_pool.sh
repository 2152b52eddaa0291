#!/bin/bash
# A/B of the helper pool: calls taking turns (SVH_POOL_SERIAL=1) vs several jobs in flight (default);
# pipelined lockstep visual odometry, 1 x 16 and 2 x 16 objects, interleaved runs -> gpurun_out/ab_pool.txt
mkdir -p gpurun_out
out=gpurun_out/ab_pool.txt
: > $out
for rep in 1 2 3; do
  for shape in 1x16 2x16; do
    for serial in 1 0; do
      r=$(SVH_POOL_SERIAL=$serial LOCKSTEP_PIPELINED=1 SVH_MATCHER_TIMING=1 timeout 300 python tools/gpu_legs.py lockstep$shape 2> gpurun_out/ab_pool_err_${shape}_${serial}.txt | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read())['vo_lockstep']
print(json.dumps(d)[:900])")
      echo "rep $rep shape $shape serial $serial: $r" >> $out
      grep "lockstep timing" gpurun_out/ab_pool_err_${shape}_${serial}.txt | tail -2 >> $out
    done
  done
done
