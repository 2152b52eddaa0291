#!/bin/bash
# Round-end evidence on the GPU box: bench line, rocprofv3 kernel trace of the same command,
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate --pmc passes).  Outputs under gpurun_out/.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
python $R/bench.py > $O/bench_full.json 2> $O/bench_full.err
KT_LINES=60 $R/tools/gpu_ktrace.sh bench python $R/bench.py --no-cpu-baseline > /dev/null
ISO="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 --lanes 1 --group 4 --spinup 0 --profile-in-timed-region 0"
pmc() {  # name counter
  for try in 1 2 3; do
    rm -rf /tmp/pmc_$1
    if timeout 300 rocprofv3 --pmc $2 -d /tmp/pmc_$1 -o $1 -- $ISO > /tmp/pmc_$1.log 2>&1; then break; fi
  done
  find /tmp/pmc_$1 -name "*.db" | head -1
}
F=$(pmc fetch FETCH_SIZE)
W=$(pmc write WRITE_SIZE)
python $R/tools/pmc_summary.py $F $W 4 > $O/pmc_traffic.json
tail -c 600 $O/bench_full.json
