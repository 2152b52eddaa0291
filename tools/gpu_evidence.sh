#!/bin/bash
# Round evidence on the GPU box for the CURRENT build -> gpurun_out/evidence/ (copy what is to be
# judged into profiles/):
#   bench lines (driver command; 2 ranks sharing the GPU over gloo; hd1080; sequence)
#   rocprofv3 --kernel-trace --stats summaries of the same commands and of the secondary legs
#   PMC passes (SQ issue counters, FETCH_SIZE / WRITE_SIZE) -- counters only, separate runs
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
mkdir -p $O
# PMC passes first: bench.py quotes the summaries (measured HBM traffic, VALU counts) only when their
# build stamp is the loaded library's, so they must be in profiles/ before the bench lines are taken
GRAFT_REPO_ROOT=$R bash $R/tools/gpu_pmc_all.sh > $O/pmc_summary.txt 2>&1
cp $R/gpurun_out/pmc_issue.json $R/gpurun_out/pmc_traffic.json $O/
ROUND=${ROUND:-r03}
cp $R/gpurun_out/pmc_issue.json $R/profiles/${ROUND}_pmc_issue.json
cp $R/gpurun_out/pmc_traffic.json $R/profiles/${ROUND}_pmc_traffic.json
cd $R
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_line.json 2> $O/bench_line.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --no-extras --no-cpu-baseline > $O/bench_line_2ranks_gloo_1gpu.json 2> /dev/null
# hd1080 = configs[3]: the batch of 64 pairs on this one GPU (the default); then the share of one GPU of
# eight (8 pairs per step: latency-bound) on the host stage (automatic there) and on the device stage
timeout 600 python bench.py --workload hd1080 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_hd1080.json 2> /dev/null
timeout 600 python bench.py --workload hd1080 --batch 8 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_line_hd1080_b8.json 2> /dev/null
timeout 600 python bench.py --workload hd1080 --batch 8 --stage device --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_line_hd1080_b8_device_stage.json 2> /dev/null
timeout 600 python bench.py --workload hd1080 --batch 256 --group 16 --lanes 6 --steps 8 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_line_hd1080_x256.json 2> /dev/null
timeout 600 python bench.py --workload hd1080 --stage host --batch 64 --group 4 --lanes 8 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_hd1080_x64_host_stage.json 2> /dev/null
timeout 600 python bench.py --lanes 3 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_lanes3.json 2> /dev/null
timeout 600 python bench.py --lanes 6 --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_lanes6.json 2> /dev/null
timeout 600 python bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline --soak 60 > $O/bench_line_soak60.json 2> /dev/null
timeout 600 python bench.py --force-dist --dist-backend nccl --steps 10 --warmup 3 --no-extras --no-cpu-baseline > $O/bench_line_1rank_nccl.json 2> $O/bench_line_1rank_nccl.err
timeout 600 python bench.py --workload sequence --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/bench_line_sequence.json 2> /dev/null
cd /tmp
kt() {  # name cmd...
  local name=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/kt_$name
    if timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o $name -- "$@" > /tmp/kt_$name.log 2>&1; then break; fi
  done
  DB=$(find /tmp/kt_$name -name "*.db" | head -1)
  [ -n "$DB" ] && python $R/tools/rocpd_summary.py $DB > $O/kernel_stats_$name.txt
}
kt kitti python $R/bench.py --no-extras --no-cpu-baseline --steps 12 --warmup 3
kt hd1080 python $R/bench.py --workload hd1080 --no-extras --no-cpu-baseline --steps 8 --warmup 3
kt hd1080_b8 python $R/bench.py --workload hd1080 --batch 8 --no-extras --no-cpu-baseline --steps 20 --warmup 5
kt sequence python $R/bench.py --workload sequence --no-extras --no-cpu-baseline --steps 12 --warmup 3
kt matcher python $R/tools/gpu_legs.py matcher
kt vo python $R/tools/gpu_legs.py vo
kt map python $R/tools/gpu_legs.py map
kt vo_replicas16 python $R/tools/gpu_legs.py replicas16
ls -la $O | tail -20
python - <<PY
import json
for f in ("bench_line","bench_line_2ranks_gloo_1gpu","bench_line_hd1080","bench_line_hd1080_b8","bench_line_hd1080_b8_device_stage","bench_line_hd1080_x256","bench_line_hd1080_x64_host_stage","bench_line_sequence","bench_line_lanes3","bench_line_lanes6","bench_line_soak60","bench_line_1rank_nccl"):
    try:
        d=json.loads([l for l in open("$O/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"]), "n_gpus", d["n_gpus"], "cores", d["config"]["host_cores_used"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d.get("outputs_match_golden"), d["config"].get("stage_groups_device_handed_back"))
    except Exception as e: print(f, "ERR", e)
PY
GRAFT_REPO_ROOT=$R bash $R/tools/gpu_pmc_all.sh _hd1080 --workload hd1080 > $O/pmc_summary_hd1080.txt 2>&1
cp $R/gpurun_out/pmc_issue_hd1080.json $R/gpurun_out/pmc_traffic_hd1080.json $O/
