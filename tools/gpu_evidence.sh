#!/bin/bash
# Round evidence on the GPU box for the CURRENT build -> gpurun_out/evidence/ (copy what is to be
# judged into profiles/ under the round's prefix):
#   PMC passes first (SQ issue counters, FETCH_SIZE / WRITE_SIZE; counters only, separate runs): bench.py
#   quotes them only when their build stamp is the loaded library's
#   bench lines: the driver's command; batch vs stream on the sequence and 1920x1080 workloads; 2 ranks
#   sharing the GPU over gloo; the RCCL branch with one rank; a soak run
#   rocprofv3 --kernel-trace --stats summaries of the same commands and of the secondary legs
ulimit -c 0
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/evidence
ROUND=${ROUND:-r06}
mkdir -p $O
GRAFT_REPO_ROOT=$R timeout 600 bash $R/tools/gpu_pmc_all.sh > $O/pmc_summary.txt 2>&1
cp $R/gpurun_out/pmc_issue.json $R/gpurun_out/pmc_traffic.json $O/
cp $R/gpurun_out/pmc_issue.json $R/profiles/${ROUND}_pmc_issue.json
cp $R/gpurun_out/pmc_traffic.json $R/profiles/${ROUND}_pmc_traffic.json
# the same passes on the 1920x1080 workload (bench.py --workload hd1080 quotes these)
GRAFT_REPO_ROOT=$R timeout 600 bash $R/tools/gpu_pmc_all.sh _hd1080 --workload hd1080 > $O/pmc_summary_hd1080.txt 2>&1
cp $R/gpurun_out/pmc_issue_hd1080.json $R/gpurun_out/pmc_traffic_hd1080.json $O/
cp $R/gpurun_out/pmc_issue_hd1080.json $R/profiles/${ROUND}_pmc_issue_hd1080.json
cp $R/gpurun_out/pmc_traffic_hd1080.json $R/profiles/${ROUND}_pmc_traffic_hd1080.json
# k_match_list's instruction split: the product and the eight probe builds (tools/Makefile `probe`), event counters
GRAFT_REPO_ROOT=$R timeout 1500 bash $R/tools/gpu_match_split.sh > /dev/null 2>&1
cp $R/gpurun_out/match_split.txt $O/match_split_raw.txt
# counters of the overlapped run (device-wide, nothing serialised): tools/devcount.cpp
make -C $R/tools libdevcount.so > /dev/null 2>&1
GRAFT_REPO_ROOT=$R timeout 900 bash $R/tools/gpu_devcount.sh > $O/devcount_summary.txt 2>&1
cp $R/gpurun_out/devcount.json $O/devcount.json
cp $R/gpurun_out/devcount.json $R/profiles/${ROUND}_devcount.json
GRAFT_REPO_ROOT=$R timeout 900 bash $R/tools/gpu_devcount.sh _hd1080 --workload hd1080 > $O/devcount_summary_hd1080.txt 2>&1
cp $R/gpurun_out/devcount_hd1080.json $O/devcount_hd1080.json
cd $R
b() { name=$1; shift; timeout 600 python bench.py "$@" > $O/bench_line$name.json 2> $O/bench_line$name.err; }
b "" --gpus 1 --steps 20 --warmup 5
Q="--no-extras --no-cpu-baseline"
b _kitti_stream $Q --steps 10 --warmup 3 --api stream
b _sequence $Q --workload sequence --steps 30 --warmup 5
b _sequence_batch_api $Q --workload sequence --steps 30 --warmup 5 --api batch
b _hd1080 $Q --workload hd1080 --steps 20 --warmup 3
b _hd1080_batch_api $Q --workload hd1080 --steps 20 --warmup 3 --api batch
b _hd1080_b8 $Q --workload hd1080 --batch 8 --steps 400 --warmup 20
b _hd1080_b8_batch_api_device_stage $Q --workload hd1080 --batch 8 --steps 100 --warmup 10 --api batch --stage device
b _hd1080_x256 $Q --workload hd1080 --batch 256 --group 16 --lanes 6 --steps 8 --warmup 2
b _2ranks_gloo_1gpu $Q --gpus 2 --steps 10 --warmup 2 --dist-backend gloo
b _1rank_nccl $Q --force-dist --dist-backend nccl --steps 10 --warmup 3
b _soak60 $Q --steps 8 --warmup 2 --soak 60
# the Matcher's timeline / roofline on its own, and the single Elas::process call (no environment variable anywhere).
# Before the eight-rank runs: for some seconds after eight processes have torn down their contexts a single call's
# second device phase takes 0.39 instead of 0.21 ms (measured: fresh 0.60, right after svh_shard --ranks 8 0.70, later 0.54)
timeout 200 python tools/matcher_probe.py 200 2> /dev/null | tail -1 > $O/matcher_probe.json
env -u GPU_MAX_HW_QUEUES timeout 200 python tools/gpu_single_latency.py 400 > $O/single_call_latency.txt 2>&1
# SCALE readiness: eight ranks sharing the one GPU over gloo, two workers each, small steps (bookkeeping dry runs)
b _8ranks_gloo_1gpu_kitti $Q --gpus 8 --dist-backend gloo --steps 4 --warmup 1 --batch 128 --lanes 2
b _8ranks_gloo_1gpu_sequence $Q --gpus 8 --dist-backend gloo --steps 4 --warmup 1 --workload sequence --lanes 2
b _8ranks_gloo_1gpu_hd1080 $Q --gpus 8 --dist-backend gloo --steps 8 --warmup 2 --workload hd1080 --lanes 2 --group 8   # (16 pairs per launch: the 32 lanes of eight ranks spend these few steps allocating)
# the C++ driver above the C-ABI (apps/svh_shard.cpp): one rank with its RCCL communicator, four ranks sharing the GPU
env -u GPU_MAX_HW_QUEUES timeout 300 stereo-vision_amd/bin/svh_shard --ranks 1 --gather rccl --pairs-per-rank 6144 --steps 20 --warmup 5 2> /dev/null | grep '^{' > $O/shard_driver_1rank_rccl.json
env -u GPU_MAX_HW_QUEUES timeout 300 stereo-vision_amd/bin/svh_shard --ranks 4 --pairs-per-rank 1536 --steps 20 --warmup 5 2> /dev/null | grep '^{' > $O/shard_driver_4ranks_pipes_1gpu.json
env -u GPU_MAX_HW_QUEUES timeout 300 stereo-vision_amd/bin/svh_shard --ranks 8 --total 430 --steps 20 --warmup 5 --lanes 2 2> /dev/null | grep '^{' > $O/shard_driver_8ranks_sequence_strong.json
SVH_MATCH_LIST=0 timeout 300 python bench.py $Q --steps 10 --warmup 3 > $O/bench_line_keyed_matcher.json 2> /dev/null
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.txt 2>&1
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
kt hd1080 python $R/bench.py --workload hd1080 --no-extras --no-cpu-baseline --steps 20 --warmup 3
kt hd1080_b8 python $R/bench.py --workload hd1080 --batch 8 --no-extras --no-cpu-baseline --steps 200 --warmup 10
kt sequence python $R/bench.py --workload sequence --no-extras --no-cpu-baseline --steps 20 --warmup 3
kt matcher python $R/tools/gpu_legs.py matcher
kt vo python $R/tools/gpu_legs.py vo
LOCKSTEP_PIPELINED=0 kt vo_lockstep16 python $R/tools/gpu_legs.py lockstep1x16
LOCKSTEP_PIPELINED=1 kt vo_lockstep16_pipelined python $R/tools/gpu_legs.py lockstep1x16
kt vo_replicas16 python $R/tools/gpu_legs.py replicas16
SETTINGS_STEPS=6 kt settings_middlebury python $R/tools/gpu_legs.py settings:middlebury
SETTINGS_STEPS=6 kt settings_subsampling python $R/tools/gpu_legs.py settings:subsampling
LOCKSTEP_PIPELINED=0 SVH_MATCHER_TIMING=1 timeout 300 python $R/tools/gpu_legs.py lockstep1x16 2>&1 | python $R/tools/show_lockstep.py > $O/lockstep_phases.txt
LOCKSTEP_PIPELINED=1 SVH_MATCHER_TIMING=1 timeout 300 python $R/tools/gpu_legs.py lockstep1x16 2>&1 | python $R/tools/show_lockstep.py > $O/lockstep_phases_pipelined.txt
LOCKSTEP_LIBC_RAND=1 timeout 300 python $R/tools/gpu_legs.py lockstep 2>&1 | python $R/tools/show_lockstep.py > $O/lockstep_libc_rand.txt
python - <<PY
import json, glob, os
for f in sorted(glob.glob("$O/bench_line*.json")):
    try:
        d=json.loads([l for l in open(f) if l.startswith("{")][-1]); r=d["roofline"]
        print(os.path.basename(f), round(d["value"]), "n_gpus", d["n_gpus"], "api", d["config"].get("api"), "cores", d["config"]["host_cores_used"], r["kernel"], "frac", round(r["frac"],3), "bound", r["bound"], "iso_us", round(r["avg_launch_us"],1), "golden", d.get("outputs_match_golden"), "oracle", d.get("outputs_match_oracle"))
    except Exception as e: print(os.path.basename(f), "ERR", e)
PY
cat $O/smoke.txt | tail -2
