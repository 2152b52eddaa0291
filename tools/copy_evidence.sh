#!/bin/bash
# gpurun_out/evidence (tools/gpu_evidence.sh on the GPU box) -> profiles/<round>_*
R=${1:-r06}; E=gpurun_out/evidence
for f in $E/bench_line*.json $E/kernel_stats_*.txt $E/shard_driver_*.json; do cp $f profiles/${R}_$(basename $f); done
for n in pmc_issue pmc_traffic pmc_issue_hd1080 pmc_traffic_hd1080 devcount devcount_hd1080 matcher_probe; do cp $E/$n.json profiles/${R}_$n.json; done
cp $E/pmc_summary.txt profiles/${R}_pmc_summary.txt; cp $E/pmc_summary_hd1080.txt profiles/${R}_pmc_summary_hd1080.txt
cp $E/match_split_raw.txt profiles/${R}_match_split_raw_final.txt
cp $E/single_call_latency.txt profiles/${R}_single_call_latency.txt; cp $E/smoke.txt profiles/${R}_smoke.txt
cp $E/lockstep_phases.txt profiles/${R}_vo_lockstep_phases.txt; cp $E/lockstep_phases_pipelined.txt profiles/${R}_vo_lockstep_phases_pipelined.txt
cp $E/lockstep_libc_rand.txt profiles/${R}_vo_lockstep_libc_rand.txt
python tools/match_split_summary.py profiles/${R}_match_split_raw_baseline.txt profiles/${R}_match_split_raw_final.txt > profiles/${R}_match_split.txt; cat tools/match_split_notes.txt >> profiles/${R}_match_split.txt
python tools/design_table_update.py $R
