#!/bin/bash
# SQ counter passes over a short isolated run (lanes=1: kernels do not overlap)
#   tools/gpu_pmc.sh [kernel,list]   -> gpurun_out/pmc_sq.txt, gpurun_out/pmc_sq_[ab].db
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CMD="python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --batch 16 --no-extras --lanes 1 --group 4 --spinup 0 --profile-in-timed-region 0"
run() {  # name counters...
  local name=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmc_$name
    if timeout 300 rocprofv3 --pmc "$@" -d /tmp/pmc_$name -o $name -- $CMD > /tmp/pmc_$name.log 2>&1; then break; fi
  done
  find /tmp/pmc_$name -name "*.db" | head -1
}
A=$(run a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS)
B=$(run b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)
echo "dbs: $A $B"
cp $A $R/gpurun_out/pmc_sq_a.db; cp $B $R/gpurun_out/pmc_sq_b.db
python $R/tools/pmc_dump.py $A $B --only=${1:-k_match_list,k_support_lds} | tee $R/gpurun_out/pmc_sq.txt
