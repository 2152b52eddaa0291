#!/bin/bash
# PMC evidence for the current build (kernels serialised: one lane, groups of 4 urban pairs):
#   SQ issue counters (two passes)  -> gpurun_out/pmc_issue.json
#   FETCH_SIZE / WRITE_SIZE passes  -> gpurun_out/pmc_traffic.json
# counters only (no --kernel-trace etc. in the same run), each pass under its own timeout
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out
# usage: gpu_pmc_all.sh [suffix [extra bench.py arguments]]   e.g.  gpu_pmc_all.sh _hd1080 --workload hd1080
SUF=$1; [ $# -gt 0 ] && shift
ISO="python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --batch 32 --lanes 1 --group 4 --spinup 0 --profile-in-timed-region 0 $@"
pmc() {  # name counters...
  local name=$1; shift
  for try in 1 2 3; do
    rm -rf /tmp/pmc_$name
    if timeout 300 rocprofv3 --pmc "$@" -d /tmp/pmc_$name -o $name -- $ISO > /tmp/pmc_$name.log 2>&1; then break; fi
  done
  find /tmp/pmc_$name -name "*.db" | head -1
}
A=$(pmc a SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS)
B=$(pmc b SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_SALU SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_RD SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE)
python $R/tools/pmc_issue.py $A $B > $O/pmc_issue$SUF.json
F=$(pmc fetch FETCH_SIZE)
W=$(pmc write WRITE_SIZE)
python $R/tools/pmc_summary.py $F $W 4 > $O/pmc_traffic$SUF.json
python - <<PY
import json
d=json.load(open("$O/pmc_issue$SUF.json")); t=json.load(open("$O/pmc_traffic$SUF.json"))
print(d.get("build"))
tot=0
for k,v in sorted(d["kernels"].items(), key=lambda kv:-kv[1]["valu_wave_instr"]):
    tot+=v["valu_wave_instr"]
    print("%-22s us %7.1f valu %9d busy(model) %.2f active(meas) %s cyc/valu %s salu %s lds %.2f w/simd %s parked %s stalled %s conf %s" % (k, v["launch_us_under_pmc"], v["valu_wave_instr"], v["valu_busy"], v.get("valu_active"), v.get("cyc_per_valu"), v.get("salu_cycles"), v["lds_busy"], v["waves_per_simd"], v["parked"], v["stalled"], v["lds_bank_conflict_cycles"]))
print("valu per pair", tot/4)
alias={"k_support_lds":"k_support","k_match_keyed":"k_match","k_match_list":"k_match"}
for k,v in t["kernels"].items():
    us=[x["launch_us_under_pmc"] for n,x in d["kernels"].items() if alias.get(n,n)==k]
    gbs=v["hbm_bytes"]/us[0]/1e3 if us and us[0]>0 else 0
    print("%-22s hbm MB %7.1f  %7.1f GB/s = %.2f of 8 TB/s" % (k, v["hbm_bytes"]/1e6, gbs, gbs/8000))
PY
