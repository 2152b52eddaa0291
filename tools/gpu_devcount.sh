#!/bin/bash
# Device-wide counters over the timed region of an ordinary (overlapped) bench run: three runs, one counter set
# each (tools/devcount.cpp; bench.py --devcount).   usage: gpu_devcount.sh [suffix [extra bench.py arguments]]
cd $GRAFT_REPO_ROOT
O=gpurun_out; mkdir -p $O
SUF=$1; [ $# -gt 0 ] && shift
run() {  # name counters
  timeout 400 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --profile-in-timed-region 0 --devcount "$2" "${@:3}" \
      > $O/devcount_$1$SUF.json 2> $O/devcount_$1$SUF.err
  echo "devcount $1 rc $?"; tail -c 600 $O/devcount_$1$SUF.err
}
run issue GRBM_GUI_ACTIVE,SQ_BUSY_CYCLES,SQ_WAVES,SQ_WAVE_CYCLES,SQ_INSTS_VALU,SQ_ACTIVE_INST_VALU,SQ_INSTS_LDS,SQ_ACTIVE_INST_LDS,SQ_WAIT_INST_LDS "$@"
run wait GRBM_GUI_ACTIVE,SQ_WAIT_ANY,SQ_WAIT_INST_ANY,SQ_ACTIVE_INST_ANY,SQ_INSTS_SALU,SQ_INST_CYCLES_SALU,SQ_INSTS_VMEM_RD,SQ_LDS_BANK_CONFLICT,SQ_LDS_IDX_ACTIVE "$@"
run fetch FETCH_SIZE "$@"
run write WRITE_SIZE "$@"
python - <<PY
import json
for n in ("issue", "wait", "fetch", "write"):
    try:
        d = json.loads(open("$O/devcount_%s$SUF.json" % n).read().strip().splitlines()[-1])
        print(n, round(d["value"]), json.dumps(d.get("devcount", {}).get("counters")))
    except Exception as e:
        print(n, "no line:", e)
PY
python tools/devcount_summary.py $O/devcount_issue$SUF.json $O/devcount_wait$SUF.json $O/devcount_fetch$SUF.json $O/devcount_write$SUF.json > $O/devcount$SUF.json
python -c "import json; print(json.dumps(json.load(open('$O/devcount$SUF.json'))['derived']))"
