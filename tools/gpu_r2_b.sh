#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for st in device host; do
SVH_HOST_PROF=1 timeout 300 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --stage $st > $O/r2_stage_$st.json 2> $O/r2_stage_$st.err
grep "host prof" $O/r2_stage_$st.err
done
python - <<'PY'
import json
for f in ("r2_stage_device","r2_stage_host"):
    try:
        d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"]), d["config"]["host_cores_used"], d["config"]["stage_groups_device_handed_back"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), d["ranks"][0]["host_core_ceiling_pairs_per_s"])
        print("   ", d["roofline"]["kernels_us_probe_step"])
    except Exception as e: print(f, "ERR", e)
PY
