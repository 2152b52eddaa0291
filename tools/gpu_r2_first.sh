#!/bin/bash
# round-2 first GPU pass: new tests, the driver's bench command, the 2-rank path on this 1-GPU box
cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout 900 python -m pytest tests/test_multirank.py tests/test_elas_gpu.py -x -q -m gpu 2>&1 | tail -3
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/r2_bench_n1.json 2> $O/r2_bench_n1.err ) 2>&1 | grep real
tail -c 400 $O/r2_bench_n1.err
timeout 600 python bench.py --gpus 2 --steps 10 --warmup 2 --dist-backend gloo --no-extras --no-cpu-baseline > $O/r2_bench_n2_gloo.json 2> $O/r2_bench_n2_gloo.err
tail -c 300 $O/r2_bench_n2_gloo.err
SVH_HOST_PROF=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-extras --no-cpu-baseline > $O/r2_hostprof.json 2> $O/r2_hostprof.err
grep "host prof" $O/r2_hostprof.err
python - <<'PY'
import json
for f in ("r2_bench_n1","r2_bench_n2_gloo","r2_hostprof"):
    try:
        d=json.loads([l for l in open("gpurun_out/%s.json"%f) if l.startswith("{")][-1])
        print(f, round(d["value"]), d["n_gpus"], d["config"]["host_cores_used"], d["roofline"]["kernel"], round(d["roofline"]["frac"],3), [ (r["pairs_per_s"]//1, r["host_core_ceiling_pairs_per_s"]) for r in d["ranks"]])
        for k in ("value_synthetic","throughput_host_buffers","cpu_baseline"):
            if k in d: print("  ",k, {a:b for a,b in d[k].items() if a in ("value","nproc_workers","pcie_GBps","first_map_equals_device_path")})
    except Exception as e: print(f, "ERR", e)
PY
