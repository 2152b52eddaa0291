# every step under its own timeout: a hung kernel must not eat the GPU budget
timeout 300 python -m pytest tests/test_elas_gpu.py -x -q 2>&1 | tail -2
timeout 120 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --batch 64 --lanes 1 --group 4 > gpurun_out/probe.json 2>gpurun_out/probe.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/probe.json"))
k=d["roofline"]["kernels_us_probe_step"]
print("probe pairs/s", round(d["value"]), "sum_us", round(sum(k.values()),1))
print(k)
PY
timeout 180 python bench.py --no-cpu-baseline > gpurun_out/bench.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/bench.json"))
print("bench", d["value"], d["roofline"]["kernel"], d["roofline"]["frac"])
PY
