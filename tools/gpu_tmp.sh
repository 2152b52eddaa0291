for cfg in "1 8" "2 4" "2 8" "4 2" "8 1"; do
  set -- $cfg
  python bench.py --workload hd1080 --no-cpu-baseline --group $1 --lanes $2 --steps 8 --warmup 2 > /tmp/o.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("/tmp/o.json"))
print("hd1080 group $1 lanes $2  pairs/s %.0f ms/step %.2f" % (d["value"], d["ms_per_step"]))
PY
done
