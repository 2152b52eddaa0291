for cfg in "4 16 768" "5 16 800" "6 16 768" "8 16 768" "6 16 1536" "6 12 768" "6 20 768" "12 16 768"; do
  set -- $cfg
  python bench.py --no-cpu-baseline --group $1 --lanes $2 --batch $3 --steps 5 --unique 128 > /tmp/o.json 2>/dev/null
  python - <<PY
import json
d=json.load(open("/tmp/o.json"))
print("group $1 lanes $2 batch $3  pairs/s %.0f  dom %s avg_us %.1f frac %.4f" % (d["value"], d["roofline"]["kernel"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"]))
PY
done
