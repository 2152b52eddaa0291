python -m pytest tests/test_elas_gpu.py -x -q 2>&1 | tail -2
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
ISO="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --batch 32 --lanes 1 --group 4 --spinup 0 --profile-in-timed-region 0"
for try in 1 2 3; do rm -rf /tmp/pf; if timeout 300 rocprofv3 --pmc FETCH_SIZE -d /tmp/pf -o f -- $ISO > /tmp/pf.log 2>&1; then break; fi; done
F=$(find /tmp/pf -name "*.db" | head -1)
python - <<PY
import sqlite3,re
db=sqlite3.connect("$F")
grid={d:g for d,g in db.execute("select dispatch_id, grid_x*grid_y*grid_z from kernels")}
rows={}
for n,d,t in db.execute("select name,dispatch_id,sum(counter_value) from pmc_events where counter_name='FETCH_SIZE' group by name,dispatch_id"):
    m=re.search(r"(k_\w+)",n)
    if m: rows.setdefault(m.group(1),[]).append((grid.get(d,0),t))
for k,v in sorted(rows.items()):
    g=max(x for x,_ in v); s=[t for x,t in v if x==g]
    print("%-20s read MB per launch %.1f"%(k, 2*sum(s)/len(s)*1024/1e6))
PY
cd $R && bash tools/gpu_probe.sh 2>&1 | tail -4
