cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for try in 1 2 3; do
  rm -rf /tmp/ht
  if timeout 600 rocprofv3 --hip-runtime-trace --stats -d /tmp/ht -o ht -- python $R/bench.py --no-cpu-baseline --steps 6 > /tmp/ht.log 2>&1; then break; fi
done
DB=$(find /tmp/ht -name "*.db" | head -1)
python - <<PY
import sqlite3
db=sqlite3.connect("$DB")
names=[r[0] for r in db.execute("select name from sqlite_master where type='view'")]
print(names)
for v in ("regions","hip_api","api"):
    if v in names:
        cols=[d[0] for d in db.execute("select * from %s limit 1"%v).description]
        print(v, cols)
try:
    rows=db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from regions group by name order by 3 desc limit 25").fetchall()
    for r in rows: print("%-40s %8d %10.1f ms %9.2f us"%r)
    span=db.execute("select (max(end)-min(start))/1e6 from regions").fetchone()
    print("span ms", span)
except Exception as e: print("ERR", e)
PY
tail -2 /tmp/ht.log | cut -c1-200
