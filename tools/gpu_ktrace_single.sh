#!/bin/bash
# kernel trace of single svh_elas_process calls (1242x375): which launches make up the two device phases of a call
R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
rm -rf /tmp/kt_single; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_single -o s -- python $R/tools/gpu_single_latency.py 300 > /tmp/kt_single.log 2>&1
tail -1 /tmp/kt_single.log | cut -c1-300
DB=$(find /tmp/kt_single -name "*.db" | head -1); python $R/tools/rocpd_summary.py $DB | head -40 | cut -c1-130
