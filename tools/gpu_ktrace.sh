#!/bin/bash
# kernel trace of an arbitrary command: tools/gpu_ktrace.sh <out-name> <cmd...>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
name=$1; shift
for try in 1 2 3 4; do
  rm -rf /tmp/kt_$name
  if timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$name -o $name -- "$@" > /tmp/kt_$name.log 2>&1; then break; fi
done
DB=$(find /tmp/kt_$name -name "*.db" | head -1)
if [ -z "$DB" ]; then echo "gpu_ktrace: rocprofv3 failed in every try (see /tmp/kt_$name.log)" >&2; exit 1; fi
python $R/tools/rocpd_summary.py $DB | tee $R/gpurun_out/ktrace_$name.txt | head -${KT_LINES:-40}
