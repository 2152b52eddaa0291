#!/bin/bash
# k_match_list's instruction split: one counter pass per probe build (tools/Makefile `probe`, SVH_ML_PROBE = 1..8) and
# one for the product, kernels serialised (one lane, groups of 4 urban pairs, as tools/gpu_pmc_all.sh), then the event
# counters of probe 7.  -> gpurun_out/match_split.txt   (usage: gpu_match_split.sh [variants...], default all)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out; mkdir -p $O
ISO="python $R/bench.py --steps 2 --warmup 1 --no-extras --no-cpu-baseline --batch 32 --lanes 1 --group 4 --spinup 0 --profile-in-timed-region 0"
VARS=${@:-0 1 2 3 4 5 6 8}
OUT=$O/match_split.txt
python -c "import sys; sys.path.insert(0,'$R/stereo-vision_amd'); import svhip; print('#', svhip.lib().svh_version().decode())" > $OUT
for n in $VARS; do
  lib=$R/stereo-vision_amd/libsvhip.so; [ $n != 0 ] && lib=$R/tools/bin/libsvhip_probe$n.so
  for try in 1 2; do
    rm -rf /tmp/ms_$n
    if SVH_LIB=$lib timeout 150 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES -d /tmp/ms_$n -o p -- $ISO > /tmp/ms_$n.log 2>&1; then break; fi
  done
  db=$(find /tmp/ms_$n -name "*.db" | head -1)
  echo "== probe $n" >> $OUT
  python $R/tools/pmc_dump.py $db --only=k_match_list >> $OUT
done
# event counters (probe 7): four urban pairs, one group
SVH_LIB=$R/tools/bin/libsvhip_probe7.so python - >> $OUT <<PY
import sys, ctypes as C, numpy as np
sys.path.insert(0, "$R/stereo-vision_amd"); sys.path.insert(0, "$R/tests")
import helpers as H, svhip as S
L = S.lib()
ls, rs = zip(*[H.golden_pair("urban%d_1242x375" % i) for i in (1, 2, 3, 4)])
e = S.Elas(H.robotics())
e.process_batch(np.stack(ls), np.stack(rs))            # warm-up (allocations)
cnt = (C.c_ulonglong * 16)()
L.svh_probe_ml_counters(cnt, 1)
st, D1, D2 = e.process_batch(np.stack(ls), np.stack(rs))
L.svh_probe_ml_counters(cnt, 1)
names = ["wave_pixels", "wp_any_live", "live_lanes", "fast_wp", "fast_excl_wp", "fast_band_clipped_wp", "cold_wp",
         "trips_plain", "lane_trips_plain", "trips_excl", "lane_trips_excl", "live_lanes_fast", "trips_edge", "edge_wp"]
print("== event counters, 4 urban pairs (status %s)" % st)
for n, v in zip(names, cnt): print("   %-24s %12d" % (n, v))
PY
cat $OUT
