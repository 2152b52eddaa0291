R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
for lib in libsvhip_A.so libsvhip.so; do
rm -rf /tmp/kt_$lib; SVH_LIB=$R/stereo-vision_amd/$lib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$lib -o m -- python $R/tools/gpu_legs.py matcher > /tmp/kt.log 2>&1
DB=$(find /tmp/kt_$lib -name "*.db" | head -1); echo "== $lib"; python $R/tools/rocpd_summary.py $DB | head -14 | cut -c1-120
done
