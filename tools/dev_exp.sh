set -x
cd $GRAFT_REPO_ROOT
for o in "" "--opt=large_merge:0"; do
  python tools/bench_large.py protein 30 $o 2>&1 | grep "^protein"
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3_pb4 -- python3 $GRAFT_REPO_ROOT/tools/bench_large.py protein 20 > /dev/null 2>&1
find $GRAFT_REPO_ROOT/gpurun_out/r3_pb4 -name "*kernel_stats.csv" | xargs cat
