set -x
REPO=$(pwd)
mkdir -p gpurun_out/g8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/gpurun_out/g8/dtrace -- python3 $REPO/tools/stream_bench.py 100000 2048 10000 2 encoded > $REPO/gpurun_out/g8/dense.out 2> $REPO/gpurun_out/g8/dense.err || { tail -5 $REPO/gpurun_out/g8/dense.err; exit 1; }
cd $REPO
find gpurun_out/g8/dtrace -type f | head -20
kf=$(find gpurun_out/g8/dtrace -name '*kernel_trace.csv' | head -1)
cf=$(find gpurun_out/g8/dtrace -name '*memory_copy*.csv' | head -1)
tail -3 gpurun_out/g8/dense.out
if [ -n "$cf" ]; then head -3 $cf; python3 tools/exp/stream_timeline.py $kf $cf > gpurun_out/g8/dense_timeline.txt; head -40 gpurun_out/g8/dense_timeline.txt; fi
rm -rf gpurun_out/g8/dtrace
