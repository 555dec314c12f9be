REPO=$(pwd)
mkdir -p gpurun_out/g8
cd /tmp && export TMPDIR=/tmp
export MVS_BENCH_TIMING=0
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/g8/dtrace -- python3 $REPO/tools/stream_bench.py 100000 2048 10000 1 encoded > $REPO/gpurun_out/g8/dense.out 2> $REPO/gpurun_out/g8/dense.err || { tail -5 $REPO/gpurun_out/g8/dense.err; exit 1; }
cd $REPO
kf=$(find gpurun_out/g8/dtrace -name '*kernel_trace.csv' | head -1)
python3 tools/exp/step_kernels.py $kf k_dense_count 0; python3 tools/exp/step_kernels.py $kf k_dense_count 1
rm -rf gpurun_out/g8/dtrace
