#!/bin/bash
# Kernel trace (device-to-host copies show as __amd_rocclr_copyBuffer kernels) of one row block of the 10 %-dense 100k leg, with the
# row passes beside the copies (stream_dense 1, the default) and between them (stream_dense 4): tools/exp/trace_block.py
REPO=$(pwd)
OUT=$REPO/gpurun_out/${1:-r06g}; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
export MVS_BENCH_TIMING=0
for mode in 1 4; do
  export MVS_STREAM_DENSE=$mode
  timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $OUT/dtrace$mode -- python3 $REPO/tools/stream_bench.py 100000 2048 10000 1 encoded > $OUT/dense_mode$mode.out 2> $OUT/dense_mode$mode.err || { tail -5 $OUT/dense_mode$mode.err; exit 1; }
  kf=$(find $OUT/dtrace$mode -name '*kernel_trace.csv' | head -1)
  (cd $REPO && echo "== stream_dense $mode: one row block (between two k_dense_count launches), late in the run" && python3 tools/exp/trace_block.py $kf k_dense_count -6) > $OUT/dense_block_mode$mode.txt
  rm -rf $OUT/dtrace$mode
done
unset MVS_STREAM_DENSE
