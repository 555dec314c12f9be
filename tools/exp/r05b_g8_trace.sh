# kernel-level timeline of one rank's step of an 8-way split (everything the exchange delivers already in place)
set -x
REPO=$(pwd)
mkdir -p gpurun_out/g8
timeout -k 10 500 python -m pytest tests/test_plan_gpu.py tests/test_distributed_gpu.py -x -q 2>&1 | tail -3 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/g8/trace -- python3 $REPO/tools/strong_model.py 100000 2048 --ranks 8 --reps 6 > $REPO/gpurun_out/g8/model.out 2> $REPO/gpurun_out/g8/model.err || { tail -5 $REPO/gpurun_out/g8/model.err; exit 1; }
cd $REPO
f=$(find gpurun_out/g8/trace -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 tools/exp/step_kernels.py $f k_recode_rows 2 > gpurun_out/g8/step_kernels.txt
tail -45 gpurun_out/g8/step_kernels.txt
rm -rf gpurun_out/g8/trace
timeout -k 10 300 python3 tools/strong_model.py 100000 2048 --ranks 1,8 --reps 10 > gpurun_out/g8/model_1_8.out 2>&1; grep -v "^ *122" gpurun_out/g8/model_1_8.out
