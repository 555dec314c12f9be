# kernel-level timeline of one rank's C++ step of an 8-way split of configs[2] (bin/mvs_step_bench, exchange bytes in place)
set -x
REPO=$(pwd)
mkdir -p gpurun_out/g8b
python3 tools/make_synth_db.py 100000 2048 2345 /tmp/mvs_r06_db_100000_2048/ > gpurun_out/g8b/db.out 2>&1 || exit 1
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/g8b/trace -- $REPO/metagenome_vector_sketches_amd/bin/mvs_step_bench --db /tmp/mvs_r06_db_100000_2048/ --ranks 8 --rank ${RANK_SEL:-1} --steps 10 --warmup 5 --probe 0 > $REPO/gpurun_out/g8b/bench.out 2> $REPO/gpurun_out/g8b/bench.err || { tail -5 $REPO/gpurun_out/g8b/bench.err; exit 1; }
cd $REPO
f=$(find gpurun_out/g8b/trace -name '*kernel_trace.csv' | head -1)
[ -n "$f" ] && python3 tools/exp/step_kernels.py $f k_recode_rows 2 > gpurun_out/g8b/step_kernels.txt
cat gpurun_out/g8b/step_kernels.txt
rm -rf gpurun_out/g8b/trace
timeout -k 10 300 $REPO/metagenome_vector_sketches_amd/bin/mvs_step_bench --db /tmp/mvs_r06_db_100000_2048/ --ranks 8 --rank ${RANK_SEL:-1} > gpurun_out/g8b/bench_plain.out 2>&1
tail -5 gpurun_out/g8b/bench_plain.out
