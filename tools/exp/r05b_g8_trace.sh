# kernel-level timeline of one rank's step of an 8-way split (everything the exchange delivers already in place),
# + the guarded strong legs of bench.py --gpus 2 in a rehearsal (file transport, both ranks on this card)
set -x
REPO=$(pwd)
mkdir -p gpurun_out/g8
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --output-format csv -d $REPO/gpurun_out/g8/trace -- python3 $REPO/tools/strong_model.py 100000 2048 --ranks 8 --reps 6 > $REPO/gpurun_out/g8/model.out 2> $REPO/gpurun_out/g8/model.err || { tail -5 $REPO/gpurun_out/g8/model.err; exit 1; }
cd $REPO
f=$(find gpurun_out/g8/trace -name '*kernel_trace.csv' | head -1)
python3 tools/exp/step_kernels.py $f k_recode_rows 2 > gpurun_out/g8/step_kernels.txt
tail -50 gpurun_out/g8/step_kernels.txt
tail -4 gpurun_out/g8/model.out
rm -rf gpurun_out/g8/trace
export MVS_BENCH_REHEARSAL=1
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --strong-steps 2 > gpurun_out/g8/bench_r2.json 2> gpurun_out/g8/bench_r2.err || { tail -20 gpurun_out/g8/bench_r2.err; exit 1; }
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/g8/bench_r2.json') if x.startswith('{')][-1])
print('r2', d['value'], {k:(v.get('ms_per_step'), v.get('kept_cells')) if isinstance(v, dict) else v for k,v in d.get('strong',{}).items()})
"
timeout -k 10 300 python bench.py --gpus 2 --steps 3 --warmup 1 --strong-steps 2 --strong-timeout 2 > gpurun_out/g8/bench_r2_to.json 2> gpurun_out/g8/bench_r2_to.err; echo "rc=$?"
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/g8/bench_r2_to.json') if x.startswith('{')][-1])
print('r2 timeout', d['value'], d.get('strong'))
"
unset MVS_BENCH_REHEARSAL
# the 10 %-dense streamed leg: kernels and copies on one timeline
cd /tmp
timeout -k 10 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $REPO/gpurun_out/g8/dtrace -- python3 $REPO/tools/stream_bench.py 100000 2048 10000 2 encoded > $REPO/gpurun_out/g8/dense.out 2> $REPO/gpurun_out/g8/dense.err || { tail -5 $REPO/gpurun_out/g8/dense.err; exit 1; }
cd $REPO
kf=$(find gpurun_out/g8/dtrace -name '*kernel_trace.csv' | head -1)
cf=$(find gpurun_out/g8/dtrace -name '*memory_copy_trace.csv' | head -1)
head -2 $cf
python3 tools/exp/stream_timeline.py $kf $cf > gpurun_out/g8/dense_timeline.txt; tail -3 gpurun_out/g8/dense.out
head -30 gpurun_out/g8/dense_timeline.txt
rm -rf gpurun_out/g8/dtrace
