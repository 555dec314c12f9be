# Final records of round 6 (second session) on the final sources: the whole GPU suite, the default bench line, rocprofv3 stats + counters
# of the configs[1] step / configs[2] two-stage / one-rank strong step / rank 1 of 8 through the C++ host, every rank's C++ step of
# the 1 / 2 / 4 / 8-way splits (five runs) and the model on them.
set -x
REPO=$(pwd)
OUT=gpurun_out/r06
mkdir -p $OUT $OUT/step_bench
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $OUT/gputest.log 2>&1; echo "gpu tests rc=$?" | tee $OUT/gputest.rc; tail -2 $OUT/gputest.log
[ "$(cat $OUT/gputest.rc)" = "gpu tests rc=0" ] || exit 1
DB=/tmp/mvs_r06_db_100000_2048/
python3 tools/make_synth_db.py 100000 2048 2345 $DB > $OUT/db.out 2>&1 || exit 1
B=$REPO/metagenome_vector_sketches_amd/bin
for rep in 1 2 3 4 5; do
  for G in 1 2 4 8; do
    timeout -k 10 300 $B/mvs_step_bench --db $DB --ranks $G --steps 60 --warmup 60 > "$OUT/step_bench/step_bench_G${G}_run${rep}.json" 2> "$OUT/step_bench/step_bench_G${G}_run${rep}.stderr" || exit 1
  done
  echo "step_bench run $rep done"
done
python3 tools/strong_model.py --from-cpp $OUT/step_bench/step_bench_G*_run*.json > $OUT/strong_model_cpp_configs2.json 2> $OUT/strong_model.err || { tail -5 $OUT/strong_model.err; exit 1; }
tail -c 1500 $OUT/strong_model_cpp_configs2.json; echo
timeout -k 10 600 python3 bench.py > $OUT/bench.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
echo "bench done"
