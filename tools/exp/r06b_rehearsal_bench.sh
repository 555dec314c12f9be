# second session, final sources: the configs[2] rehearsal of the C++ step under the executable again (the rebuild kernel changed: shard
# folders byte-identical to the round-1 scheme's, checksum = bench.py's), then the default bench line for profiles/r06_bench.json
set -x
STEPS=2 WARMUP=2 bash tools/exp/r06_step_rehearsal.sh gpurun_out/r06c > gpurun_out/r06c_rehearsal.log 2>&1 || { tail -20 gpurun_out/r06c_rehearsal.log; exit 1; }
cat gpurun_out/r06c/summary.txt
timeout -k 10 600 python3 bench.py > gpurun_out/r06_bench_final.json 2> gpurun_out/r06_bench_final.err || { tail -5 gpurun_out/r06_bench_final.err; exit 1; }
echo bench done
