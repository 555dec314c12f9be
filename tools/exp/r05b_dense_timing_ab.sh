# does the library's own timing (events + the host waiting for launch k before it queues launch k + 1) cost the streamed dense path its wall?
for t in 1 0 1 0; do
  echo "== MVS_BENCH_TIMING=$t"
  MVS_BENCH_TIMING=$t timeout -k 5 200 python3 tools/stream_bench.py 100000 2048 10000 3 encoded 2>&1 | grep "^run\|wall_ms" | cut -c1-330
done
