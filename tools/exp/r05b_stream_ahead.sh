# comparisons running ahead of the blocks' way out (option stream_ahead): tests, then the 10 %-dense 100k leg, encoded rows
timeout -k 10 500 python -m pytest tests/test_stream_gpu.py tests/test_encode_gpu.py -x -q 2>&1 | tail -2 || exit 1
for a in 1 6 1 6 3; do
  echo "== MVS_STREAM_AHEAD=$a (no timing events)"
  MVS_STREAM_AHEAD=$a MVS_BENCH_TIMING=0 timeout -k 5 200 python3 tools/stream_bench.py 100000 2048 10000 3 encoded 2>&1 | grep "^run [123]" | cut -c1-70
done
echo "== density sweep points, ahead 1 / 6"
for a in 1 6; do for c in 1024 4096; do MVS_STREAM_AHEAD=$a MVS_BENCH_TIMING=0 timeout -k 5 200 python3 tools/stream_bench.py 100000 2048 $c 3 encoded 2>&1 | grep "^run [23]" | cut -c1-100; done; done
MVS_STREAM_AHEAD=6 MVS_BENCH_TIMING=0 MVS_STREAM_TRACE=1 timeout -k 5 200 python3 tools/stream_bench.py 100000 2048 10000 2 encoded 2>&1 | grep "stream trace" | tail -1 | cut -c1-1800
