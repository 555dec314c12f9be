# k_search_filter: register buffers in flight (option search_depth) x query counts, 10^6 resident sketches
mkdir -p gpurun_out/g8
timeout -k 10 300 python -m pytest tests/test_search_gpu.py -x -q 2>&1 | tail -2 || exit 1
for nq in 64 256 512; do
for depth in 3 4 5 6; do
  echo "== search_depth $depth, $nq queries"
  MVS_SEARCH_DEPTH=$depth timeout -k 10 200 python3 tools/search_bench.py 1000000 2048 $nq 6 2>&1 | grep "^N " | head -1 | cut -c1-330
done
done
