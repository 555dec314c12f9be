# the encoded rows' way to the host: hipMemcpyAsync (a blit kernel that fills the card) against k_copy_to_host with few workgroups
timeout -k 10 400 python -m pytest tests/test_stream_gpu.py tests/test_encode_gpu.py -x -q 2>&1 | tail -1 || exit 1
for w in 0 8 16 32 64 128 0 32; do
  echo "== stream_copy_wgs $w"
  MVS_STREAM_COPY_WGS=$w MVS_BENCH_TIMING=0 timeout -k 5 200 python3 tools/stream_bench.py 100000 2048 10000 3 encoded 2>&1 | grep "^run [123]" | cut -c1-60
done
