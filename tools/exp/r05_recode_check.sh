# round 5: radix trials on the 24-bit multipliers (k_recode_rows, k_coarse_build): parity tests, then times
set -x
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py tests/test_pairwise_gpu.py tests/test_search_gpu.py -x -q > gpurun_out/r05_recode_tests.log 2>&1 || { tail -30 gpurun_out/r05_recode_tests.log; exit 1; }
tail -2 gpurun_out/r05_recode_tests.log
python tools/strong_model.py 100000 2048 --ranks 1,8 > gpurun_out/r05_recode_model.log 2>&1
cat gpurun_out/r05_recode_model.log
MVS_BENCH_STEP_TIMES=1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r05_recode_bench.json 2> gpurun_out/r05_recode_bench.err
grep strong_run gpurun_out/r05_recode_bench.err
MVS_BENCH_STEP_TIMES=1 python bench.py --config 3 --gpus 1 --steps 10 --warmup 3 > gpurun_out/r05_recode_c3.json 2> gpurun_out/r05_recode_c3.err
grep strong_run gpurun_out/r05_recode_c3.err
