set -x
timeout -k 10 900 python -m pytest tests/test_plan_gpu.py tests/test_distributed_gpu.py tests/test_cli_gpu.py -x -q -m gpu > gpurun_out/a_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/a_tests.log
bash tools/exp/r06b_g8_trace.sh > gpurun_out/a_trace.log 2>&1; grep -A40 "^step of" gpurun_out/a_trace.log | head -45; tail -2 gpurun_out/g8b/bench_plain.out | cut -c1-400
