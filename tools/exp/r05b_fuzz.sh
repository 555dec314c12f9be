# randomised cross-checks on the final build of the round
mkdir -p gpurun_out/fuzz
timeout -k 10 400 python tests/fuzz_plan.py --seconds 240 --seed 52 > gpurun_out/fuzz/plan.log 2>&1; echo "plan rc=$?"; tail -2 gpurun_out/fuzz/plan.log
timeout -k 10 300 python tests/fuzz_stream.py --seconds 150 --seed 53 > gpurun_out/fuzz/stream.log 2>&1; echo "stream rc=$?"; tail -2 gpurun_out/fuzz/stream.log
timeout -k 10 300 python tests/fuzz_project.py --seconds 100 --seed 54 > gpurun_out/fuzz/project.log 2>&1; echo "project rc=$?"; tail -2 gpurun_out/fuzz/project.log
