# round 5: block plans -- parity tests, then the strong-scaled step at 1 rank and rehearsed with 2 / 4 ranks on one card
set -x
timeout -k 10 900 python -m pytest tests/test_plan_gpu.py -x -q > gpurun_out/r5_plan_tests.log 2>&1 || { tail -40 gpurun_out/r5_plan_tests.log; exit 1; }
tail -3 gpurun_out/r5_plan_tests.log
timeout -k 10 300 python bench.py --config 3 --gpus 1 --steps 10 --warmup 3 > gpurun_out/r5_c3_g1.json 2> gpurun_out/r5_c3_g1.err || { tail -20 gpurun_out/r5_c3_g1.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/r5_c3_g1.json") if x.startswith("{")][-1]
d=json.loads(l)
print("c3_g1", d["value"], d["ms_per_step"], d["config"].get("kept_cells"), json.dumps(d["stages"]), d["roofline"]["frac"])
PY
