set -x
timeout -k 10 600 python -m pytest tests/test_plan_gpu.py -x -q 2>&1 | tail -3
export MVS_BENCH_REHEARSAL=1
for g in 2 4; do
timeout -k 10 300 python bench.py --gpus $g --config 3 --steps 3 --warmup 1 > gpurun_out/r5_c3_r$g.json 2> gpurun_out/r5_c3_r$g.err || { tail -20 gpurun_out/r5_c3_r$g.err; exit 1; }
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r5_c3_r$g.json") if x.startswith("{")][-1])
print("c3_r$g", d["ms_per_step"], d["config"]["kept_cells"], d["config"]["cells_checksum"])
for x in d["timeline"]: print("   %9.3f ms  %s" % (x[1], x[0]))
PY
done
unset MVS_BENCH_REHEARSAL
python bench.py --config 3 --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_c3_g1.json 2> gpurun_out/r5_c3_g1.err
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r5_c3_g1.json") if x.startswith("{")][-1])
print("c3_g1", d["ms_per_step"], d["config"]["kept_cells"], d["config"]["cells_checksum"], json.dumps(d["stages"]))
for x in d["timeline"]: print("   %9.3f ms  %s" % (x[1], x[0]))
PY
