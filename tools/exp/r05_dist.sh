# round 5: the multi-rank path with ranks sharing the one card (gloo / file transport), then the rehearsed strong-scaled bench
set -x
timeout -k 10 1000 python -m pytest tests/test_distributed_gpu.py -x -q > gpurun_out/r5_dist_tests.log 2>&1 || { tail -60 gpurun_out/r5_dist_tests.log; exit 1; }
tail -3 gpurun_out/r5_dist_tests.log
export MVS_BENCH_REHEARSAL=1
for g in 2 4; do
timeout -k 10 300 python bench.py --gpus $g --config 3 --steps 3 --warmup 1 > gpurun_out/r5_c3_r$g.json 2> gpurun_out/r5_c3_r$g.err || { tail -20 gpurun_out/r5_c3_r$g.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/r5_c3_r$g.json") if x.startswith("{")][-1]
d=json.loads(l)
print("c3_r$g", d["value"], d["ms_per_step"], d["config"].get("kept_cells"), json.dumps(d["stages"]))
PY
done
