# round-5 baseline on the round-4 sources: strong-scaled step at 1 rank and rehearsed with 2 / 4 ranks on one card
set -x
python bench.py --config 3 --gpus 1 --steps 10 --warmup 3 > gpurun_out/r5b_c3_g1.json 2> gpurun_out/r5b_c3_g1.err || tail -5 gpurun_out/r5b_c3_g1.err
export MVS_BENCH_REHEARSAL=1
timeout -k 10 300 python bench.py --gpus 2 --config 3 --steps 5 --warmup 2 > gpurun_out/r5b_c3_r2.json 2> gpurun_out/r5b_c3_r2.err || tail -5 gpurun_out/r5b_c3_r2.err
timeout -k 10 300 python bench.py --gpus 4 --config 3 --steps 5 --warmup 2 > gpurun_out/r5b_c3_r4.json 2> gpurun_out/r5b_c3_r4.err || tail -5 gpurun_out/r5b_c3_r4.err
for f in c3_g1 c3_r2 c3_r4; do python - <<PY
import json
try:
    l=[x for x in open("gpurun_out/r5b_$f.json") if x.startswith("{")][-1]
    d=json.loads(l)
    print("$f", d["value"], d["ms_per_step"], d["config"].get("collectives"), d["config"].get("kept_cells"), d.get("stages"))
except Exception as e:
    print("$f FAILED", e)
PY
done
