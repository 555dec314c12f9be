set -x
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/bm_c2.json 2> gpurun_out/bm_c2.err || tail -5 gpurun_out/bm_c2.err
python bench.py --config 3 --steps 5 --warmup 2 > gpurun_out/bm_c3.json 2> gpurun_out/bm_c3.err || tail -5 gpurun_out/bm_c3.err
export MVS_BENCH_REHEARSAL=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bm_c2_r2.json 2> gpurun_out/bm_c2_r2.err || tail -5 gpurun_out/bm_c2_r2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --config 3 --steps 3 --warmup 1 > gpurun_out/bm_c3_r2.json 2> gpurun_out/bm_c3_r2.err || tail -5 gpurun_out/bm_c3_r2.err
python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29513 bench.py --gpus 4 --config 4 --steps 2 --warmup 1 > gpurun_out/bm_c4_r4.json 2> gpurun_out/bm_c4_r4.err || tail -5 gpurun_out/bm_c4_r4.err
for f in c2 c3 c2_r2 c3_r2 c4_r4; do python - <<PY
import json
try:
    d=json.load(open("gpurun_out/bm_$f.json"))
    print("$f", d["value"], d["unit"][:30], d["ms_per_step"], d["config"].get("collectives"), d["config"].get("kept_cells"), d.get("stages"))
except Exception as e:
    print("$f FAILED", e)
PY
done
