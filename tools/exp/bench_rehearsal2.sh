# two ranks sharing the one card (file transport + gloo): exercises the multi-rank branch of the default bench step
export MVS_BENCH_REHEARSAL=1
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29521 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/reh2.json 2> gpurun_out/reh2.err || tail -8 gpurun_out/reh2.err
python - <<PY
import json
l=[x for x in open("gpurun_out/reh2.json") if x.startswith("{")][-1]
d=json.loads(l)
print(d["value"], d["ms_per_step"], d["config"]["collectives"], d["config"]["kept_cells"], d["config"]["schedule"], d["stages"])
PY
