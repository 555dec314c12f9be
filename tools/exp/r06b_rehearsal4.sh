# four ranks sharing the one card through the driver's launcher: the default line and --config 3 (second session, final sources)
export MVS_BENCH_REHEARSAL=1
for cfg in "" "--config 3"; do
  echo "== torch.distributed.run x4 $cfg"
  timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 4 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 4 --steps 3 --warmup 1 --strong-steps 2 --no-cpu-baseline $cfg > gpurun_out/reh4b.json 2> gpurun_out/reh4b.err || { tail -5 gpurun_out/reh4b.err; exit 1; }
  python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/reh4b.json") if x.startswith("{")][-1])
print(d["n_gpus"], d["scaling"], "%.3g" % d["value"], "ms/step %.1f" % d["ms_per_step"], d["config"].get("kept_cells"), d["config"].get("cells_checksum"), d["config"]["collectives"], d["config"].get("collectives_note"))
for k, v in d.get("strong", {}).items():
    print("  ", k, v if not isinstance(v, dict) else (v.get("ms_per_step"), v.get("kept_cells"), v.get("cells_checksum")))
PY
done
