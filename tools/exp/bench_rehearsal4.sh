# four ranks sharing the one card, started by bench.py itself (no launcher): every bench mode once
export MVS_BENCH_REHEARSAL=1
for cfg in "" "--config 3" "--config 4" "--config 5"; do
  echo "== --gpus 4 $cfg"
  timeout -k 10 500 python bench.py --gpus 4 --steps 2 --warmup 1 --no-cpu-baseline $cfg > gpurun_out/reh4.json 2> gpurun_out/reh4.err || { tail -5 gpurun_out/reh4.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/reh4.json").read().strip().split("\n")[-1])
print(d["n_gpus"], d["scaling"], "%.3g" % d["value"], d["unit"][:20], "ms/step %.1f" % d["ms_per_step"], d["config"].get("kept_cells"), d["config"].get("schedule"), d["config"]["collectives"], d["config"].get("overlap"), d["stages"])
PY
done
