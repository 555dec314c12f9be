set -x
python bench.py --config 3 --gpus 1 --steps 20 --warmup 5 > gpurun_out/r5_c3_g1.json 2> gpurun_out/r5_c3_g1.err || { tail -20 gpurun_out/r5_c3_g1.err; exit 1; }
python - <<PY
import json
l=[x for x in open("gpurun_out/r5_c3_g1.json") if x.startswith("{")][-1]
d=json.loads(l)
print("c3_g1", d["value"], d["ms_per_step"], d["config"].get("kept_cells"), d["config"]["cells_checksum"], json.dumps(d["stages"]), d["roofline"]["frac"])
print(d["timeline"])
PY
python bench.py --steps 20 --warmup 5 > gpurun_out/r5_bench.json 2> gpurun_out/r5_bench.err || { tail -20 gpurun_out/r5_bench.err; exit 1; }
python - <<PY
import json
d=json.loads([x for x in open("gpurun_out/r5_bench.json") if x.startswith("{")][-1])
print("value", d["value"], d["ms_per_step"], d["stages"])
print("step roofline", d["roofline_pairwise_step"])
print("pairwise", d["pairwise"]["two_stage"], d["roofline_pairwise"]["frac"], d["roofline_pairwise"]["exact_kernel"]["frac"], d["roofline_pairwise"]["algorithmic_credit"]["ratio_to_peak"])
for k,v in d["strong"].items(): print(k, v["ms_per_step"], v["cells_per_s"], json.dumps(v["stages"]), v["roofline"]["frac"])
for k,v in d["search"]["queries"].items(): print("search", k, v["wall_ms"], v["wall_ms_resident_queries"], v["two_stage_candidates"])
PY
