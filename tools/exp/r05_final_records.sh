# round 5: the records DESIGN.md / BASELINE.md quote, on the final sources (one gpurun call)
set -x
python bench.py --steps 20 --warmup 5 > gpurun_out/r05_bench.json 2> gpurun_out/r05_bench.err || { tail -20 gpurun_out/r05_bench.err; exit 1; }
: > gpurun_out/r05_bench_configs_1gpu.jsonl
for c in 3 4 5; do
  python bench.py --config $c --gpus 1 --steps 10 --warmup 3 >> gpurun_out/r05_bench_configs_1gpu.jsonl 2>> gpurun_out/r05_bench_configs.err || { tail -20 gpurun_out/r05_bench_configs.err; exit 1; }
done
python tools/strong_model.py 100000 2048 --out gpurun_out/r05_strong_model_configs2.json > gpurun_out/r05_strong_model.log 2>&1
python tools/strong_model.py 100000 4096 --seed 3456 --out gpurun_out/r05_strong_model_configs3.json >> gpurun_out/r05_strong_model.log 2>&1
export MVS_BENCH_REHEARSAL=1
for g in 2 4; do
  timeout -k 10 300 python bench.py --gpus $g --config 3 --steps 3 --warmup 1 > gpurun_out/r05_rehearsal_c3_r$g.json 2> gpurun_out/r05_rehearsal_c3_r$g.err || { tail -20 gpurun_out/r05_rehearsal_c3_r$g.err; exit 1; }
done
timeout -k 10 400 python bench.py --gpus 2 --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r05_rehearsal_default_r2.json 2> gpurun_out/r05_rehearsal_default_r2.err || { tail -20 gpurun_out/r05_rehearsal_default_r2.err; exit 1; }
echo records done
