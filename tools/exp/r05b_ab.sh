set -x
mkdir -p gpurun_out/g8
timeout -k 10 300 python -m pytest tests/test_plan_gpu.py -x -q 2>&1 | tail -2 || exit 1
for spin in 0 3000; do
timeout -k 10 200 python3 tools/strong_model.py 100000 2048 --ranks 1,8 --reps 10 --link-GBps 61 --report-spin $spin > gpurun_out/g8/model_spin$spin.out 2>&1; grep -v amdgpu.ids gpurun_out/g8/model_spin$spin.out
done
timeout -k 10 200 python bench.py --config 3 --gpus 1 --steps 20 --warmup 5 > gpurun_out/g8/c3_g1.json 2> gpurun_out/g8/c3_g1.err
python3 -c "
import json
d=json.loads([x for x in open('gpurun_out/g8/c3_g1.json') if x.startswith('{')][-1])
print('c3_g1', d['ms_per_step'], d['config']['kept_cells'], d['config']['cells_checksum']); print(json.dumps(d['stages']))
"
