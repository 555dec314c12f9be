# two ranks sharing the one card through the launcher the driver uses: (1) the library's communicator (file transport), (2) rank 1 finds no
# RCCL (test hook): every rank moves to torch.distributed's collectives and the line says so, (3) --require-native-collectives: exit 3
export MVS_BENCH_REHEARSAL=1
run() {
  timeout -k 10 240 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $1 bench.py --gpus 2 --steps 5 --warmup 2 --strong-steps 3 $2 > gpurun_out/fb_$3.json 2> gpurun_out/fb_$3.err
  echo "== $3: exit $?"
  python - "$3" <<PY
import json, sys
ls=[x for x in open("gpurun_out/fb_%s.json" % sys.argv[1]) if x.startswith("{")]
if not ls:
    print("no line"); sys.exit(0)
d=json.loads(ls[-1])
print("%.4g" % d["value"], "ms/step %.2f" % d["ms_per_step"], d["config"]["collectives"], "|", d["config"]["collectives_note"], "| kept", d["config"]["kept_cells"])
for k, v in d.get("strong", {}).items():
    print("  ", k, v if not isinstance(v, dict) else (v.get("ms_per_step"), v.get("kept_cells"), v.get("cells_checksum")))
PY
}
run 29531 "" native
export MVS_BENCH_FAIL_NATIVE_COMM=1
run 29532 "" fallback
run 29533 "--require-native-collectives" required
tail -3 gpurun_out/fb_required.err
