#!/bin/bash
# The strong-scaled step under the drop-in executable on configs[2] (100k x 2048, seed 2345: bench.py --config 3's sketches):
# shard folders of 8 shards from ONE rank, from 8 ranks of one process (threads, file transport: they share the card) and of 4
# shards from 4 processes, against the round-1 scheme (MVS_STEP=0); the ranks' checksums summed = bench.py's cells_checksum.
#   tools/exp/r06_step_rehearsal.sh OUTDIR [N] [d] [seed]
set -e -o pipefail
OUT=${1:-gpurun_out/r06b}; N=${2:-100000}; D=${3:-2048}; SEED=${4:-2345}
mkdir -p "$OUT"
B=metagenome_vector_sketches_amd/bin
DB=/tmp/mvs_r06_db_${N}_${D}/
W=/tmp/mvs_r06_idx_${N}_${D}
rm -rf "$W"; mkdir -p "$W"
python tools/make_synth_db.py $N $D $SEED $DB > "$OUT/make_db.log" 2>&1
echo "db ready" | tee -a "$OUT/progress.log"
run() { # tag, shards, env...
  tag=$1; shards=$2; shift 2
  env "$@" MVS_STAGE_TIMING=1 MVS_STEP_CHECKSUM=1 $B/pairwise_comp_optimized --db $DB --max_memory_gb 12 --num_threads 8 \
      --output_folder $W/$tag --num_shards $shards --shard_idx -1 > "$OUT/$tag.stdout" 2> "$OUT/$tag.stderr"
  echo "$tag done" | tee -a "$OUT/progress.log"
}
run legacy8 8 MVS_STEP=0 MVS_PAIRWISE_CONTEXTS=1
run step8_1rank 8 MVS_PAIRWISE_CONTEXTS=1
run step8_8ranks 8 MVS_PAIRWISE_CONTEXTS=8
run step8_4ranks 8 MVS_PAIRWISE_CONTEXTS=4
run step8_2ranks 8 MVS_PAIRWISE_CONTEXTS=2
run legacy4 4 MVS_STEP=0 MVS_PAIRWISE_CONTEXTS=1
# 4 processes, one shard each, file transport (the GPU box allows 6 processes on its card)
pids=""
for k in 0 1 2 3; do
  MVS_COLLECTIVE=files MVS_COLLECTIVE_TOKEN=r06 MVS_DEVICE=0 MVS_STAGE_TIMING=1 MVS_STEP_CHECKSUM=1 $B/pairwise_comp_optimized --db $DB \
      --max_memory_gb 12 --num_threads 8 --output_folder $W/procs4 --num_shards 4 --shard_idx $k > "$OUT/procs4_$k.stdout" 2> "$OUT/procs4_$k.stderr" &
  pids="$pids $!"
done
for p in $pids; do wait $p; done
echo "procs4 done" | tee -a "$OUT/progress.log"
{
  for t in step8_1rank step8_8ranks step8_4ranks step8_2ranks; do
    ok=1
    for k in 0 1 2 3 4 5 6 7; do for f in matrix.bin row_index.bin neighbor_start.bin; do
      cmp -s $W/legacy8/shard_$k/$f $W/$t/shard_$k/$f || ok=0
    done; done
    echo "$t vs legacy8 (round-1 scheme): byte-identical=$ok"
  done
  ok=1
  for k in 0 1 2 3; do for f in matrix.bin row_index.bin neighbor_start.bin; do
    cmp -s $W/legacy4/shard_$k/$f $W/procs4/shard_$k/$f || ok=0
  done; done
  echo "procs4 vs legacy4 (round-1 scheme): byte-identical=$ok"
  du -sb $W/legacy8 | awk '{print "bytes of the 8 shard folders: "$1}'
  python3 - "$OUT" <<'PY'
import glob, re, sys
out = sys.argv[1]
for tag in ("step8_1rank", "step8_8ranks", "step8_4ranks", "step8_2ranks", "procs4"):
    s1 = s2 = kept = 0
    files = [out + "/" + tag + ".stderr"] if tag != "procs4" else sorted(glob.glob(out + "/procs4_*.stderr"))
    for f in files:
        for m in re.finditer(r"\[checksum\] rank \d+ kept (\d+) sum ([0-9a-f]+) sum2 ([0-9a-f]+)", open(f).read()):
            kept += int(m.group(1)); s1 += int(m.group(2), 16); s2 += int(m.group(3), 16)
    print("%s: kept %d cells_checksum %016x%016x" % (tag, kept, s1 % 2**64, s2 % 2**64))
PY
} | tee "$OUT/summary.txt"
# every rank's step of the G-way split timed alone on the card, exchange bytes in place (the C++ host): five runs.  60 warm-up
# steps per rank: the card's power controller needs ~0.1 s to settle after the set-up phase (with 5 warm-up steps single
# ranks of single runs measured 0.1 ms -- 7 % -- slower than the same rank in the next run)
for rep in 1 2 3 4 5; do
  for G in 1 2 4 8; do
    $B/mvs_step_bench --db $DB --ranks $G --steps ${STEPS:-60} --warmup ${WARMUP:-60} > "$OUT/step_bench_G${G}_run${rep}.json" 2> "$OUT/step_bench_G${G}_run${rep}.stderr"
  done
  echo "step_bench run $rep done" | tee -a "$OUT/progress.log"
done
grep -h "\[step\]\|\[stage\]" "$OUT"/step8_8ranks.stderr | head -40 > "$OUT/step8_8ranks_spans.txt" || true
rm -rf "$W"
[ -n "$KEEP_DB" ] || rm -rf "$DB"
