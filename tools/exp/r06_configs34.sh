#!/bin/bash
# configs[3] (100k x 4096) and configs[4] (1M x 2048) sizes through the drop-in executable's strong-scaled step on ONE card
# (`--num_shards 8 --shard_idx -1`), checksums against bench.py's `--config 4` / `5`, and mvs_step_bench on configs[3].
set -e -o pipefail
OUT=${1:-gpurun_out/r06p}; mkdir -p $OUT
B=metagenome_vector_sketches_amd/bin
W=/tmp/mvs_r06_idx34; rm -rf $W; mkdir -p $W
sumcheck() { # tag
python3 - "$OUT" "$1" <<'PY'
import re, sys
out, tag = sys.argv[1], sys.argv[2]
s1 = s2 = kept = 0
for m in re.finditer(r"\[checksum\] rank \d+ kept (\d+) sum ([0-9a-f]+) sum2 ([0-9a-f]+)", open(out + "/" + tag + ".stderr").read()):
    kept += int(m.group(1)); s1 += int(m.group(2), 16); s2 += int(m.group(3), 16)
print("%s: kept %d cells_checksum %016x%016x" % (tag, kept, s1 % 2**64, s2 % 2**64))
PY
}
DB4=/tmp/mvs_r06_db_100000_4096/
python tools/make_synth_db.py 100000 4096 3456 $DB4 > $OUT/make_db4.log 2>&1
echo "db4 ready" | tee -a $OUT/progress.log
for ranks in 1 8; do
  MVS_PAIRWISE_CONTEXTS=$ranks MVS_STAGE_TIMING=1 MVS_STEP_CHECKSUM=1 $B/pairwise_comp_optimized --db $DB4 --max_memory_gb 12 --num_threads 8 \
      --output_folder $W/c3_r$ranks --num_shards 8 --shard_idx -1 > $OUT/c3_r$ranks.stdout 2> $OUT/c3_r$ranks.stderr
  sumcheck c3_r$ranks | tee -a $OUT/summary.txt
done
ok=1; for k in 0 1 2 3 4 5 6 7; do for f in matrix.bin row_index.bin neighbor_start.bin; do cmp -s $W/c3_r1/shard_$k/$f $W/c3_r8/shard_$k/$f || ok=0; done; done
echo "configs[3]: 8 shards from 1 rank vs from 8 ranks: byte-identical=$ok" | tee -a $OUT/summary.txt
for rep in 1 2 3; do for G in 1 8; do
  $B/mvs_step_bench --db $DB4 --ranks $G --steps 40 --warmup 40 > $OUT/step_bench_c3_G${G}_run${rep}.json 2> $OUT/step_bench_c3_G${G}_run${rep}.stderr
done; done
echo "step bench c3 done" | tee -a $OUT/progress.log
rm -rf $DB4 $W/c3_r1 $W/c3_r8
DB5=/tmp/mvs_r06_db_1000000_2048/
python tools/make_synth_db.py 1000000 2048 4567 $DB5 > $OUT/make_db5.log 2>&1
echo "db5 ready" | tee -a $OUT/progress.log
MVS_PAIRWISE_CONTEXTS=1 MVS_STAGE_TIMING=1 MVS_STEP_CHECKSUM=1 $B/pairwise_comp_optimized --db $DB5 --max_memory_gb 12 --num_threads 8 \
    --output_folder $W/c4_r1 --num_shards 8 --shard_idx -1 > $OUT/c4_r1.stdout 2> $OUT/c4_r1.stderr
sumcheck c4_r1 | tee -a $OUT/summary.txt
du -sb $W/c4_r1 | awk '{print "configs[4]: bytes of the 8 shard folders: "$1}' | tee -a $OUT/summary.txt
grep -h "\[step\]\|\[stage\]\|Total computation" $OUT/c4_r1.stderr $OUT/c4_r1.stdout > $OUT/c4_r1_spans.txt || true
rm -rf $DB5 $W
