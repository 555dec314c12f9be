#!/bin/bash
# mvs_step_bench at G = 8 with and without polling in mvs_cells_report (option report_spin): is the +0.1 ms some steps take the
# host's wake-up from a blocking stream synchronisation?
set -e -o pipefail
OUT=${1:-gpurun_out/r06e}; mkdir -p $OUT
B=metagenome_vector_sketches_amd/bin
DB=/tmp/mvs_r06_db_100000_2048/
[ -f $DB/vectors.bin ] || python tools/make_synth_db.py 100000 2048 2345 $DB > $OUT/make_db.log 2>&1
for rep in 1 2 3; do
  for spin in 0 300 2000; do
    $B/mvs_step_bench --db $DB --ranks 8 --steps 30 --warmup 5 --report-spin $spin > $OUT/spin${spin}_run${rep}.json 2> $OUT/spin${spin}_run${rep}.stderr
  done
done
python3 - "$OUT" <<'PY'
import json, glob, sys
out = sys.argv[1]
for spin in (0, 300, 2000):
    for f in sorted(glob.glob(out + "/spin%d_run*.json" % spin)):
        r = json.load(open(f))
        print("report_spin %4d: slowest %.3f  medians %s  means %s" % (spin, r["slowest_rank_wall_ms_median"],
              " ".join("%.3f" % p["wall_ms_median"] for p in r["per_rank"]), " ".join("%.3f" % p["wall_ms"] for p in r["per_rank"])))
PY
