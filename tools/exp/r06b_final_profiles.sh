# rocprofv3 stats + counters of the configs[1] step / configs[2] two-stage / one-rank strong step / rank 1 of 8 through the C++ host,
# on the final sources of round 6 (second session); tools/publish_profiles.sh r06 copies the summaries into profiles/
set -x
OUT=gpurun_out/r06
mkdir -p $OUT
python3 tools/make_synth_db.py 100000 2048 2345 /tmp/mvs_r06_db_100000_2048/ > $OUT/db.out 2>&1 || exit 1
bash tools/collect_profiles.sh r06 "c1 c2 c3s g8" > $OUT/collect.log 2>&1 || { tail -20 $OUT/collect.log; exit 1; }
tail -3 $OUT/collect.log
