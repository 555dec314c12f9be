#!/bin/bash
# Collect the rocprofv3 evidence of a round on the GPU box (run through gpurun from the repository root):
#   bash tools/collect_profiles.sh r03 ["c1 c2 c2x c2d"]        (second argument: the workloads to collect, default all)
# -> gpurun_out/<tag>/{c1,c2,c2x}_{stats,pmc_*}/ + text/JSON summaries; copy what should be judged into profiles/.
# Workloads:  c1  = configs[1] step (bench.py without the configs[2] leg)
#             c2  = configs[2] pairwise, two-stage comparison (tools/run_pairwise.py 100000 2048)
#             c2x = configs[2] pairwise, exact kernel on every cell (MVS_PAIRWISE_FILTER=0)
#             c2d = 100k x 2048 with a 10 %-dense result streamed out (tools/stream_bench.py: two-stage comparison feeding the
#                   dense byte matrix, rows encoded on the device)
#             srch = 64 query sketches against 10^6 resident sketches (tools/search_bench.py: k_search_filter + re-check)
#             c3s = the strong-scaled pairwise step on one rank (bench.py --config 3 --gpus 1: k_recode_rows, the block plan's
#                   filter launch, re-check, flagged tiles, k_cells_route, the row-bucket sort)
#             g8  = ONE rank's step of an 8-way split of configs[2] through the C++ host (bin/mvs_step_bench --ranks 8 --rank 1,
#                   exchange bytes in place; needs the DB of tools/make_synth_db.py 100000 2048 2345 /tmp/mvs_r06_db_100000_2048/)
# The profiled program itself follows `--` (no env / sh wrapper); counters are collected in their own passes.
set -u
TAG=${1:-r05}
WORKLOADS=${2:-"c1 c2 c2x c2d srch c3s"}
want() { case " $WORKLOADS " in *" $1 "*) return 0;; *) return 1;; esac; }
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp

PMC_SETS=("FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"
          "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES"
          "TCC_HIT_sum TCC_MISS_sum")

run_set() {   # name, command...
    local name=$1; shift
    echo "[$name] kernel trace" ; date
    rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/${name}_stats" -- "$@" > "$OUT/${name}_stats.out" 2> "$OUT/${name}_stats.err" || return 1
    echo "$*" > "$OUT/${name}_stats/command.txt"
    local i=0
    for set in "${PMC_SETS[@]}"; do
        i=$((i + 1))
        echo "[$name] pmc pass $i: $set"
        rocprofv3 --pmc $set --kernel-trace --output-format csv -d "$OUT/${name}_pmc/p$i" -- "$@" > "$OUT/${name}_pmc_p$i.out" 2> "$OUT/${name}_pmc_p$i.err" || return 1
    done
    echo "$*" > "$OUT/${name}_pmc/command.txt"
}

if want c1; then run_set c1 python3 "$REPO/bench.py" --steps 20 --warmup 10 --no-cpu-baseline --pairwise-samples 0 --stream-samples 0 --search-samples 0 --density-samples 0 --strong-steps 0 || exit 1; fi
if want c2; then export MVS_PAIRWISE_FILTER=1; run_set c2 python3 "$REPO/tools/run_pairwise.py" 100000 2048 12 || exit 1; unset MVS_PAIRWISE_FILTER; fi
if want c2x; then export MVS_PAIRWISE_FILTER=0; run_set c2x python3 "$REPO/tools/run_pairwise.py" 100000 2048 6 || exit 1; unset MVS_PAIRWISE_FILTER; fi
if want c2d; then run_set c2d python3 "$REPO/tools/stream_bench.py" 100000 2048 10000 2 encoded || exit 1; fi
if want srch; then run_set srch python3 "$REPO/tools/search_bench.py" 1000000 2048 64 6 || exit 1; fi
if want c3s; then run_set c3s python3 "$REPO/bench.py" --config 3 --gpus 1 --steps 3 --warmup 2 || exit 1; fi
if want g8; then run_set g8 "$REPO/metagenome_vector_sketches_amd/bin/mvs_step_bench" --db /tmp/mvs_r06_db_100000_2048/ --ranks 8 --rank 1 --steps 10 --warmup 3 --probe 0 || exit 1; fi

cd "$REPO"
for w in c1 c2 c2x c2d srch c3s g8; do
    [ -d "$OUT/${w}_stats" ] || continue
    python3 tools/pmc_summary.py --stats "$OUT/${w}_stats" > "$OUT/${w}_kernel_stats.txt"
    python3 tools/pmc_summary.py "$OUT/${w}_pmc" > "$OUT/${w}_pmc_summary.txt"
    python3 tools/pmc_summary.py --derived "$OUT/${w}_pmc" "$OUT/${w}_stats" > "$OUT/${w}_derived.txt"
done
SPECS=""
[ -d "$OUT/c1_pmc" ] && SPECS="$SPECS configs[1]=$OUT/c1_pmc"
[ -d "$OUT/c2_pmc" ] && SPECS="$SPECS configs[2]=$OUT/c2_pmc"
[ -d "$OUT/c2x_pmc" ] && SPECS="$SPECS configs[2]-exact=$OUT/c2x_pmc"
[ -d "$OUT/c2d_pmc" ] && SPECS="$SPECS dense-100k=$OUT/c2d_pmc"
[ -d "$OUT/srch_pmc" ] && SPECS="$SPECS search-64x1M=$OUT/srch_pmc"
[ -d "$OUT/c3s_pmc" ] && SPECS="$SPECS strong-configs[2]-1rank=$OUT/c3s_pmc"
[ -d "$OUT/g8_pmc" ] && SPECS="$SPECS strong-configs[2]-rank1of8-cpp=$OUT/g8_pmc"
python3 tools/pmc_summary.py --traffic "$OUT/pmc_traffic.json" $SPECS
# keep the merged-back payload small: the raw per-dispatch CSVs of the PMC passes are summarised above
find "$OUT" -name "*_counter_collection.csv" -size +4M -delete
find "$OUT" -name "*_kernel_trace.csv" -size +4M -delete
echo done
