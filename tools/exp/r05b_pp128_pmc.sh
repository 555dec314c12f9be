# counters of the pp128 k-loop benchmark (tools/microbench/pp128.hip): matrix-pipe busy cycles and the clock, coarse-plane-like
# operands against all-zero operands; the program itself after `--`, counters in their own passes
REPO=$(pwd)
OUT=$REPO/gpurun_out/pp128
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in 0 1 6; do
  for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VALU_MFMA_I8 SQ_BUSY_CYCLES GRBM_GUI_ACTIVE"; do
    tag=$(echo $set | cut -c1-12 | tr ' ' '_')
    timeout -k 5 120 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/m${m}_$tag -- $REPO/tools/microbench/pp128 277 2048 2 $m > $OUT/m${m}_$tag.out 2>&1 || { tail -3 $OUT/m${m}_$tag.out; exit 1; }
  done
done
cd $REPO
python3 - <<'PY'
import csv, glob, os
from collections import defaultdict
for m in (0, 1, 6):
    vals = defaultdict(list)
    dur = []
    for f in glob.glob("gpurun_out/pp128/m%d_*/**/*counter_collection.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pp128ILb1" in r["Kernel_Name"] or "k_pp128<true>" in r["Kernel_Name"]:
                vals[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob("gpurun_out/pp128/m%d_*/**/*kernel_trace.csv" % m, recursive=True):
        for r in csv.DictReader(open(f)):
            if "k_pp128ILb1" in r["Kernel_Name"] or "k_pp128<true>" in r["Kernel_Name"]:
                dur.append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    mean = {k: sum(v) / len(v) for k, v in vals.items()}
    d = sum(dur) / max(len(dur), 1)
    gui = mean.get("GRBM_GUI_ACTIVE", 0) / 8.0            # the counter is summed over the 8 XCDs
    print("mode %d: %.3f ms per launch (profiled), clock %.2f GHz, MFMA busy %.3f of SIMD cycles, %s" % (
        m, d, gui / (d * 1e6) if d else 0, mean.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (gui * 1024) if gui else 0,
        ", ".join("%s %.4g" % (k, v) for k, v in sorted(mean.items()))))
PY
rm -rf gpurun_out/pp128/m*_*/
