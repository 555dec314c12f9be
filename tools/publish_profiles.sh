#!/bin/bash
# Copy the summaries of gpurun_out/<tag>/ (made by tools/collect_profiles.sh) into profiles/<tag>_*:
#   bash tools/publish_profiles.sh r03
set -e
TAG=${1:-r05}
SRC=gpurun_out/$TAG
for w in c1 c2 c2x c2d srch c3s g8; do
    [ -f $SRC/${w}_kernel_stats.txt ] || continue
    grep "^k_" $SRC/${w}_kernel_stats.txt > profiles/${TAG}_${w}_kernel_stats.txt
    cp $SRC/${w}_pmc_summary.txt profiles/${TAG}_${w}_pmc_summary.txt
    cp "$(find $SRC/${w}_stats -name '*kernel_stats.csv' | head -1)" profiles/${TAG}_${w}_rocprof_kernel_stats.csv
    cp $SRC/${w}_derived.txt profiles/${TAG}_${w}_derived.txt
done
cp $SRC/pmc_traffic.json profiles/${TAG}_pmc_traffic.json
