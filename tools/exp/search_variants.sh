#!/bin/bash
# search_bench for several query counts (streaming filter by size up to 512 rows, by number beyond)
for nq in 96 160 192 320 384 448 512 576 640; do
    echo "== nq $nq"
    MVS_FILTER_VARIANT=50 python tools/search_bench.py 1000000 2048 $nq 8 2>&1 | tail -2 | head -1 | cut -c1-60,140-260
done
