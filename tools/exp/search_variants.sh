#!/bin/bash
# search_bench for several query counts: streaming filter from the fragment-major plane (default) / the row-major plane
for nq in 1 16 64 256 512 640 700 1023; do
  for fm in 1 0; do
    echo "== nq $nq MVS_FRAGMENT_MAJOR $fm"
    MVS_FRAGMENT_MAJOR=$fm python tools/search_bench.py 1000000 2048 $nq 8 2>&1 | tail -2 | cut -c1-60,140-260
  done
done
