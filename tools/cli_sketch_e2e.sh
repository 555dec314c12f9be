#!/bin/bash
# End-to-end timing of `project_everything sketch`: first run parses the hash text (and leaves <file>.csr), the second
# maps the binary cache.   bash tools/cli_sketch_e2e.sh [samples] [hashes] [dimension]
N=${1:-10000}; H=${2:-50000}; D=${3:-2048}
BIN=$(pwd)/metagenome_vector_sketches_amd/bin
W=${TMPDIR:-/tmp}/mvs_sketch_e2e_$$
mkdir -p $W
/usr/bin/time -f "generate text: %e s" $BIN/mvs_make_hashes $W/h.txt $N $H 1234
ls -l $W/h.txt | awk '{print "text bytes", $5}'
for run in parse cache cache; do
  /usr/bin/time -f "sketch ($run): %e s wall" $BIN/project_everything sketch $W/h.txt $W/db_$run -d $D > $W/out_$run.log 2> $W/err_$run.log
  tail -1 $W/err_$run.log; grep "Time to compute" $W/out_$run.log
  sha256sum $W/db_$run/vectors.bin $W/db_$run/vector_norms.txt | awk '{print substr($1,1,16)}' | tr '\n' ' '; echo
done
ls -l $W/h.txt.csr | awk '{print "cache bytes", $5}'
rm -rf $W
