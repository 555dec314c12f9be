python tools/exp/pp_check.py 100000 2048 8,9,10,8,9,10 "" 5
export MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so
for v in 31 32 33; do
  echo "== filter_variant $v"; MVS_FILTER_VARIANT=$v python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -1
done
