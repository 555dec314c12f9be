export MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so
for dbg in 0 1 2 3; do
  echo "== pp filter, debug $dbg"; MVS_PAIRWISE_DEBUG=$dbg python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -1
done
