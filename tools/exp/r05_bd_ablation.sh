# what each stream of the direct-B filter's k-loop costs beside the MFMAs (ablation build: results are garbage)
export MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so
for dbg in 0 16 32 64 48 112; do
  echo "== pairwise_debug $dbg (16: no A fragment reads, 32: no A copies, 64: no B loads)"
  MVS_PAIRWISE_DEBUG=$dbg python tools/run_pairwise.py 100000 2048 4 2>&1 | tail -2
done
echo "== zero data"
ZERO_DATA=1 MVS_PAIRWISE_DEBUG=0 python tools/run_pairwise.py 100000 2048 4 2>&1 | tail -2
ZERO_DATA=1 MVS_PAIRWISE_DEBUG=16 python tools/run_pairwise.py 100000 2048 4 2>&1 | tail -2
