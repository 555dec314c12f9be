export MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so
for v in 1 21 22 23 0 11 12 13; do
  echo "== filter_variant $v"; MVS_FILTER_VARIANT=$v python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -1
done
echo "== debug 1 (no k-loop)"; MVS_PAIRWISE_DEBUG=1 python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -1
echo "== debug 2 (no epilogue)"; MVS_PAIRWISE_DEBUG=2 python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -1
cat /sys/fs/cgroup/cpu.max 2>/dev/null; cat /sys/fs/cgroup/cpu/cpu.cfs_quota_us 2>/dev/null; nproc
