echo "== production library"
python tools/run_pairwise.py 100000 2048 4 2>&1 | tail -4
echo "== ablation library, debug 0"
MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so python tools/run_pairwise.py 100000 2048 4 2>&1 | tail -4
echo "== ablation library, debug 112, bdirect 0 / 1"
MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so MVS_PAIRWISE_DEBUG=112 MVS_PAIRWISE_BDIRECT=0 python tools/run_pairwise.py 100000 2048 3 2>&1 | tail -3
MVS_HIP_LIBRARY=$PWD/metagenome_vector_sketches_amd/libmvs_hip_abl.so MVS_PAIRWISE_DEBUG=112 python -c "
import metagenome_vector_sketches_amd as p
c=p.Context(0); print('debug option', c.get_option('pairwise_debug'), p._capi.LIB_PATH)"
