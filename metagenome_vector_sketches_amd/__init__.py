"""metagenome_vector_sketches_amd -- MI355X (gfx950) implementation of the random-projection sketching +
all-vs-all sketch comparison hot path of RolandFaure/metagenome_vector_sketches.

Layout
  csrc/         HIP kernels (mvs_project.hip, mvs_pairwise.hip) + the C ABI (mvs_capi.hip -> libmvs_hip.so,
                declared in include/mvs_hip.h) and, under csrc/host/, the C++ host side that keeps the
                reference's command lines and file formats (project_everything, standalone_projection,
                pairwise_comp_optimized, query_pc_mat, the pc_mat:: reader, the read_pc_mat_module binding)
  bin/          the built executables (git-ignored)
  _capi.py      ctypes binding of the C ABI (Context, SketchSet)
  parallel.py   row-sharded comparison across the GPUs of a node (torch.distributed all-gather of limb planes)
  search.py     query-by-hashes search over a DB folder (counterpart of the reference's FAISS path)
  synth.py      synthetic FracMinHash-like inputs for tests and bench

Importing the package does not load the HIP library; the first call that needs it does, and fails
loudly if it is missing (there is no CPU fallback).
"""
from . import _capi  # noqa: F401
from ._capi import Context, MvsError, load_library  # noqa: F401

__all__ = ["Context", "MvsError", "load_library"]
