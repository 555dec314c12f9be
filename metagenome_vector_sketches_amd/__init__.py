"""metagenome_vector_sketches_amd -- MI355X (gfx950) implementation of the random-projection sketching +
all-vs-all sketch comparison hot path of RolandFaure/metagenome_vector_sketches.

Layout
  csrc/         HIP kernels + the C ABI (libmvs_hip.so, declared in include/mvs_hip.h) and the C++ host
                drivers that keep the reference's command lines
  _capi.py      ctypes binding of the C ABI
  sketch.py     host-side mirror of the reference's sketch interface (transform_set_into_vector,
                sketch(), standalone projection protocol)
  pairwise.py   host-side mirror of pairwise_comp_optimized (DB reader, shard loop, kept-cell lists)
  synth.py      synthetic FracMinHash-like inputs for tests and bench

Importing the package does not load the HIP library; the first call that needs it does, and fails
loudly if it is missing (there is no CPU fallback).
"""
from . import _capi  # noqa: F401
from ._capi import Context, MvsError, load_library  # noqa: F401

__all__ = ["Context", "MvsError", "load_library"]
