"""ctypes binding of oracle/libmvs_oracle.so (the C restatement, oracle/mvs_oracle.c).

TEST INFRASTRUCTURE ONLY -- see oracle/mvs_oracle.h.  Nothing under metagenome_vector_sketches_amd/
imports this module.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c = ctypes
_P = _c.c_void_p

CELL_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("dot", "<i4"), ("q", "<i4")])

_libs = {}


def build(native=False):
    target = "native" if native else "all"
    subprocess.run(["make", "-s", "-C", _HERE, target], check=True)


def load(native=False):
    """native=True: the -march=native build (bench cpu_baseline on the GPU box); falls back to the
    portable build when it cannot be compiled there."""
    key = bool(native)
    if key in _libs:
        return _libs[key]
    name = "libmvs_oracle_native.so" if native else "libmvs_oracle.so"
    path = os.path.join(_HERE, name)
    if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(os.path.join(_HERE, "mvs_oracle.c")):
        try:
            build(native)
        except Exception:
            if native:
                return load(False)
            if not os.path.exists(path):
                raise
    lib = ctypes.CDLL(path)
    sig = {
        "mvs_oracle_splitmix64": (_c.c_uint64, [_c.c_uint64]),
        "mvs_oracle_project": (None, [_P, _c.c_int64, _c.c_int, _P]),
        "mvs_oracle_project_csr": (None, [_P, _P, _c.c_int64, _c.c_int, _P, _c.c_int]),
        "mvs_oracle_project_csr_fast": (None, [_P, _P, _c.c_int64, _c.c_int, _P, _c.c_int]),
        "mvs_oracle_sumsq": (_c.c_int64, [_P, _c.c_int]),
        "mvs_oracle_norm_f32path": (_c.c_double, [_P, _c.c_int]),
        "mvs_oracle_norm": (_c.c_double, [_P, _c.c_int]),
        "mvs_oracle_format_norm": (_c.c_int, [_c.c_double, _c.c_char_p, _c.c_int]),
        "mvs_oracle_norm_sq_from_text": (_c.c_double, [_c.c_char_p]),
        "mvs_oracle_saturate_i16": (None, [_P, _c.c_int64, _P]),
        "mvs_oracle_dot_i32": (_c.c_int32, [_P, _P, _c.c_int]),
        "mvs_oracle_dot_i16": (_c.c_int32, [_P, _P, _c.c_int]),
        "mvs_oracle_keep_i32": (_c.c_int, [_c.c_int32, _c.c_int, _c.c_double, _c.c_double]),
        "mvs_oracle_keep_i16": (_c.c_int, [_c.c_int32, _c.c_int, _c.c_double, _c.c_double]),
        "mvs_oracle_quantize": (_c.c_int32, [_c.c_int32, _c.c_int, _c.c_double, _c.c_double]),
        "mvs_oracle_pairwise_rows": (_c.c_int64, [_P, _c.c_int, _c.c_int64, _c.c_int, _P, _c.c_int64, _c.c_int64,
                                                  _c.c_int64, _P, _c.c_int64, _c.c_int]),
        "mvs_oracle_chunk_size": (_c.c_int64, [_c.c_double, _c.c_int]),
        "mvs_oracle_shard_rows": (None, [_c.c_int64, _c.c_int, _c.c_int, _c.POINTER(_c.c_int64),
                                         _c.POINTER(_c.c_int64)]),
        "mvs_oracle_dots_dense": (None, [_P, _c.c_int64, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64,
                                         _P, _c.c_int]),
        "mvs_oracle_max_threads": (_c.c_int, []),
    }
    for name, (res, args) in sig.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    _libs[key] = lib
    return lib


def splitmix64(x):
    return load().mvs_oracle_splitmix64(_c.c_uint64(x & (2**64 - 1)))


def project(hashes, d):
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    out = np.empty(d, dtype=np.int32)
    load().mvs_oracle_project(h.ctypes.data, len(h), d, out.ctypes.data)
    return out


def project_csr(hashes, offsets, d, threads=0, fast=False, native=False):
    h = np.ascontiguousarray(hashes, dtype=np.uint64)
    o = np.ascontiguousarray(offsets, dtype=np.int64)
    n = len(o) - 1
    out = np.empty((n, d), dtype=np.int32)
    lib = load(native)
    fn = lib.mvs_oracle_project_csr_fast if fast else lib.mvs_oracle_project_csr
    fn(h.ctypes.data, o.ctypes.data, n, d, out.ctypes.data, threads)
    return out


def sumsq(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return load().mvs_oracle_sumsq(v.ctypes.data, len(v))


def norm(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return load().mvs_oracle_norm(v.ctypes.data, len(v))


def norm_f32path(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    return load().mvs_oracle_norm_f32path(v.ctypes.data, len(v))


def format_norm(x):
    buf = ctypes.create_string_buffer(64)
    load().mvs_oracle_format_norm(float(x), buf, 64)
    return buf.value.decode()


def norm_sq_from_text(text):
    return load().mvs_oracle_norm_sq_from_text(text.encode())


def saturate_i16(v):
    v = np.ascontiguousarray(v, dtype=np.int32)
    out = np.empty(v.shape, dtype=np.int16)
    load().mvs_oracle_saturate_i16(v.ctypes.data, v.size, out.ctypes.data)
    return out


def pairwise_rows(sketches, norms_sq, row_begin=0, row_end=None, chunk=192, threads=0, native=False):
    sk = np.ascontiguousarray(sketches)
    assert sk.dtype in (np.int32, np.int16)
    n, d = sk.shape
    if row_end is None:
        row_end = n
    n2 = np.ascontiguousarray(norms_sq, dtype=np.float64)
    lib = load(native)
    cap = max(1024, 64 * (row_end - row_begin))
    while True:
        cells = np.empty(cap, dtype=CELL_DTYPE)
        cnt = lib.mvs_oracle_pairwise_rows(sk.ctypes.data, sk.dtype.itemsize, n, d, n2.ctypes.data, row_begin,
                                           row_end, chunk, cells.ctypes.data, cap, threads)
        if cnt < 0:
            raise RuntimeError("oracle pairwise failed: %d" % cnt)
        if cnt > cap:
            cap = cnt
            continue
        return cells[:cnt]


def dots_dense(sketches, r0, r1, c0, c1, threads=0, native=False):
    sk = np.ascontiguousarray(sketches, dtype=np.int32)
    n, d = sk.shape
    out = np.empty((r1 - r0, c1 - c0), dtype=np.int32)
    load(native).mvs_oracle_dots_dense(sk.ctypes.data, n, d, r0, r1, c0, c1, out.ctypes.data, threads)
    return out


def chunk_size(max_memory_gb, d):
    return load().mvs_oracle_chunk_size(float(max_memory_gb), d)


def shard_rows(n, num_shards, shard_idx):
    b, e = _c.c_int64(), _c.c_int64()
    load().mvs_oracle_shard_rows(n, num_shards, shard_idx, ctypes.byref(b), ctypes.byref(e))
    return b.value, e.value


def max_threads():
    return load().mvs_oracle_max_threads()
