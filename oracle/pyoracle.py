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


def search_scores(db_vectors, db_norms_text, query_sketch, d, j):
    """Query-by-hashes search as src/jaccard.py scores it (numpy restatement, float64 throughout; the reference
    holds queries, index and inner products in float32, hence the 1e-5 tolerance of the test that uses this).
    db_vectors: int sketches [N, d] as stored in vectors.bin; db_norms_text: the norms parsed from vector_norms.txt
    (:192 `float(line.split()[1])`); query_sketch: int sketch [d] of the query (standalone_projection's output, :98-118).
      :117      query vector = sketch / sqrt(dimension)
      :123-124  query_norm = ||query||, then the query is L2-normalised
      :18-61    the index holds the L2-normalised database vectors (vector / sqrt(d), normalize_L2, IndexFlatIP)
      :199      jaccard = ip*qn*nn / (nn^2 + qn^2 - ip*qn*nn), reported when > j (:200), best first
      :184      a query with norm 0 reports nothing
    Returns [(database index, jaccard)] sorted by descending jaccard (ties: ascending index)."""
    v = np.asarray(query_sketch, dtype=np.float64) / np.sqrt(float(d))                    # :117
    qn = float(np.linalg.norm(v))                                                         # :123
    if qn == 0.0:                                                                         # :184
        return []
    x = np.asarray(db_vectors, dtype=np.float64) / np.sqrt(float(d))
    xn = np.linalg.norm(x, axis=1)
    with np.errstate(invalid="ignore", divide="ignore"):
        ip = (x @ v) / (xn * qn)                                                          # normalised inner products
        nn = np.asarray(db_norms_text, dtype=np.float64)
        jac = ip * qn * nn / (nn ** 2 + qn ** 2 - ip * qn * nn)                           # :199
    order = np.argsort(-np.where(np.isnan(jac), -np.inf, jac), kind="stable")
    return [(int(k), float(jac[k])) for k in order if jac[k] > j]                         # :200
