"""ctypes binding of libmvs_hip.so (include/mvs_hip.h).

The library is the product: if it is missing or cannot be loaded this module raises -- there is no
Python or CPU fallback for the hot path.
"""
import ctypes
import os
import weakref

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# MVS_HIP_LIBRARY: profiling tools point this at the ablation build (make -C csrc ablations); nothing else should
LIB_PATH = os.environ.get("MVS_HIP_LIBRARY") or os.path.join(_HERE, "libmvs_hip.so")

MVS_OK, MVS_E_INVALID, MVS_E_HIP, MVS_E_CAPACITY, MVS_E_NOMEM, MVS_E_RANGE, MVS_E_ABORTED = 0, 1, 2, 3, 4, 5, 6
MEM_HOST, MEM_DEVICE = 0, 1
KEEP_INT32, KEEP_INT16 = 0, 1
LIMBS_K3 = 0x103
BLOCK_SYMMETRIC, BLOCK_MIRROR_ALL = 1, 2

CELL_DTYPE = np.dtype([("row", "<i4"), ("col", "<i4"), ("dot", "<i4"), ("q", "<i4")])

# every symbol include/mvs_hip.h declares: (name, restype, argtypes)
_c = ctypes
_P = _c.c_void_p


class RowBlock(ctypes.Structure):
    """mvs_row_block: one CSR piece of the streamed comparison result"""
    _fields_ = [("row_begin", _c.c_int64), ("row_end", _c.c_int64), ("n_cells", _c.c_int64),
                ("row_ptr", _c.POINTER(_c.c_int64)), ("col", _c.POINTER(_c.c_int32)),
                ("q", _c.POINTER(_c.c_uint8)), ("q16", _c.POINTER(_c.c_uint16))]


ROW_BLOCK_CB = ctypes.CFUNCTYPE(_c.c_int, _P, _c.POINTER(RowBlock))


class EncodedRows(ctypes.Structure):
    """mvs_encoded_rows: one piece of the streamed comparison result with its rows encoded in the shard codec"""
    _fields_ = [("row_begin", _c.c_int64), ("row_end", _c.c_int64), ("n_cells", _c.c_int64), ("n_rows", _c.c_int64),
                ("rows", _c.POINTER(_c.c_uint32)), ("first_col", _c.POINTER(_c.c_uint32)),
                ("offset", _c.POINTER(_c.c_uint64)), ("jac_bytes", _c.POINTER(_c.c_uint32)),
                ("bytes", _c.POINTER(_c.c_uint8)), ("n_bytes", _c.c_int64)]


ENCODED_ROWS_CB = ctypes.CFUNCTYPE(_c.c_int, _P, _c.POINTER(EncodedRows))
SYMBOLS = [
    ("mvs_version", _c.c_char_p, []),
    ("mvs_last_error", _c.c_char_p, []),
    ("mvs_device_count", _c.c_int, [_c.POINTER(_c.c_int)]),
    ("mvs_ctx_create", _c.c_int, [_c.c_int, _c.POINTER(_P)]),
    ("mvs_ctx_destroy", _c.c_int, [_P]),
    ("mvs_ctx_set_option", _c.c_int, [_P, _c.c_char_p, _c.c_int64]),
    ("mvs_ctx_get_option", _c.c_int, [_P, _c.c_char_p, _c.POINTER(_c.c_int64)]),
    ("mvs_ctx_set_stream", _c.c_int, [_P, _P]),
    ("mvs_ctx_use_own_stream", _c.c_int, [_P]),
    ("mvs_ctx_synchronize", _c.c_int, [_P]),
    ("mvs_ctx_set_timing", _c.c_int, [_P, _c.c_int]),
    ("mvs_ctx_kernel_ms", _c.c_int, [_P, _c.c_int, _c.POINTER(_c.c_float)]),
    ("mvs_ctx_pairwise_candidates", _c.c_int, [_P, _c.POINTER(_c.c_int64)]),
    ("mvs_comm_library", _c.c_int, [_c.c_char_p, _c.c_size_t, _c.POINTER(_c.c_int)]),
    ("mvs_ctx_pairwise_stats", _c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    ("mvs_project_csr", _c.c_int, [_P, _P, _c.c_int, _P, _c.c_int64, _c.c_int, _P, _c.c_int]),
    ("mvs_project_csr_stats", _c.c_int, [_P, _P, _c.c_int, _P, _c.c_int64, _c.c_int, _P, _c.c_int, _P,
                                          _c.POINTER(_c.c_int64)]),
    ("mvs_sketch_sumsq", _c.c_int, [_P, _P, _c.c_int, _c.c_int64, _c.c_int, _P, _c.c_int]),
    ("mvs_sketch_stats", _c.c_int, [_P, _P, _c.c_int, _c.c_int64, _c.c_int, _P, _c.c_int, _c.POINTER(_c.c_int64)]),
    ("mvs_norms_sq_text", _c.c_int, [_P, _P, _c.c_int64, _c.c_int, _P]),
    ("mvs_sketch_saturate_i16", _c.c_int, [_P, _P, _c.c_int, _c.c_int64, _P, _c.c_int]),
    ("mvs_sketch_max_abs", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.POINTER(_c.c_int64)]),
    ("mvs_limbs_for_max_abs", _c.c_int, [_c.c_int64]),
    ("mvs_limb_geometry", _c.c_int, [_c.c_int64, _c.c_int, _c.c_int, _c.POINTER(_c.c_int64),
                                      _c.POINTER(_c.c_int), _c.POINTER(_c.c_size_t)]),
    ("mvs_limb_split", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int, _c.c_int, _P, _c.c_int,
                                   _c.c_int64]),
    ("mvs_sketch_set_create", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int, _c.POINTER(_P)]),
    ("mvs_sketch_set_from_planes", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int, _c.c_int,
                                               _c.POINTER(_P)]),
    ("mvs_sketch_set_alloc", _c.c_int, [_P, _c.c_int64, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    ("mvs_sketch_set_fill", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64]),
    ("mvs_sketch_set_fill_stats", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.POINTER(_c.c_int64)]),
    ("mvs_sketch_set_info", _c.c_int, [_P, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int),
                                        _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int)]),
    ("mvs_sketch_set_destroy", _c.c_int, [_P]),
    ("mvs_pairwise_rows", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _P, _c.c_int64,
                                      _c.c_int, _c.POINTER(_c.c_int64)]),
    ("mvs_pairwise_stream", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_size_t, ROW_BLOCK_CB, _P,
                                        _c.POINTER(_c.c_int64)]),
    ("mvs_pairwise_stream_encoded", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_size_t,
                                                ENCODED_ROWS_CB, _P, _c.POINTER(_c.c_int64)]),
    ("mvs_ctx_stream_stats", _c.c_int, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64),
                                         _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int)]),
    ("mvs_pairwise_block", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int,
                                       _P, _c.c_int64, _c.POINTER(_c.c_int64)]),
    ("mvs_cells_sort", _c.c_int, [_P, _P, _c.c_int64, _P]),
    ("mvs_search_block", _c.c_int, [_P, _P, _P, _c.c_double, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                     _c.c_int64, _c.POINTER(_c.c_int64)]),
    ("mvs_pairwise_dots", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int,
                                      _c.c_int]),
    ("mvs_sketch_set_planes", _c.c_int, [_P, _c.POINTER(_P)]),
    ("mvs_sketch_set_touch", _c.c_int, [_P]),
    ("mvs_comm_unique_id", _c.c_int, [_P]),
    ("mvs_comm_create", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    ("mvs_comm_create_callbacks", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    ("mvs_comm_create_files", _c.c_int, [_P, _c.c_char_p, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    ("mvs_comm_create_rendezvous", _c.c_int, [_P, _c.c_char_p, _c.c_int, _c.c_int, _c.POINTER(_P)]),
    ("mvs_comm_destroy", _c.c_int, [_P]),
    ("mvs_comm_info", _c.c_int, [_P, _c.POINTER(_c.c_int), _c.POINTER(_c.c_int), _c.POINTER(_c.c_int)]),
    ("mvs_allgather_planes", _c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int, _c.c_int]),
    ("mvs_allgather_rows", _c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int, _c.c_int]),
    ("mvs_allgather_f64", _c.c_int, [_P, _P, _P, _c.c_int64]),
    ("mvs_allgather_bytes", _c.c_int, [_P, _P, _P, _c.c_int64]),
    ("mvs_allreduce_max_i64", _c.c_int, [_P, _P, _c.POINTER(_c.c_int64)]),
    ("mvs_shard_layout", _c.c_int, [_c.c_int64, _c.c_int, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
    ("mvs_sketch_set_attach_derived", _c.c_int, [_P, _P, _P]),
    ("mvs_sketch_set_prepare_rows", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64]),
    ("mvs_sketch_set_recode_rows", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64]),
    ("mvs_sketch_set_planes_from_wire", _c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64]),
    ("mvs_plan_begin", _c.c_int, [_P, _P, _P, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int, _P, _c.c_int64]),
    ("mvs_plan_wire", _c.c_int, [_P, _P]),
    ("mvs_plan_rows_ready", _c.c_int, [_P, _c.c_int64, _c.c_int64]),
    ("mvs_plan_filter", _c.c_int, [_P, _P, _c.c_int]),
    ("mvs_plan_finish", _c.c_int, [_P, _c.POINTER(_P)]),
    ("mvs_plan_stats", _c.c_int, [_P, _c.POINTER(_c.c_double), _c.POINTER(_c.c_int64)]),
    ("mvs_cells_route", _c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _c.c_int64, _P,
                                    _c.c_int64, _P, _P, _c.c_int64, _c.c_int64, _c.c_int64]),
    ("mvs_cells_collect", _c.c_int, [_P, _P, _c.c_int, _c.c_int, _c.c_int64, _c.c_int64, _c.c_int64, _P, _c.c_int64, _P]),
    ("mvs_cells_report", _c.c_int, [_P, _P, _c.c_int, _c.c_int64, _c.c_int64, _P, _c.POINTER(_c.c_int64)]),
    ("mvs_cells_sort_rows", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P]),
    ("mvs_cells_sort_rows_ahead", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, _P, _P, _c.c_int64]),
    ("mvs_sketch_set_wire_rows", _c.c_int, [_P, _P, _P, _c.c_int64, _c.c_int64]),
    ("mvs_device_alloc", _c.c_int, [_P, _c.c_size_t, _c.c_int, _c.POINTER(_P)]),
    ("mvs_device_free", _c.c_int, [_P, _P]),
    ("mvs_device_zero", _c.c_int, [_P, _P, _c.c_size_t]),
    ("mvs_device_copy", _c.c_int, [_P, _P, _c.c_int, _P, _c.c_int, _c.c_size_t]),
    ("mvs_event_create", _c.c_int, [_P, _c.c_int, _c.POINTER(_P)]),
    ("mvs_event_record", _c.c_int, [_P, _P]),
    ("mvs_ctx_wait_event", _c.c_int, [_P, _P]),
    ("mvs_event_synchronize", _c.c_int, [_P]),
    ("mvs_event_elapsed_ms", _c.c_int, [_P, _P, _c.POINTER(_c.c_float)]),
    ("mvs_event_destroy", _c.c_int, [_P]),
    ("mvs_cells_stream", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, ROW_BLOCK_CB, _P, _c.POINTER(_c.c_int64)]),
    ("mvs_cells_stream_encoded", _c.c_int, [_P, _P, _c.c_int64, _c.c_int64, _c.c_int64, ENCODED_ROWS_CB, _P, _c.POINTER(_c.c_int64)]),
    ("mvs_chunk_size", _c.c_int64, [_c.c_double, _c.c_int]),
    ("mvs_shard_rows", None, [_c.c_int64, _c.c_int, _c.c_int, _c.POINTER(_c.c_int64), _c.POINTER(_c.c_int64)]),
]

_lib = None


class MvsError(RuntimeError):
    def __init__(self, code, message, needed=None):
        super().__init__("libmvs_hip error %d: %s" % (code, message))
        self.code = code
        self.needed = needed     # MVS_E_CAPACITY: the size the call reported it needs


def load_library():
    """Load libmvs_hip.so (once).  torch, when present, is imported first so that the process uses a
    single HIP runtime (torch's bundled libamdhip64 has the same SONAME as the system one)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmvs_hip.so not found at %s -- build it with `make -C metagenome_vector_sketches_amd/csrc` "
            "(or __graft_entry__.build()); there is no CPU fallback" % LIB_PATH)
    try:
        import torch  # noqa: F401  (side effect: loads the HIP runtime torch ships)
    except Exception:  # torch is plumbing for tests/bench, not a requirement of the library
        pass
    lib = ctypes.CDLL(LIB_PATH)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the ABI is incomplete
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def _check(rc):
    if rc != MVS_OK:
        raise MvsError(rc, load_library().mvs_last_error().decode("utf-8", "replace"))


def _is_torch(x):
    return type(x).__module__.startswith("torch")


def _buf(x, dtype=None, writable=False):
    """-> (pointer, mem flag, keepalive).  numpy arrays are host buffers, torch CUDA tensors device."""
    if x is None:
        return None, MEM_HOST, None
    if _is_torch(x):
        if not x.is_contiguous():
            raise ValueError("tensor must be contiguous")
        return x.data_ptr(), (MEM_DEVICE if x.is_cuda else MEM_HOST), x
    a = np.ascontiguousarray(x, dtype=dtype) if not writable else x
    if writable and (not a.flags["C_CONTIGUOUS"] or (dtype is not None and a.dtype != np.dtype(dtype))):
        raise ValueError("output array must be C-contiguous %s" % dtype)
    return a.ctypes.data, MEM_HOST, a


class SketchSet:
    """Limb planes of N samples resident in HBM (mvs_sketch_set)."""

    def __init__(self, ctx, handle, keep=None):
        self.ctx, self._h, self._keep = ctx, handle, keep
        ctx._sets.add(self)
        n, d, limbs, n_alloc, d_pad = (_c.c_int64(), _c.c_int(), _c.c_int(), _c.c_int64(), _c.c_int())
        _check(ctx.lib.mvs_sketch_set_info(handle, n, d, limbs, n_alloc, d_pad))
        self.n, self.d, self.limbs, self.n_alloc, self.d_pad = n.value, d.value, limbs.value, n_alloc.value, d_pad.value

    def fill(self, sketches, row_offset=0):
        """re-code `sketches` into rows [row_offset, row_offset + len) of a set made by sketch_set_alloc"""
        n, d = sketches.shape
        p, m, k = _buf(sketches)
        eb = self.ctx._elem_bytes(sketches)
        _check(self.ctx.lib.mvs_sketch_set_fill(self._h, p, eb, m, int(row_offset), n))

    def fill_stats(self, sketches, row_offset=0):
        """fill() that also returns the largest |v| of the rows it re-coded (one upload instead of a max_abs pass and a
        fill pass; a value beyond what the set's limb count holds means: allocate again with more limbs)"""
        n, d = sketches.shape
        p, m, k = _buf(sketches)
        eb = self.ctx._elem_bytes(sketches)
        mx = _c.c_int64()
        _check(self.ctx.lib.mvs_sketch_set_fill_stats(self._h, p, eb, m, int(row_offset), n, ctypes.byref(mx)))
        return mx.value

    def touch(self):
        """the caller has rewritten the planes: data the library derived from them is rebuilt on the next comparison"""
        _check(self.ctx.lib.mvs_sketch_set_touch(self._h))

    def close(self):
        if self._h:
            self.ctx.lib.mvs_sketch_set_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


COMM_ID_BYTES = 128
PLAN_MIRROR_OUTSIDE = 1
CELLS_HEADER_BYTES = 64
WIRE_MAX_ABS = 32004          # mvs_sketch_set_planes_from_wire: largest |v| for which the low limb pins the value
PLAN_STALE = 1 << 62          # mvs_plan_finish under option plan_speculate: the cell count of a plan that must run again


class Comm:
    """Communicator of the multi-GPU exchange step (mvs_comm): RCCL (one process per GPU), the file transport
    (ranks sharing a device), or caller-supplied collectives."""

    def __init__(self, ctx, handle, keep=None):
        self.ctx, self._h, self._keep = ctx, handle, keep
        ctx._comms.add(self)             # a communicator holds a pointer to its context: closed before the context is
        r, w, k = _c.c_int(), _c.c_int(), _c.c_int()
        _check(ctx.lib.mvs_comm_info(handle, r, w, k))
        self.rank, self.world, self.is_rccl = r.value, w.value, bool(k.value)

    def allgather_planes(self, planes, rows_per_rank, limbs, d_pad):
        pp, pm, pk = _buf(planes)
        if pm != MEM_DEVICE:
            raise ValueError("planes must be a device buffer")
        _check(self.ctx.lib.mvs_allgather_planes(self.ctx._h, self._h, pp, int(rows_per_rank), int(limbs), int(d_pad)))

    def allgather_rows(self, planes, rows_per_rank, row_first, row_count, limbs, d_pad):
        pp, pm, pk = _buf(planes)
        if pm != MEM_DEVICE:
            raise ValueError("planes must be a device buffer")
        _check(self.ctx.lib.mvs_allgather_rows(self.ctx._h, self._h, pp, int(rows_per_rank), int(row_first), int(row_count),
                                               int(limbs), int(d_pad)))

    def allgather_f64(self, values, count_per_rank):
        vp, vm, vk = _buf(values)
        if vm != MEM_DEVICE:
            raise ValueError("values must be a device buffer")
        _check(self.ctx.lib.mvs_allgather_f64(self.ctx._h, self._h, vp, int(count_per_rank)))

    def allgather_bytes(self, buf, bytes_per_rank):
        bp, bm, bk = _buf(buf)
        if bm != MEM_DEVICE:
            raise ValueError("buf must be a device buffer")
        _check(self.ctx.lib.mvs_allgather_bytes(self.ctx._h, self._h, bp, int(bytes_per_rank)))

    def allreduce_max(self, value):
        v = _c.c_int64(int(value))
        _check(self.ctx.lib.mvs_allreduce_max_i64(self.ctx._h, self._h, ctypes.byref(v)))
        return v.value

    def close(self):
        if self._h:
            self.ctx.lib.mvs_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def comm_library():
    """(path, version) of the RCCL the library's communicators use -- bound at run time; raises MvsError without one"""
    lib = load_library()
    buf = ctypes.create_string_buffer(4096)
    ver = _c.c_int()
    _check(lib.mvs_comm_library(buf, len(buf), ctypes.byref(ver)))
    return buf.value.decode("utf-8", "replace"), ver.value


def comm_unique_id():
    """rank 0: the 128-byte RCCL id to hand to the other ranks"""
    buf = ctypes.create_string_buffer(COMM_ID_BYTES)
    _check(load_library().mvs_comm_unique_id(buf))
    return buf.raw


class Context:
    """One device + one stream (mvs_ctx)."""

    def __init__(self, device=0, stream=None):
        self.lib = load_library()
        h = _P()
        _check(self.lib.mvs_ctx_create(int(device), ctypes.byref(h)))
        self._h = h
        self.device = device
        self._sets = weakref.WeakSet()   # sketch sets hold a pointer to the context: close them first
        self._comms = weakref.WeakSet()  # communicators likewise
        if stream is not None:
            self.set_stream(stream)

    def close(self):
        if getattr(self, "_h", None):
            for s in list(self._sets):
                s.close()
            for m in list(self._comms):
                m.close()
            self.lib.mvs_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream):
        """stream: a torch.cuda.Stream or a raw hipStream_t value (0 = HIP's default stream);
        None switches back to the context's own non-blocking stream."""
        if stream is None:
            _check(self.lib.mvs_ctx_use_own_stream(self._h))
            return
        if hasattr(stream, "cuda_stream"):
            stream = stream.cuda_stream
        _check(self.lib.mvs_ctx_set_stream(self._h, _P(int(stream)) if int(stream) else None))

    def synchronize(self):
        _check(self.lib.mvs_ctx_synchronize(self._h))

    def comm_rccl(self, unique_id, rank, world):
        h = _P()
        _check(self.lib.mvs_comm_create(self._h, unique_id, int(rank), int(world), ctypes.byref(h)))
        return Comm(self, h)

    def comm_files(self, path_prefix, rank, world):
        h = _P()
        _check(self.lib.mvs_comm_create_files(self._h, path_prefix.encode(), int(rank), int(world), ctypes.byref(h)))
        return Comm(self, h)

    def comm_rendezvous(self, path_prefix, rank, world):
        """RCCL communicator whose ranks find each other under a path prefix (mvs_comm_create_rendezvous)"""
        h = _P()
        _check(self.lib.mvs_comm_create_rendezvous(self._h, path_prefix.encode(), int(rank), int(world), ctypes.byref(h)))
        return Comm(self, h)

    def set_option(self, name, value):
        """tuning / diagnostic switch of this context (include/mvs_hip.h lists them); never changes a result"""
        _check(self.lib.mvs_ctx_set_option(self._h, name.encode(), int(value)))

    def get_option(self, name):
        v = _c.c_int64()
        _check(self.lib.mvs_ctx_get_option(self._h, name.encode(), ctypes.byref(v)))
        return v.value

    def options(self, **kw):
        """`with ctx.options(pairwise_filter=0): ...` -- set, run, restore"""
        ctx = self

        class _Scope:
            def __enter__(self):
                self.old = {k: ctx.get_option(k) for k in kw}
                for k, v in kw.items():
                    ctx.set_option(k, v)
                return ctx

            def __exit__(self, *exc):
                for k, v in self.old.items():
                    ctx.set_option(k, v)
                return False
        return _Scope()

    def set_timing(self, enabled=True):
        _check(self.lib.mvs_ctx_set_timing(self._h, 1 if enabled else 0))

    def kernel_ms(self, which):
        ms = _c.c_float()
        _check(self.lib.mvs_ctx_kernel_ms(self._h, which, ctypes.byref(ms)))
        return ms.value

    def pairwise_candidates(self):
        """candidate pairs the last comparison's coarse filter passed to the exact re-check (0: exact kernel only)"""
        v = _c.c_int64()
        _check(self.lib.mvs_ctx_pairwise_candidates(self._h, ctypes.byref(v)))
        return v.value

    def pairwise_stats(self):
        """(candidates, flagged_tiles, filter_tiles) of the last two-stage comparison: pairs re-checked one by one, and how
        many of the filter's 256 x 256 tiles went to the exact kernel whole"""
        a, b, t = _c.c_int64(), _c.c_int64(), _c.c_int64()
        _check(self.lib.mvs_ctx_pairwise_stats(self._h, ctypes.byref(a), ctypes.byref(b), ctypes.byref(t)))
        return a.value, b.value, t.value

    # ---- projection ----
    def project_csr(self, hashes, offsets, d, out=None):
        """hashes: uint64 numpy array or torch CUDA tensor (int64 view of the bits is accepted);
        offsets: host int64 array (n_samples+1).  Returns out (numpy unless `out` is given)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        if _is_torch(hashes):
            hp, hm, hk = _buf(hashes)
        else:
            hp, hm, hk = _buf(hashes, np.uint64)
        if out is None:
            out = np.empty((n, d), dtype=np.int32)
        op, om, ok = _buf(out, np.int32, writable=True)
        _check(self.lib.mvs_project_csr(self._h, hp, hm, offsets.ctypes.data, n, int(d), op, om))
        return out

    def project_csr_stats(self, hashes, offsets, d, out, sumsq):
        """project_csr into the device tensor `out`, filling the device int64 tensor `sumsq` with the exact
        sums of squares; returns the largest |v| (fused into the projection kernel when every sample is
        a single unit)."""
        offsets = np.ascontiguousarray(offsets, dtype=np.int64)
        n = len(offsets) - 1
        hp, hm, hk = _buf(hashes) if _is_torch(hashes) else _buf(hashes, np.uint64)
        op, om, ok = _buf(out)
        sp, sm, sk = _buf(sumsq)
        if om != MEM_DEVICE or sm != MEM_DEVICE:
            raise ValueError("out and sumsq must be device buffers")
        m = _c.c_int64()
        _check(self.lib.mvs_project_csr_stats(self._h, hp, hm, offsets.ctypes.data, n, int(d), op, om, sp,
                                              ctypes.byref(m)))
        return m.value

    def sumsq(self, sketches, out=None):
        n, d = sketches.shape
        ip, im, ik = _buf(sketches, np.int32)
        if out is None:
            out = np.empty(n, dtype=np.int64)
        op, om, ok = _buf(out, np.int64, writable=True)
        _check(self.lib.mvs_sketch_sumsq(self._h, ip, im, n, d, op, om))
        return out

    def norms_sq_text(self, sumsq, d, out):
        """device int64 sums of squares -> device float64 squared norms as they come back from vector_norms.txt
        (6-significant-digit text round trip, exact); asynchronous on the context's stream."""
        ip, im, ik = _buf(sumsq, np.int64)
        op, om, ok = _buf(out, np.float64, writable=True)
        if im != MEM_DEVICE or om != MEM_DEVICE:
            raise ValueError("norms_sq_text works on device arrays")
        _check(self.lib.mvs_norms_sq_text(self._h, ip, int(sumsq.shape[0]), int(d), op))
        return out

    def stats(self, sketches, out=None):
        """-> (sumsq, max_abs): per-row exact sum of squares and the largest |v|, one pass."""
        n, d = sketches.shape
        ip, im, ik = _buf(sketches, np.int32)
        if out is None:
            out = np.empty(n, dtype=np.int64)
        op, om, ok = _buf(out, np.int64, writable=True)
        m = _c.c_int64()
        _check(self.lib.mvs_sketch_stats(self._h, ip, im, n, d, op, om, ctypes.byref(m)))
        return out, m.value

    def saturate_i16(self, sketches, out=None):
        ip, im, ik = _buf(sketches, np.int32)
        n_elems = int(np.prod(sketches.shape))
        if out is None:
            out = np.empty(tuple(sketches.shape), dtype=np.int16)
        op, om, ok = _buf(out, np.int16, writable=True)
        _check(self.lib.mvs_sketch_saturate_i16(self._h, ip, im, n_elems, op, om))
        return out

    # ---- pairwise ----
    @staticmethod
    def _elem_bytes(sketches):
        if _is_torch(sketches):
            return sketches.element_size()
        return np.asarray(sketches).dtype.itemsize

    def max_abs(self, sketches):
        eb = self._elem_bytes(sketches)
        p, m, k = _buf(sketches)
        out = _c.c_int64()
        _check(self.lib.mvs_sketch_max_abs(self._h, p, eb, m, int(np.prod(sketches.shape)), ctypes.byref(out)))
        return out.value

    def limb_geometry(self, n, d, limbs):
        n_alloc, d_pad, nbytes = _c.c_int64(), _c.c_int(), _c.c_size_t()
        _check(self.lib.mvs_limb_geometry(n, d, limbs, n_alloc, d_pad, nbytes))
        return n_alloc.value, d_pad.value, nbytes.value

    def limb_split(self, sketches, limbs, planes, d_pad, row_offset=0):
        n, d = sketches.shape
        eb = self._elem_bytes(sketches)
        p, m, k = _buf(sketches)
        pp, pm, pk = _buf(planes)
        if pm != MEM_DEVICE:
            raise ValueError("planes must be a device buffer")
        _check(self.lib.mvs_limb_split(self._h, p, eb, m, n, d, limbs, pp, d_pad, row_offset))

    def sketch_set(self, sketches, limbs=None):
        """limbs=None lets the library choose the limb code from max|v|; an explicit code (1..4 or
        LIMBS_K3) allocates and fills the planes with that scheme (the caller guarantees the range)."""
        n, d = sketches.shape
        eb = self._elem_bytes(sketches)
        if eb not in (2, 4):
            raise ValueError("sketches must be int32 or int16")
        p, m, k = _buf(sketches)
        h = _P()
        if limbs is None:
            _check(self.lib.mvs_sketch_set_create(self._h, p, eb, m, n, d, ctypes.byref(h)))
        else:
            _check(self.lib.mvs_sketch_set_alloc(self._h, n, d, int(limbs), ctypes.byref(h)))
            rc = self.lib.mvs_sketch_set_fill(h, p, eb, m, 0, n)
            if rc != MVS_OK:
                self.lib.mvs_sketch_set_destroy(h)
                _check(rc)
            self.synchronize()
        return SketchSet(self, h)

    def sketch_set_alloc(self, n, d, limbs):
        """zeroed planes for n samples with limb code `limbs`; fill row ranges with SketchSet.fill"""
        h = _P()
        _check(self.lib.mvs_sketch_set_alloc(self._h, int(n), int(d), int(limbs), ctypes.byref(h)))
        return SketchSet(self, h)

    def sketch_set_from_planes(self, planes, n, n_alloc, d, d_pad, limbs):
        pp, pm, pk = _buf(planes)
        if pm != MEM_DEVICE:
            raise ValueError("planes must be a device buffer")
        h = _P()
        _check(self.lib.mvs_sketch_set_from_planes(self._h, pp, n, n_alloc, d, d_pad, limbs, ctypes.byref(h)))
        return SketchSet(self, h, keep=planes)

    def pairwise_rows(self, sset, norms_sq, row_begin=0, row_end=None, keep_mode=KEEP_INT32, capacity=None,
                      cells_out=None):
        """Returns (cells, n_cells).  cells is a numpy structured array (CELL_DTYPE) unless a device
        buffer `cells_out` (torch int32 tensor of shape [capacity, 4]) is given.  Grows the host buffer
        and retries on MVS_E_CAPACITY."""
        if row_end is None:
            row_end = sset.n
        np_, nm, nk = _buf(norms_sq) if _is_torch(norms_sq) else _buf(norms_sq, np.float64)
        count = _c.c_int64()
        if cells_out is not None:
            cp, cm, ck = _buf(cells_out)
            cap = cells_out.shape[0]
            rc = self.lib.mvs_pairwise_rows(self._h, sset._h, np_, nm, keep_mode, row_begin, row_end, cp, cap, cm,
                                            ctypes.byref(count))
            _check(rc)
            return cells_out, count.value
        cap = capacity if capacity is not None else max(1024, 64 * (row_end - row_begin))
        while True:
            cells = np.empty(cap, dtype=CELL_DTYPE)
            rc = self.lib.mvs_pairwise_rows(self._h, sset._h, np_, nm, keep_mode, row_begin, row_end,
                                            cells.ctypes.data, cap, MEM_HOST, ctypes.byref(count))
            if rc == MVS_E_CAPACITY and capacity is None:
                cap = count.value
                continue
            _check(rc)
            return cells[:count.value], count.value

    def pairwise_stream(self, sset, norms_sq, on_block=None, row_begin=0, row_end=None, keep_mode=KEEP_INT32,
                        device_budget_bytes=0):
        """mvs_pairwise_stream: the kept cells of rows [row_begin, row_end) in CSR pieces of whole rows, ascending.
        on_block(row_begin, row_end, row_ptr, col, q) is called per piece with numpy COPIES (q is uint8, or uint16 in the
        mismatched-norms case); a truthy return stops the comparison (MvsError MVS_E_ABORTED).  Without on_block the
        pieces are collected: returns (row_ptr int64 [rows + 1], col int32 [n], q [n], n)."""
        if row_end is None:
            row_end = sset.n
        np_, nm, nk = _buf(norms_sq) if _is_torch(norms_sq) else _buf(norms_sq, np.float64)
        parts, errors = [], []

        def trampoline(_user, bp):
            try:
                b = bp.contents
                rows, n = b.row_end - b.row_begin, b.n_cells
                rp = np.ctypeslib.as_array(b.row_ptr, shape=(rows + 1,)).copy()
                col = np.ctypeslib.as_array(b.col, shape=(n,)).copy() if n else np.empty(0, np.int32)
                if n == 0:
                    q = np.empty(0, np.uint8)
                elif b.q:
                    q = np.ctypeslib.as_array(b.q, shape=(n,)).copy()
                else:
                    q = np.ctypeslib.as_array(b.q16, shape=(n,)).copy()
                if on_block is not None:
                    return 1 if on_block(b.row_begin, b.row_end, rp, col, q) else 0
                parts.append((b.row_begin, b.row_end, rp, col, q))
                return 0
            except BaseException as e:      # noqa: BLE001 -- an exception must not unwind through the C frames
                errors.append(e)
                return -1

        cb = ROW_BLOCK_CB(trampoline)
        count = _c.c_int64()
        rc = self.lib.mvs_pairwise_stream(self._h, sset._h, np_, nm, keep_mode, int(row_begin), int(row_end),
                                          int(device_budget_bytes), cb, None, ctypes.byref(count))
        if errors:
            raise errors[0]
        _check(rc)
        if on_block is not None:
            return count.value
        rows = row_end - row_begin
        row_ptr = np.zeros(rows + 1, dtype=np.int64)
        at, expect = 0, row_begin
        for (b0, b1, rp, col, q) in parts:
            assert b0 == expect and rp[0] == 0, "pieces must arrive in ascending row order, each row once"
            row_ptr[b0 - row_begin:b1 - row_begin + 1] = rp + at
            at += int(rp[-1])
            expect = b1
        assert expect == row_end and at == count.value
        wide = any(p[4].dtype == np.uint16 for p in parts)
        col = np.concatenate([p[3] for p in parts]) if parts else np.empty(0, np.int32)
        q = np.concatenate([p[4].astype(np.uint16 if wide else np.uint8) for p in parts]) if parts else np.empty(0, np.uint8)
        return row_ptr, col, q, count.value

    def pairwise_stream_encoded(self, sset, norms_sq, row_begin=0, row_end=None, keep_mode=KEEP_INT32, device_budget_bytes=0):
        """mvs_pairwise_stream_encoded, pieces collected: returns dict(rows uint32 [R], first_col uint32 [R], offset uint64
        [R] into `bytes`, jac_bytes uint32 [R], bytes uint8 [B] = what the shard writer appends to matrix.bin, n_cells)"""
        if row_end is None:
            row_end = sset.n
        np_, nm, nk = _buf(norms_sq) if _is_torch(norms_sq) else _buf(norms_sq, np.float64)
        parts, errors = [], []

        def trampoline(_user, bp):
            try:
                b = bp.contents
                r, nb = b.n_rows, b.n_bytes

                def arr(ptr, n, dt):
                    return np.ctypeslib.as_array(ptr, shape=(n,)).copy() if n else np.empty(0, dt)
                parts.append((b.row_begin, b.row_end, arr(b.rows, r, np.uint32), arr(b.first_col, r, np.uint32),
                              arr(b.offset, r, np.uint64), arr(b.jac_bytes, r, np.uint32), arr(b.bytes, nb, np.uint8), b.n_cells))
                return 0
            except BaseException as e:      # noqa: BLE001
                errors.append(e)
                return -1

        cb = ENCODED_ROWS_CB(trampoline)
        count = _c.c_int64()
        rc = self.lib.mvs_pairwise_stream_encoded(self._h, sset._h, np_, nm, keep_mode, int(row_begin), int(row_end),
                                                  int(device_budget_bytes), cb, None, ctypes.byref(count))
        if errors:
            raise errors[0]
        _check(rc)
        expect, at = row_begin, 0
        offs = []
        for p in parts:
            assert p[0] == expect, "pieces must arrive in ascending row order"
            expect = p[1]
            offs.append(p[4] + np.uint64(at))
            at += len(p[6])
        assert expect == row_end and sum(p[7] for p in parts) == count.value
        cat = lambda i, dt: np.concatenate([p[i] for p in parts]) if parts else np.empty(0, dt)   # noqa: E731
        return {"rows": cat(2, np.uint32), "first_col": cat(3, np.uint32),
                "offset": np.concatenate(offs) if offs else np.empty(0, np.uint64), "jac_bytes": cat(5, np.uint32),
                "bytes": cat(6, np.uint8), "n_cells": count.value, "pieces": len(parts)}

    def stream_stats(self):
        """what the last pairwise_stream did: dict(kernel_ms, bytes, row_blocks, pieces, two_stage: 0 exact kernel, 1 two-stage as one list, 2 two-stage through the dense byte matrix)"""
        k, b, r, p_, t = _c.c_double(), _c.c_int64(), _c.c_int64(), _c.c_int64(), _c.c_int()
        _check(self.lib.mvs_ctx_stream_stats(self._h, ctypes.byref(k), ctypes.byref(b), ctypes.byref(r), ctypes.byref(p_),
                                             ctypes.byref(t)))
        return {"kernel_ms": k.value, "bytes": b.value, "row_blocks": r.value, "pieces": p_.value, "two_stage": int(t.value)}

    def pairwise_block(self, sset, norms_sq, row_begin, row_end, col_begin, col_end, flags, cells, n_cells,
                       keep_mode=KEEP_INT32):
        """Append the kept cells of one block to the device buffer `cells` ([capacity, 4] int32) at index
        n_cells; returns the new count."""
        np_, nm, nk = _buf(norms_sq)
        cp, cm, ck = _buf(cells)
        if nm != MEM_DEVICE or cm != MEM_DEVICE:
            raise ValueError("norms_sq and cells must be device buffers")
        count = _c.c_int64(int(n_cells))
        rc = self.lib.mvs_pairwise_block(self._h, sset._h, np_, keep_mode, row_begin, row_end, col_begin, col_end,
                                         flags, cp, cells.shape[0], ctypes.byref(count))
        if rc == MVS_E_CAPACITY:
            raise MvsError(rc, self.lib.mvs_last_error().decode("utf-8", "replace"), needed=count.value)
        _check(rc)
        return count.value

    def search_block(self, sset, norms_sq, jaccard_min, row_begin, row_end, col_begin, col_end, cells):
        """pairs (query row, database column) with Jaccard estimate > jaccard_min -> number of hits written to
        the device buffer `cells`, sorted by (row, col)"""
        np_, nm, nk = _buf(norms_sq)
        cp, cm, ck = _buf(cells)
        if nm != MEM_DEVICE or cm != MEM_DEVICE:
            raise ValueError("norms_sq and cells must be device buffers")
        count = _c.c_int64()
        rc = self.lib.mvs_search_block(self._h, sset._h, np_, float(jaccard_min), row_begin, row_end, col_begin,
                                       col_end, cp, cells.shape[0], ctypes.byref(count))
        if rc == MVS_E_CAPACITY:       # count = the hits there are: the caller can come back with exactly that room
            raise MvsError(rc, self.lib.mvs_last_error().decode("utf-8", "replace"), needed=count.value)
        _check(rc)
        return count.value

    # ---- block plans (include/mvs_hip.h "block plans"): a rank's share of the symmetric multi-rank schedule ----
    def attach_derived(self, sset, coarse_fm, row_stats):
        """the filter's inputs of `sset` live in these device buffers from now on (n_alloc * d_pad bytes, n_alloc * 16 bytes)"""
        cp, cm, ck = _buf(coarse_fm)
        rp, rm, rk = _buf(row_stats)
        if cm != MEM_DEVICE or rm != MEM_DEVICE:
            raise ValueError("device buffers required")
        _check(self.lib.mvs_sketch_set_attach_derived(sset._h, cp, rp))
        sset._derived = (coarse_fm, row_stats)

    def prepare_rows(self, sset, row_first, row_count):
        _check(self.lib.mvs_sketch_set_prepare_rows(self._h, sset._h, int(row_first), int(row_count)))

    def recode_rows(self, sset, sketches, row_first, row_count):
        """limb planes + filter inputs of rows [row_first, row_first + row_count) in one pass: the first len(sketches) of them
        from `sketches` (device [n, d]; None: none), the rest zero rows"""
        n = 0 if sketches is None else sketches.shape[0]
        sp, sm, sk = _buf(sketches)
        if n and sm != MEM_DEVICE:
            raise ValueError("sketches must be a device buffer")
        _check(self.lib.mvs_sketch_set_recode_rows(self._h, sset._h, sp, self._elem_bytes(sketches) if n else 4, n, int(row_first),
                                                   int(row_count)))

    def planes_from_wire(self, sset, lo_wire, row_first, row_count):
        """limb planes of rows [row_first, row_first + row_count) from their low limbs (lo_wire: device bytes, row r at
        r * d_pad) and the attached coarse plane / statistics (mvs_sketch_set_planes_from_wire)"""
        lp, lm, lk = _buf(lo_wire)
        if lm != MEM_DEVICE:
            raise ValueError("the wire buffer must be a device buffer")
        _check(self.lib.mvs_sketch_set_planes_from_wire(self._h, sset._h, lp, int(row_first), int(row_count)))

    def plan_rows_ready(self, row_begin, row_end):
        """statistics and norms of these rows are in place: their filter constants in one launch (mvs_plan_rows_ready)"""
        _check(self.lib.mvs_plan_rows_ready(self._h, int(row_begin), int(row_end)))

    def plan_wire(self, lo_wire):
        """the plan in progress rebuilds the limb planes of the foreign rows its second half reads from lo_wire (mvs_plan_wire)"""
        lp, lm, lk = _buf(lo_wire)
        if lm != MEM_DEVICE:
            raise ValueError("the wire buffer must be a device buffer")
        _check(self.lib.mvs_plan_wire(self._h, lp))

    def plan_begin(self, sset, norms_sq, frame_begin, frame_end, mirror_outside, cells, keep_mode=KEEP_INT32):
        np_, nm, nk = _buf(norms_sq)
        cp, cm, ck = _buf(cells)
        if nm != MEM_DEVICE or cm != MEM_DEVICE:
            raise ValueError("norms_sq and cells must be device buffers")
        _check(self.lib.mvs_plan_begin(self._h, sset._h, np_, keep_mode, int(frame_begin), int(frame_end),
                                       PLAN_MIRROR_OUTSIDE if mirror_outside else 0, cp, cells.shape[0]))

    def plan_filter(self, blocks):
        """blocks: [(row_begin, row_end, col_begin, col_end)] -> ONE filter launch (asynchronous)"""
        arr = (_c.c_int64 * (4 * len(blocks)))(*[int(x) for b in blocks for x in b[:4]])
        _check(self.lib.mvs_plan_filter(self._h, arr, len(blocks)))

    def plan_finish(self):
        """re-check + flagged tiles; -> device address of the running cell count"""
        p = _P()
        _check(self.lib.mvs_plan_finish(self._h, ctypes.byref(p)))
        return p.value

    def plan_stats(self):
        ms = (_c.c_double * 4)()
        cnt = (_c.c_int64 * 6)()
        _check(self.lib.mvs_plan_stats(self._h, ms, cnt))
        return {"filter_ms": ms[0], "recheck_ms": ms[1], "tiles_ms": ms[2], "span_ms": ms[3], "candidates": cnt[0],
                "flagged_tiles": cnt[1], "filter_tiles": cnt[2], "filter_launches": cnt[3], "exact_mode": bool(cnt[4] & 1),
                "speculated": bool(cnt[4] & 2), "stale": bool(cnt[4] & 4), "d_pad": cnt[5]}

    def cells_route(self, raw, d_n_raw, block_pad, block_rows, n_total, own_begin, own_end, own_out, d_own_count, send,
                    foreign_capacity, status=0, max_abs=0):
        """d_n_raw: device ADDRESS (int) of the cell count (plan_finish); d_own_count: device tensor of one int64 / uint64"""
        rp, rm, rk = _buf(raw)
        op, om, ok = _buf(own_out)
        cp, cm, ck = _buf(d_own_count)
        sp, sm, sk = _buf(send)
        if rm != MEM_DEVICE or om != MEM_DEVICE or cm != MEM_DEVICE or (send is not None and sm != MEM_DEVICE):
            raise ValueError("device buffers required")
        _check(self.lib.mvs_cells_route(self._h, rp, _P(int(d_n_raw)), raw.shape[0], int(block_pad), int(block_rows), int(n_total),
                                        int(own_begin), int(own_end), op, own_out.shape[0], cp, sp, int(foreign_capacity),
                                        int(status), int(max_abs)))

    def cells_collect(self, recv, world, rank, foreign_capacity, own_begin, own_end, own_out, d_own_count):
        rp, rm, rk = _buf(recv)
        op, om, ok = _buf(own_out)
        cp, cm, ck = _buf(d_own_count)
        _check(self.lib.mvs_cells_collect(self._h, rp, int(world), int(rank), int(foreign_capacity), int(own_begin), int(own_end),
                                          op, own_out.shape[0], cp))

    def cells_report(self, recv, world, foreign_capacity, own_rows, d_own_count):
        """-> (cells of this shard, [(foreign cells, status, max_abs, raw cells, raw capacity)] per rank, most cells in one
        row); synchronises"""
        rp, rm, rk = _buf(recv)
        cp, cm, ck = _buf(d_own_count)
        out = (_c.c_int64 * (2 + 5 * world))()
        _check(self.lib.mvs_cells_report(self._h, rp, int(world), int(foreign_capacity), int(own_rows), cp, out))
        return out[0], [tuple(out[1 + 5 * r:6 + 5 * r]) for r in range(world)], out[1 + 5 * world]

    def cells_sort_rows(self, cells_in, n, own_begin, own_end, d_own_count, cells_out):
        ip, im, ik = _buf(cells_in)
        op, om, ok = _buf(cells_out)
        cp, cm, ck = _buf(d_own_count)
        _check(self.lib.mvs_cells_sort_rows(self._h, ip, int(n), int(own_begin), int(own_end), cp, op))

    def cells_sort_rows_ahead(self, cells_in, own_begin, own_end, d_own_count, cells_out):
        """the row-bucket sort queued before the report's read-back: the count is the one on the device, both buffers' sizes bound it"""
        ip, im, ik = _buf(cells_in)
        op, om, ok = _buf(cells_out)
        cp, cm, ck = _buf(d_own_count)
        _check(self.lib.mvs_cells_sort_rows_ahead(self._h, ip, int(cells_in.shape[0]), int(own_begin), int(own_end), cp, op,
                                                  int(cells_out.shape[0])))

    def cells_sort(self, cells_in, n, cells_out):
        ip, im, ik = _buf(cells_in)
        op, om, ok = _buf(cells_out)
        if im != MEM_DEVICE or om != MEM_DEVICE:
            raise ValueError("device buffers required")
        _check(self.lib.mvs_cells_sort(self._h, ip, int(n), op))

    def pairwise_dots(self, sset, r0, r1, c0, c1, algo=0, out=None):
        if out is None:
            out = np.empty((r1 - r0, c1 - c0), dtype=np.int32)
        op, om, ok = _buf(out, np.int32, writable=True)
        _check(self.lib.mvs_pairwise_dots(self._h, sset._h, r0, r1, c0, c1, op, om, algo))
        return out


def chunk_size(max_memory_gb, d):
    return load_library().mvs_chunk_size(float(max_memory_gb), int(d))


def shard_layout(n_total, world):
    """(rows per shard = ceil(n / world), the same rounded up to a multiple of 256: the block size in storage coordinates)"""
    a, b = _c.c_int64(), _c.c_int64()
    _check(load_library().mvs_shard_layout(int(n_total), int(world), ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


def shard_rows(n, num_shards, shard_idx):
    b, e = _c.c_int64(), _c.c_int64()
    load_library().mvs_shard_rows(n, num_shards, shard_idx, ctypes.byref(b), ctypes.byref(e))
    return b.value, e.value


def limbs_for_max_abs(max_abs):
    return load_library().mvs_limbs_for_max_abs(int(max_abs))
