"""Row-sharded all-vs-all comparison across the GPUs of one node (one process per GPU).

Partitioning is the reference's own (src/pairwise_comp_optimized.cpp:938-940): rank r owns rows
[r*ceil(N/G), min((r+1)*ceil(N/G), N)) and ends up with exactly the `shard_r/` of a `--num_shards G --shard_idx r`
run.  The reference's "exchange" is every shard process re-reading the shared vectors.bin and comparing its rows
against ALL columns (:949-982); here a rank sketches (or loads) only its own rows and the step is

  own rows  -> limb planes + the filter's inputs (coarse plane, row statistics) for OWN rows only
  exchange  -> all-gather of the row statistics and norms (bytes per row), of the coarse plane in row CHUNKS, then of
               the limb planes -- on the communicator's stream (RCCL over xGMI), in that order: the filter needs only the
               first two, the limb planes are read by the exact re-check at the very end
  compare   -> the rank's share of the symmetric schedule as ONE block plan (libmvs_hip: mvs_plan_*): every unordered
               pair of row blocks is compared by exactly one rank (block (r, r+k) by rank r for 0 < k < G/2, the opposite
               block of an even G split in halves), kept cells outside the rank's own square are mirrored.  The filter
               runs as one launch for the diagonal block -- it needs nothing from anybody, so it starts before the
               exchange has delivered a byte -- and one launch per arrived chunk of the peers' rows
  cells     -> split on the device into this shard's cells and mirror images that belong to other ranks' rows; ONE
               fixed-size all-gather of the latter (its header carries every rank's status and largest |v|: no separate
               agreement collectives), collect, sort.

Host synchronisations per step: one inside mvs_plan_finish (candidate and flagged-tile counts size the later launches)
and one for the final cell counts.

Storage coordinates: every rank's block of the gathered buffers is padded to a multiple of 256 rows
(_capi.shard_layout), so that all blocks start on the tile grid and on the 16-row grid of the fragment-major planes
whatever N and G are; kept cells are translated back to sample indices when they are routed.

The numeric work goes through an `ops` object (`GpuOps`: libmvs_hip.so through the C ABI; tests/ substitutes a CPU
stand-in built on the oracle to exercise this module with the gloo backend), the exchange through a `collectives`
object: `NativeCollectives` wraps the communicator of the C ABI (mvs_comm: RCCL bound at run time, or the file transport
for ranks sharing a device) -- the same entry points the C++ `pairwise_comp_optimized` uses with MVS_COLLECTIVE=rccl;
`TorchCollectives` wraps a torch.distributed process group (gloo in the CPU tests, nccl = RCCL otherwise).
MVS_SHARDED_SYMMETRIC=0 selects the plain rows x all-columns schedule (no mirroring, no cell exchange).
"""
import os

import numpy as np

from . import _capi


# ---------------------------------------------------------------------------------------------------------------
# collectives
# ---------------------------------------------------------------------------------------------------------------
class _Done:
    """handle of an exchange that has already happened"""

    def wait(self):
        return None


class _StreamHandle:
    """an exchange queued on a side stream: wait() makes the CURRENT stream wait for it (no host wait)"""

    def __init__(self, event):
        self.event = event

    def wait(self):
        import torch
        torch.cuda.current_stream().wait_event(self.event)


class _ThreadHandle:
    """an exchange that runs on a worker thread (the file transport blocks its caller): wait() joins it, then orders the
    current stream behind what it queued"""

    def __init__(self, future):
        self.future = future

    def wait(self):
        import torch
        ev = self.future.result()
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)


class _Collectives:
    """submit(fn): run the collective calls in fn() so that they see everything the compute stream has queued so far;
    returns a handle whose wait() orders the compute stream behind them.  Base: synchronous."""
    stream = None

    def submit(self, fn):
        fn()
        return _Done()

    def _submit_on_stream(self, fn):
        import torch
        main = torch.cuda.current_stream()
        self.stream.wait_stream(main)
        with torch.cuda.stream(self.stream):
            fn()
            ev = torch.cuda.Event()
            ev.record(self.stream)
        return _StreamHandle(ev)


class TorchCollectives(_Collectives):
    """the collectives of the exchange over torch.distributed.  stream: a torch.cuda.Stream the collectives are issued
    on (None: the caller's current stream, synchronous semantics)"""
    kind = "torch.distributed"

    def __init__(self, dist, rank, world, stream=None):
        self.dist, self.rank, self.world, self.stream = dist, rank, world, stream
        self._send = None

    def submit(self, fn):
        if self.stream is None:
            return super().submit(fn)
        return self._submit_on_stream(fn)

    def allgather_blocks(self, buf, block_elems):
        """buf: 1-D tensor of world * block_elems elements whose block `rank` is filled in"""
        send = buf[self.rank * block_elems:(self.rank + 1) * block_elems]
        if self._send is None or self._send.shape != send.shape or self._send.dtype != send.dtype:
            self._send = send.new_empty(send.shape)
        self._send.copy_(send)     # a separate send block: NCCL / gloo need not support aliased in-place gathers
        self.dist.all_gather_into_tensor(buf[:self.world * block_elems], self._send)

    def allgather_rows(self, buf, rows_per_rank, row_first, row_count, row_bytes):
        """rows [row_first, row_first + row_count) of every rank's block of a byte buffer (mvs_allgather_rows)"""
        if row_count == rows_per_rank:
            return self.allgather_blocks(buf, rows_per_rank * row_bytes)
        if row_count == 0:
            return
        view = buf[:self.world * rows_per_rank * row_bytes].view(self.world, rows_per_rank * row_bytes)[
            :, row_first * row_bytes:(row_first + row_count) * row_bytes]
        tmp = buf.new_empty(self.world * row_count * row_bytes)
        self.dist.all_gather_into_tensor(tmp, view[self.rank].contiguous())
        view.copy_(tmp.view(self.world, row_count * row_bytes))


class NativeCollectives(_Collectives):
    """the same over the communicator of the C ABI (in place on the device, on the stream of the communicator's
    context).  stream: that stream as a torch.cuda.Stream when it is NOT the compute stream (bench.py gives the
    communicator a context of its own on a side stream, so that an exchange runs beside the compute stream's kernels).
    RCCL calls are asynchronous: submit() queues them on the side stream and returns.  The file transport (ranks sharing
    one card: rehearsals and tests) blocks inside every call, so its calls run on a worker thread."""

    def __init__(self, comm, stream=None):
        self.comm, self.rank, self.world, self.stream = comm, comm.rank, comm.world, stream
        self.kind = "libmvs_hip mvs_comm (%s)" % ("RCCL" if comm.is_rccl else "file transport")
        self._pool = None

    def submit(self, fn):
        if self.stream is None:
            return super().submit(fn)
        if self.comm.is_rccl:
            return self._submit_on_stream(fn)
        import torch
        from concurrent.futures import ThreadPoolExecutor
        if self._pool is None:
            self._pool = ThreadPoolExecutor(max_workers=1)      # ONE thread: the collectives keep their order
        ready = torch.cuda.Event()
        ready.record(torch.cuda.current_stream())
        device = torch.cuda.current_device()

        def job():
            torch.cuda.set_device(device)
            self.stream.wait_event(ready)
            fn()
            ev = torch.cuda.Event()
            ev.record(self.stream)
            return ev
        return _ThreadHandle(self._pool.submit(job))

    def allgather_blocks(self, buf, block_elems):
        self.comm.allgather_bytes(buf, block_elems * buf.element_size())

    def allgather_rows(self, buf, rows_per_rank, row_first, row_count, row_bytes):
        # mvs_allgather_rows speaks of rows of limbs * d_pad bytes with d_pad a multiple of 128
        if row_bytes % 128 != 0:
            raise ValueError("row size must be a multiple of 128 bytes")
        self.comm.allgather_rows(buf, rows_per_rank, row_first, row_count, 1, row_bytes)


# ---------------------------------------------------------------------------------------------------------------
# partitioning
# ---------------------------------------------------------------------------------------------------------------
def shard_rows(n_total, world, rank):
    """src/pairwise_comp_optimized.cpp:938-940"""
    rps = (n_total + world - 1) // world
    b = min(rank * rps, n_total)
    return b, min(b + rps, n_total)


def half_split(block_pad):
    """where the opposite block of an even world is cut: a multiple of 256 rows near the middle"""
    return ((block_pad // 256 + 1) // 2) * 256


def block_plan(world, rank, block_pad, symmetric=True):
    """Rectangles (row_begin, row_end, col_begin, col_end) in STORAGE coordinates (rank p's rows at [p * block_pad,
    (p + 1) * block_pad)) that rank `rank` compares.  Symmetric schedule: its diagonal block first (tiles below the
    diagonal are skipped by the kernel and produced by mirroring), then the blocks whose kept cells are mirrored into other
    ranks' rows -- over all ranks every unordered pair of rows is covered exactly once.  Otherwise: its rows against every
    block (the reference's schedule)."""
    P = block_pad
    rb, re = rank * P, (rank + 1) * P
    plan = [(rb, re, rb, re)]
    if not symmetric:
        return plan + [(rb, re, p * P, (p + 1) * P) for p in range(world) if p != rank]
    for k in range(1, (world - 1) // 2 + 1):
        p = (rank + k) % world
        plan.append((rb, re, p * P, (p + 1) * P))
    if world > 1 and world % 2 == 0:
        p = (rank + world // 2) % world
        h = half_split(P)
        if rank < p:                      # the lower rank takes the first half of ITS rows against all of p's
            if h > 0:
                plan.append((rb, rb + h, p * P, (p + 1) * P))
        elif h < P:                       # the higher rank takes all of its rows against the second half of p's
            plan.append((rb, re, p * P + h, (p + 1) * P))
    return plan


def chunk_bounds(block_pad, chunks, first=None):
    """[(c0, c1)] cutting a block of block_pad rows into at most `chunks` pieces on multiples of 256 rows; first (0 < first < 1):
    the share of the first piece, the others split the rest evenly"""
    tiles = block_pad // 256
    chunks = max(1, min(chunks, tiles))
    if first is not None and chunks > 1:
        t0 = max(1, min(tiles - (chunks - 1), int(round(tiles * first))))
        rest = tiles - t0
        cuts = [0] + [(t0 + rest * k // (chunks - 1)) * 256 for k in range(chunks)]
    else:
        cuts = [(tiles * k // chunks) * 256 for k in range(chunks + 1)]
    return [(cuts[k], cuts[k + 1]) for k in range(chunks) if cuts[k + 1] > cuts[k]]


def clip_blocks(blocks, block_pad, c0, c1):
    """the parts of the rectangles whose columns lie at offsets [c0, c1) of their rank block"""
    out = []
    for (rb, re, cb, ce) in blocks:
        base = (cb // block_pad) * block_pad
        lo, hi = max(cb, base + c0), min(ce, base + c1)
        if hi > lo:
            out.append((rb, re, lo, hi))
    return out


# ---------------------------------------------------------------------------------------------------------------
# numeric back end
# ---------------------------------------------------------------------------------------------------------------
class GpuOps:
    """numeric back end on one MI355X: everything is a call into libmvs_hip.so"""

    def __init__(self, ctx, device):
        import torch
        self.ctx, self.device = ctx, device
        # buffers here are torch tensors (zero fills, copies and slices run on torch's stream): the library
        # must issue its kernels on that same stream or nothing orders them against each other
        self._main = torch.cuda.current_stream(torch.device(device))
        ctx.set_stream(self._main)
        self.speculate = os.environ.get("MVS_PLAN_SPECULATE", "1") != "0"
        if os.environ.get("MVS_PLAN_OVERLAP", "0") == "1":      # A/B: filter launches of a plan alternate between two streams
            ctx.set_option("plan_overlap", 1)

    def layout(self, n_total, world):
        return _capi.shard_layout(n_total, world)

    def new_cells(self, capacity):
        import torch
        return torch.empty((capacity, 4), dtype=torch.int32, device=self.device)

    def new_bytes(self, nbytes):
        import torch
        return torch.zeros(nbytes, dtype=torch.uint8, device=self.device)

    def new_planes(self, nbytes):
        import torch
        return torch.zeros(nbytes, dtype=torch.int8, device=self.device)

    def new_counter(self, rows=0):
        """the shard's state block (mvs_cells_route): count, largest row, one uint32 per own row (+ one)"""
        import torch
        return torch.zeros(2 + (rows + 2) // 2, dtype=torch.int64, device=self.device)

    def zero_count(self):
        """a count reference (what plan_finish returns) that reads zero: a rank whose comparison failed routes nothing"""
        if getattr(self, "_zero", None) is None:
            self._zero = self.new_counter()
        return self._zero.data_ptr()

    def zero_(self, t):
        t.zero_()

    def to_device(self, host_array):
        import torch
        return torch.from_numpy(np.ascontiguousarray(host_array)).to(self.device)

    def to_host_cells(self, cells, n):
        return cells[:n].cpu().numpy().view(_capi.CELL_DTYPE).reshape(-1)

    def max_abs(self, sketches):
        return self.ctx.max_abs(sketches)

    def limbs_for(self, max_abs):
        return _capi.limbs_for_max_abs(max_abs)

    def limb_geometry(self, n, d, limbs):
        return self.ctx.limb_geometry(n, d, limbs)

    def limb_split(self, sketches, limbs, planes, d_pad, row_offset):
        self.ctx.limb_split(sketches, limbs, planes, d_pad, row_offset)

    def open_set(self, planes, n, n_alloc, d, d_pad, limbs, coarse_fm, stats):
        sset = self.ctx.sketch_set_from_planes(planes, n, n_alloc, d, d_pad, limbs)
        self.ctx.attach_derived(sset, coarse_fm, stats)
        return sset

    def close_set(self, sset):
        sset.close()

    def touch_set(self, sset):
        sset.touch()

    def prepare_rows(self, sset, first, count):
        self.ctx.prepare_rows(sset, first, count)

    def recode_rows(self, sset, sketches, limbs, planes, d_pad, first, count):
        """limb planes and filter inputs of storage rows [first, first + count) in one pass: the first len(sketches) of them
        from `sketches`, the rest (behind a shard's last sample) as zero rows"""
        if _capi._is_torch(sketches) and sketches.is_cuda:
            self.ctx.recode_rows(sset, sketches if sketches.shape[0] else None, first, count)
            return
        if sketches.shape[0]:                      # host sketches: upload inside the limb split, then derive
            self.ctx.limb_split(sketches, limbs, planes, d_pad, first)
        self.ctx.prepare_rows(sset, first, count)

    def wire_rows(self, planes, lo_wire, d_pad, first, count):
        """the LOW limbs of rows [first, first + count) of two-limb planes into the wire buffer (row r at r * d_pad)"""
        rows = planes[:(first + count) * 2 * d_pad].view(-1, 2, d_pad)
        lo_wire[first * d_pad:(first + count) * d_pad].view(count, d_pad).copy_(rows[first:first + count, 0, :])

    def planes_from_wire(self, sset, lo_wire, first, count):
        """both limb planes of rows [first, first + count) from their low limbs + the coarse plane / statistics that are in
        place (mvs_sketch_set_planes_from_wire)"""
        self.ctx.planes_from_wire(sset, lo_wire, first, count)

    def plan_begin(self, sset, norms_sq, f0, f1, mirror_outside, raw, keep_mode):
        # a step of the same shape as the previous one runs its plan ahead of the read-backs (the decision is taken in
        # mvs_plan_begin; ShardedComparison.finish knows what a stale plan looks like): MVS_PLAN_SPECULATE=0 turns it off
        if self.speculate:
            self.ctx.set_option("plan_speculate", 1)
        try:
            self.ctx.plan_begin(sset, norms_sq, f0, f1, mirror_outside, raw, keep_mode=keep_mode)
        finally:
            if self.speculate:
                self.ctx.set_option("plan_speculate", 0)

    def plan_filter(self, blocks):
        self.ctx.plan_filter(blocks)

    def plan_rows_ready(self, row_begin, row_end):
        self.ctx.plan_rows_ready(row_begin, row_end)

    def plan_wire(self, lo_wire):
        """the plan in progress rebuilds, inside its finish, the limb planes of exactly the foreign rows it reads"""
        self.ctx.plan_wire(lo_wire)

    def plan_exact_mode(self):
        """the plan that has just begun runs without a filter: its blocks are compared by the exact kernel when handed over"""
        return self.ctx.plan_stats()["exact_mode"]

    def plan_finish(self):
        return self.ctx.plan_finish()

    def plan_stats(self):
        return self.ctx.plan_stats()

    def cells_route(self, raw, d_n_raw, block_pad, block_rows, n_total, own, own_out, d_own, send, cap_f, status, max_abs):
        self.ctx.cells_route(raw, d_n_raw, block_pad, block_rows, n_total, own[0], own[1], own_out, d_own, send, cap_f,
                             status, max_abs)

    def cells_collect(self, recv, world, rank, cap_f, own, own_out, d_own):
        self.ctx.cells_collect(recv, world, rank, cap_f, own[0], own[1], own_out, d_own)

    def cells_report(self, recv, world, cap_f, own, d_own):
        return self.ctx.cells_report(recv, world, cap_f, own[1] - own[0], d_own)

    def sort_cells_ahead(self, cells_in, cells_out, own, d_own):
        """the row-bucket sort before the host knows the count (mvs_cells_sort_rows_ahead)"""
        self.ctx.cells_sort_rows_ahead(cells_in, own[0], own[1], d_own, cells_out)

    def sort_cells(self, cells_in, n, cells_out, own=None, d_own=None, max_row=None):
        """(row, col) order: by row buckets when no row holds more than 64 cells (the usual shard), else the general sort"""
        if own is not None and max_row is not None and max_row <= 64:
            self.ctx.cells_sort_rows(cells_in, n, own[0], own[1], d_own, cells_out)
        else:
            self.ctx.cells_sort(cells_in, n, cells_out)


# ---------------------------------------------------------------------------------------------------------------
# the step
# ---------------------------------------------------------------------------------------------------------------
class ShardedComparison:
    """State that survives between steps (the gathered buffers are reused while their geometry holds).

    run() is the whole exchange + comparison for local rows that are complete.  begin() / feed() / finish() do the same
    for local rows that become final in PARTS (bench.py projects a rank's samples in pieces): the planes and filter
    inputs of a finished part are handed to the all-gather at once, so the exchange of part k runs beside the projection
    of part k + 1.  The limb code has to be fixed before the first row is known, so rows are coded with `limbs_guess` (two
    base-256 limbs: |v| <= 32639, what sketches of up to tens of millions of hashes need; fewer limbs than the guess is
    exact too) and every rank's largest |v| travels in the header of the cell exchange: if the guess does not hold on some
    rank, every rank sees that and redoes the step with the limb code the data needs."""

    def __init__(self, ops, rank=0, world=1, dist=None, collectives=None):
        """dist: a torch.distributed module with an initialised default group, or collectives: a
        TorchCollectives / NativeCollectives object (takes precedence)."""
        self.ops, self.rank, self.world = ops, rank, world
        self.coll = collectives if collectives is not None else (TorchCollectives(dist, rank, world) if dist is not None else None)
        if world > 1 and self.coll is None:
            raise ValueError("world > 1 needs torch.distributed or a communicator")
        self.symmetric = os.environ.get("MVS_SHARDED_SYMMETRIC", "1") != "0"
        self.gather_chunks = max(1, int(os.environ.get("MVS_GATHER_CHUNKS", "2")))   # pieces the peers' coarse rows arrive in
        self.gather_first = float(os.environ.get("MVS_GATHER_FIRST", "0.33"))        # share of the first piece
        # two-limb sets: the exchange carries coarse plane + LOW limbs and the receiver rebuilds the high limbs -- 2 bytes per
        # entry on the links instead of 3.  The plan rebuilds, inside its finish, only the rows its re-check and flagged tiles
        # read (mvs_plan_wire: the columns of its candidates), so the cost follows the candidates: 0.10 ms at the per-rank size
        # of an 8-way split of configs[2] (52k candidates over 87.8k foreign rows), 0.03 ms for configs[3] (365 candidates).
        # Predicted at 61 GB/s per link: 2 ranks 1.67 -> 1.84 x, 4 ranks 3.14 -> 3.3 x, 8 ranks 5.58 -> 5.61 x (configs[3]: 5.6 ->
        # 6.2 x); at twice that rate it costs 1-2 % (tools/strong_model.py --wire / --no-wire).  Switched off for good once a
        # rank reports |v| beyond what the rule covers; MVS_WIRE_LOW_LIMB=0 switches it off from the start.
        self.wire = os.environ.get("MVS_WIRE_LOW_LIMB", "1") != "0" and hasattr(ops, "planes_from_wire")
        self.time_gather = False         # bench: events around the exchange (read with last_gather_ms())
        self.trace = None                # a list: (label, torch event) pairs of the last step (tools/exp/r05_overlap_trace.py)
        self._key = None
        self._planes = self._coarse = self._stats = self._n2 = self._sset = self._lo = None
        self._raw = self._own = self._xbuf = self._d_own = None
        self._d_own_rows = 0
        self._cap_f = 1 << 14
        # capacities of a step that was given no output buffer, as functions of the shard's rows (tests shrink them): the
        # shard itself, and what the rank's blocks may produce per direction (None: as the shard)
        self.own_capacity_default = lambda rows: max(1 << 16, 64 * max(rows, 1))
        self.raw_capacity_default = None
        self._sort_ahead = False         # the previous step's shard had no row beyond 64 cells and fitted its buffers
        self._step = None
        self._ev = None

    # ---- timing aids ----
    def last_gather_ms(self):
        """span from the first to the last collective of the last step on the exchange stream (0 with one rank);
        synchronises on the end event"""
        if self._ev is None or self._ev[0] is None or self._ev[1] is None:
            return 0.0
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])

    def _mark(self, which):
        """inside a submitted exchange: an event on the stream the collectives run on"""
        if not self.time_gather or self.world == 1 or getattr(self.coll, "stream", None) is None:
            return
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(self.coll.stream)
        if self._ev is None:
            self._ev = [None, None]
        if which == 0 and self._ev[0] is None:
            self._ev[0] = ev
        if which == 1:
            self._ev[1] = ev

    def _trace(self, label, stream=None):
        if self.trace is None:
            return
        import torch
        ev = torch.cuda.Event(enable_timing=True)
        ev.record(stream if stream is not None else torch.cuda.current_stream())
        self.trace.append((label, ev))

    # ---- buffers ----
    def _prepare(self, n_total, d, limbs):
        ops, world = self.ops, self.world
        rps, P = ops.layout(n_total, world)
        n_st = P * world                                 # storage rows; the rows behind a shard's samples stay zero
        n_alloc, d_pad, nbytes = ops.limb_geometry(n_st, d, limbs)
        key = (limbs, n_alloc, d_pad, n_st, n_total, d)
        if self._key != key:
            if self._sset is not None:
                ops.close_set(self._sset)
                self._sset = None
            self._planes = ops.new_planes(nbytes)
            self._coarse = ops.new_bytes(n_alloc * d_pad)          # fragment-major coarse plane (two-limb sets)
            self._stats = ops.new_bytes(n_alloc * 16)              # 16 bytes of row statistics
            self._n2 = ops.to_device(np.zeros(n_alloc, dtype=np.float64))
            self._lo = ops.new_planes(n_alloc * d_pad) if (limbs == 2 and world > 1 and self.wire) else None
            self._sset = ops.open_set(self._planes, n_st, n_alloc, d, d_pad, limbs, self._coarse, self._stats)
            self._key = key
        return rps, P, n_alloc, d_pad

    def _put_norms(self, norms_sq_part, first, count, P):
        """norms of local rows [first, first + count) into this rank's block of the gathered norm buffer"""
        n2_all, base = self._n2, self.rank * P + first
        if _capi._is_torch(norms_sq_part) and not _capi._is_torch(n2_all):
            norms_sq_part = norms_sq_part.cpu().numpy()      # host back end (tests): plain arrays
        if _capi._is_torch(norms_sq_part) and getattr(norms_sq_part, "device", None) == getattr(n2_all, "device", None):
            n2_all[base:base + count].copy_(norms_sq_part)   # already on the device: no host round trip
        else:
            host = norms_sq_part.cpu().numpy() if _capi._is_torch(norms_sq_part) else norms_sq_part
            n2_all[base:base + count] = self.ops.to_device(np.ascontiguousarray(host, dtype=np.float64))

    # ---- whole steps ----
    def run(self, sketches_local, norms_sq_local, n_total, keep_mode=_capi.KEEP_INT32, cells_out=None,
            max_abs_local=None, limbs_guess=2):
        """sketches_local: this rank's rows (int32/int16 [n_local, d]); norms_sq_local: float64 [n_local], a host
        array or a device tensor; max_abs_local: largest |v| of sketches_local if the caller already has it
        (Context.stats).  Returns (cells, n_cells, info) for this rank's shard, sorted by (row, col).
        cells_out (device [capacity, 4] int32): capacity only has to hold this shard's cells; without it the cells come
        back as a host array."""
        n_local = sketches_local.shape[0]
        if max_abs_local is None:
            max_abs_local = self.ops.max_abs(sketches_local) if n_local else 0
        self.begin(sketches_local, norms_sq_local, n_total, limbs_guess=limbs_guess)
        self.feed(0, self._step["P"], int(max_abs_local))
        return self.finish(keep_mode=keep_mode, cells_out=cells_out)

    def begin(self, sketches_local, norms_sq_local, n_total, limbs_guess=2):
        """sketches_local / norms_sq_local: the rank's FULL buffers ([n_local, d] / [n_local]); their rows become valid
        part by part (feed).  Nothing is read here."""
        n_local, d = sketches_local.shape
        rb, re = shard_rows(n_total, self.world, self.rank)
        if re - rb != n_local:
            raise ValueError("rank %d holds %d rows but its shard is [%d,%d)" % (self.rank, n_local, rb, re))
        rps, P, n_alloc, d_pad = self._prepare(n_total, d, limbs_guess)
        self.ops.touch_set(self._sset)                    # the buffers are about to be rewritten
        self._ev = None
        if self.trace is not None:
            del self.trace[:]
            self._trace("step begin")
        self._step = {"sk": sketches_local, "n2": norms_sq_local, "n_total": n_total, "limbs": limbs_guess, "rps": rps, "P": P,
                      "n_alloc": n_alloc, "d_pad": d_pad, "max_abs": 0, "fed": 0, "parts": 0, "small": [], "coarse": [], "planes": [],
                      "rows": (rb, re), "wire": bool(self.wire and limbs_guess == 2 and self.world > 1 and self._lo is not None),
                      "rebuilt": False}

    def part_bounds(self, n_total, parts):
        """[(row_begin, row_end)] cutting a rank's block (STORAGE rows 0 .. block_pad, the same bounds on every rank: a
        collective's sizes must agree) into `parts` pieces on multiples of 256 rows; a rank whose shard is shorter than the
        block simply has fewer -- or no -- rows of its own inside the later pieces"""
        _, P = self.ops.layout(n_total, self.world)
        return chunk_bounds(P, parts)

    def feed(self, row_begin, row_end, max_abs_part):
        """rows [row_begin, row_end) of this rank's BLOCK (storage rows, multiples of 256, in order -- part_bounds()) are
        final: those of them the rank owns (below its n_local) have been written to the buffers given to begin();
        max_abs_part: their largest |v|.  Their planes and filter inputs are built and, with more than one rank, the
        exchange of exactly this row range of every rank's block starts now."""
        st = self._step
        if st is None or row_begin != st["fed"] or row_end < row_begin or row_end > st["P"] or (row_begin % 256) or \
                (row_end % 256 and row_end != st["P"]):
            raise ValueError("feed(%d, %d) out of order" % (row_begin, row_end))
        ops, P, limbs, d_pad = self.ops, st["P"], st["limbs"], st["d_pad"]
        st["fed"], st["parts"] = row_end, st["parts"] + 1
        st["max_abs"] = max(st["max_abs"], int(max_abs_part))
        n_local = st["sk"].shape[0]
        own_b, own_e = min(row_begin, n_local), min(row_end, n_local)
        base = self.rank * P
        if own_e > own_b:
            self._put_norms(st["n2"][own_b:own_e], own_b, own_e - own_b, P)
        if row_end > row_begin:                   # the rows behind n_local are zero sketches
            ops.recode_rows(self._sset, st["sk"][own_b:own_e], limbs, self._planes, d_pad, base + row_begin, row_end - row_begin)
        self._trace("own rows [%d,%d) ready" % (row_begin, row_end))
        if self.world == 1 or row_end == row_begin:
            return
        coll, last = self.coll, row_end == P
        nl = limbs & 0xff

        def small():                      # bytes per row: what every peer filter launch needs first
            self._mark(0)
            coll.allgather_blocks(self._stats, P * 16)
            coll.allgather_blocks(self._n2, P)
        if last:
            st["small"].append(coll.submit(small))
        # the coarse plane of the part in row chunks (units of 16 rows = d_pad * 16 contiguous bytes), then its limb planes
        # (the first piece is the small one: it has to land while the diagonal block's filter runs, and that takes about as
        # long as a third of the coarse plane takes on a link; and the second piece must not take longer on the links than the first takes
        # to filter: transfer is about twice as fast as filtering per row, hence 1 : 2)
        for (c0, c1) in chunk_bounds(row_end - row_begin, self.gather_chunks, self.gather_first):
            a, b = row_begin + c0, row_begin + c1

            def coarse(a=a, b=b):
                self._mark(0)
                coll.allgather_rows(self._coarse, P // 16, a // 16, (b - a) // 16, 16 * d_pad)
                self._trace("coarse rows [%d,%d) gathered" % (a, b), getattr(coll, "stream", None))
            st["coarse"].append((a, b, coll.submit(coarse)))

        if st["wire"]:
            ops.wire_rows(self._planes, self._lo, d_pad, base + row_begin, row_end - row_begin)

        def planes():
            if not st["wire"]:
                coll.allgather_rows(self._planes, P, row_begin, row_end - row_begin, nl * d_pad)
                self._mark(1)
                self._trace("limb planes [%d,%d) gathered" % (row_begin, row_end), getattr(coll, "stream", None))
                return
            coll.allgather_rows(self._lo, P, row_begin, row_end - row_begin, d_pad)
            self._mark(1)
            self._trace("low limbs [%d,%d) gathered" % (row_begin, row_end), getattr(coll, "stream", None))
        st["planes"].append(coll.submit(planes))

    def _rebuild(self, st):
        """the other ranks' limb planes from their low limbs, coarse plane and statistics, on the compute stream in front of
        the re-check that reads them.  (Queued on the exchange's stream instead -- as soon as the bytes have landed, beside
        the filter launches -- it cost the filters 1.3 x its own time: tools/strong_model.py, round 5.)"""
        P = st["P"]
        for p in range(self.world):
            if p != self.rank:
                self.ops.planes_from_wire(self._sset, self._lo, p * P, P)
        self._trace("limb planes rebuilt")
        st["rebuilt"] = True

    def finish(self, keep_mode=_capi.KEEP_INT32, cells_out=None):
        st, self._step = self._step, None
        if st is None or st["fed"] != st["P"]:
            raise ValueError("finish() before the whole block was fed")
        ops, world, rank = self.ops, self.world, self.rank
        P, rps, limbs, n_total = st["P"], st["rps"], st["limbs"], st["n_total"]
        rb, re = st["rows"]
        info = {"limbs": limbs, "rows": (rb, re), "collectives": self.coll.kind if world > 1 else "none",
                "allgather_bytes_per_rank": P * ((2 if st["wire"] else (limbs & 0xff) + 1)) * st["d_pad"] + P * 24 if world > 1 else 0,
                "wire": "coarse plane + low limbs (high limbs rebuilt by the receiver)" if st["wire"] else "coarse plane + limb planes",
                "schedule": "symmetric" if (self.symmetric and world > 1) else "rows x all columns",
                "overlap": "none" if world == 1 else
                           "diagonal block beside the exchange; peers' blocks per arrived chunk of coarse rows" +
                           ("; exchange of a part beside the projection of the next" if st["parts"] > 1 else "")}
        # two capacities: cap_own = what this rank's shard may hold (the caller's buffer, if given), cap_raw = what its blocks
        # may produce per direction (own cells + as many mirror images).  They grow for different reasons: a raw list that
        # overflowed is a COLLECTIVE matter (that rank compares again, everybody exchanges again -- every rank reads it in the
        # headers), a shard that overflowed is a LOCAL one (the kept cells are still in the raw list, the other ranks' mirror
        # images still in the exchange buffer: route + collect again into a larger buffer, no collective).
        cap_own = cells_out.shape[0] if cells_out is not None else int(self.own_capacity_default(re - rb))
        cap_raw = cap_own if (cells_out is not None or self.raw_capacity_default is None) else int(self.raw_capacity_default(re - rb))
        plan = block_plan(world, rank, P, symmetric=self.symmetric)
        mirror = self.symmetric and world > 1
        status, err, d_cnt = 0, None, None
        need_compute, local_only = True, False
        for attempt in range(8):
            # raw: what this rank's blocks produce -- its own cells AND the mirror images on their way to other ranks.  The
            # buffer is only ever replaced in front of a comparison: between a plan and the routing of its cells it IS the result
            want_raw = (2 if mirror else 1) * cap_raw + 1024
            if need_compute and (self._raw is None or self._raw.shape[0] < want_raw):
                self._raw = ops.new_cells(want_raw)
            if self._own is None or self._own.shape[0] < cap_own:
                self._own = ops.new_cells(cap_own)
            if self._d_own is None or self._d_own_rows < re - rb:
                self._d_own, self._d_own_rows = ops.new_counter(re - rb), re - rb
            if need_compute and status == 0:
                try:
                    d_cnt = self._compare(st, plan, mirror, keep_mode, first=(attempt == 0))
                except _capi.MvsError as e:       # the others learn about it from the header of the cell exchange
                    status, err, d_cnt = e.code, e, None
            # every rank's send buffer = a 64-byte header {foreign cells, status, max |v|, raw cells, raw capacity} + room for
            # cap_f mirror images; ONE all-gather of them tells every rank how every other rank fared
            cap_f = self._cap_f if mirror else 0
            stride = _capi.CELLS_HEADER_BYTES + 16 * cap_f
            if self._xbuf is None or self._xbuf.shape[0] < world * stride:
                if local_only:
                    raise RuntimeError("internal: the exchange buffer of a local repeat must be the one the peers' cells are in")
                self._xbuf = ops.new_bytes(world * stride)
            xb = self._xbuf[:world * stride]
            send = xb[rank * stride:(rank + 1) * stride]
            ops.cells_route(self._raw, d_cnt if d_cnt is not None else ops.zero_count(), P, rps, n_total, (rb, re), self._own,
                            self._d_own, send, cap_f, status, st["max_abs"])
            self._trace("cells routed")
            if world > 1:
                if not local_only:                # (a local repeat rewrote this rank's own block with the same cells)
                    self.coll.submit(lambda: self.coll.allgather_blocks(xb, stride)).wait()
                if mirror:
                    ops.cells_collect(xb, world, rank, cap_f, (rb, re), self._own, self._d_own)
            # the sort, queued in front of the step's host synchronisation when the previous step says the row buckets will do:
            # the device then runs from the plan's first kernel to the sorted shard without waiting for the host
            ahead = bool(self._sort_ahead and cells_out is not None and d_cnt is not None and hasattr(ops, "sort_cells_ahead"))
            if ahead:
                ops.sort_cells_ahead(self._own, cells_out, (rb, re), self._d_own)
                self._trace("cells sorted")
            n_out, heads, max_row = ops.cells_report(xb, world, cap_f, (rb, re), self._d_own)   # the step's host synchronisation
            local_only = False
            worst = max(int(h[1]) for h in heads)
            if worst:
                raise err if err is not None else _capi.MvsError(worst, "another rank failed in its block comparisons")
            if st["wire"] and max(int(h[2]) for h in heads) > _capi.WIRE_MAX_ABS:
                # some rank's values are beyond what the low limb pins: the rebuilt planes are not to be trusted.  Every rank
                # sees the same headers: the step is redone with the limb planes themselves on the wire, from now on
                self.wire = False
                self.begin(st["sk"], st["n2"], n_total, limbs_guess=limbs)
                self.feed(0, P, st["max_abs"])
                cells, cnt, info2 = self.finish(keep_mode=keep_mode, cells_out=cells_out)
                info2["overlap"] = "|v| beyond %d: step redone with the limb planes on the wire" % _capi.WIRE_MAX_ABS
                return cells, cnt, info2
            need = ops.limbs_for(max(int(h[2]) for h in heads))
            plain = need <= 4 and limbs <= 4
            if (need > limbs) if plain else (need != limbs):
                # the guess does not hold (on some rank): every rank sees the same headers and redoes the step
                self.begin(st["sk"], st["n2"], n_total, limbs_guess=need)
                self.feed(0, P, st["max_abs"])
                cells, cnt, info2 = self.finish(keep_mode=keep_mode, cells_out=cells_out)
                info2["overlap"] = "limb guess %d did not hold: step redone with %d" % (limbs, need)
                return cells, cnt, info2
            if any(int(h[3]) >= _capi.PLAN_STALE for h in heads):
                # a rank's plan ran its second half on the previous step's counts and they did not hold (mvs_plan_finish with
                # option plan_speculate): that rank compares again -- the library will not speculate this time --, the
                # others exchange again with it
                need_compute = int(heads[rank][3]) >= _capi.PLAN_STALE
                info["plan_respeculated"] = info.get("plan_respeculated", 0) + 1
                continue
            # ---- decisions every rank takes alike: they read nothing but the exchanged headers ----
            redo = False
            mine = heads[rank]
            need_compute = int(mine[3]) > int(mine[4])      # this rank's raw list overflowed: its blocks again, with room
            if any(int(h[3]) > int(h[4]) for h in heads):   # ... and the others exchange again with it
                redo = True
                if need_compute:
                    cap_raw = max(cap_raw, int(mine[3]) // (2 if mirror else 1) + 1)
            if mirror and max(int(h[0]) for h in heads) > cap_f:
                self._cap_f = max(int(h[0]) for h in heads) * 5 // 4 + 1024
                redo = True
            if redo:
                if cells_out is None and n_out > cap_own:   # (a lower bound while lists overflow: the repeat will tell)
                    cap_own = n_out + n_out // 4
                continue
            # ---- this rank's own business: its shard did not fit.  No other rank knows, and none needs to ----
            if n_out > self._own.shape[0]:
                if cells_out is not None:
                    raise _capi.MvsError(_capi.MVS_E_CAPACITY, "%d cells for this shard but capacity is %d" %
                                         (n_out, cells_out.shape[0]), needed=n_out)
                cap_own = n_out + n_out // 4
                need_compute, local_only = False, True
                info["own_regrown"] = info.get("own_regrown", 0) + 1
                continue
            break
        else:
            raise _capi.MvsError(_capi.MVS_E_CAPACITY, "the step's buffers kept overflowing")
        if cells_out is not None and n_out > cells_out.shape[0]:
            raise _capi.MvsError(_capi.MVS_E_CAPACITY, "%d cells for this shard but capacity is %d" % (n_out, cells_out.shape[0]),
                                 needed=n_out)
        out = cells_out if cells_out is not None else ops.new_cells(max(n_out, 1))
        sorted_ahead = ahead and max_row <= 64            # (the buffers held: checked above)
        if n_out and not sorted_ahead:
            ops.sort_cells(self._own, n_out, out, (rb, re), self._d_own, max_row)
        if not sorted_ahead:
            self._trace("cells sorted")
        self._sort_ahead = max_row <= 64
        info["sorted_ahead"] = sorted_ahead
        info["exchanged_cells"] = int(heads[rank][0]) if mirror else 0
        info["blocks"] = len(plan)
        if cells_out is None:
            return ops.to_host_cells(out, n_out), n_out, info
        return out, n_out, info

    def _compare(self, st, plan, mirror, keep_mode, first):
        """the rank's block plan: diagonal block at once, the other blocks per arrived chunk -> device address of the count"""
        ops, P = self.ops, st["P"]
        rank = self.rank
        ops.plan_begin(self._sset, self._n2, rank * P, (rank + 1) * P, mirror, self._raw, keep_mode)
        self._trace("plan begin")
        ops.plan_filter(plan[:1])                          # nothing of it comes from another rank
        self._trace("filter launched: diagonal block")
        others = plan[1:]
        if first:
            for h in st["small"]:
                h.wait()                                   # row statistics + norms of every rank
        if others and hasattr(ops, "plan_rows_ready"):
            ops.plan_rows_ready(0, self.world * P)         # every peer's filter constants in one launch, not block by block
        if first and others and getattr(ops, "plan_exact_mode", lambda: False)():
            # no filter in this plan (filter switched off, another limb code): mvs_plan_filter runs the exact kernel on a block
            # at once, and that reads the other ranks' LIMB planes -- they have to be there (and rebuilt) before the call
            for (_, _, h) in st["coarse"]:
                h.wait()
            for h in st["planes"]:
                h.wait()
            if st["wire"] and not st["rebuilt"]:
                self._rebuild(st)
            ops.plan_filter(others)
            self._trace("exact kernel launched: peers' blocks")
        elif first and st["coarse"]:
            for (a, b, h) in st["coarse"]:
                h.wait()
                blocks = clip_blocks(others, P, a, b)
                if blocks:
                    ops.plan_filter(blocks)
                    self._trace("filter launched: peers' rows [%d,%d)" % (a, b))
        elif others:
            ops.plan_filter(others)
        if first:
            # the limb planes: the re-check reads them.  Every exchange of the step is joined here even if this rank's plan
            # needs nothing from anybody (rank 1 of 2): the next step rewrites the buffers the collectives read
            for h in st["planes"]:
                h.wait()
        if st["wire"] and not st["rebuilt"] and others:        # (a plan without peer blocks reads nobody's limb planes)
            # the plan rebuilds the rows its re-check and flagged tiles read -- the columns of its candidates -- and no others
            ops.plan_wire(self._lo)
        d_cnt = ops.plan_finish()
        self._trace("plan finished")
        return d_cnt
