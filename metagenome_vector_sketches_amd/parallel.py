"""Row-sharded all-vs-all comparison across the GPUs of one node (one process per GPU).

Partitioning is the reference's own (src/pairwise_comp_optimized.cpp:938-940): rank r owns rows
[r*ceil(N/G), min((r+1)*ceil(N/G), N)) and compares them against ALL N columns, producing exactly the
`shard_r/` of a `--num_shards G --shard_idx r` run.  The reference's "exchange" is every process
re-reading the shared vectors.bin; here every rank sketches (or loads) only its own rows and ONE
all-gather (RCCL over xGMI when the process group is `nccl`) of the int8 limb-plane row blocks gives
every GPU all N columns.  Norms (N doubles) are all-gathered the same way.  Results are not exchanged.

The collective calls are torch.distributed's; the numeric work goes through an `ops` object:
`GpuOps` (libmvs_hip.so through the C ABI) in production.  tests/ substitutes a CPU stand-in built on
the oracle to exercise this module with the gloo backend.

With more than one rank the default schedule is SYMMETRIC across ranks as well: every unordered pair of row
blocks is compared by exactly one rank (block (r, r+d) by rank r for 0 < d < G/2; the opposite block of an
even G is split in halves between the two ranks), which appends each kept cell AND its mirror image; the
mirrored cells that belong to other ranks' rows are then exchanged (one more all-gather, of kept cells --
a few MB) and merged.  Per-rank comparison work drops from G - 1/2 blocks to G/2.  Set
MVS_SHARDED_SYMMETRIC=0 for the plain rows x all-columns schedule.

The exchange itself goes through a small `collectives` object: `NativeCollectives` wraps the communicator of the
C ABI (mvs_comm: RCCL bound at run time, or the file transport for ranks sharing a device) -- the same entry
points the C++ `pairwise_comp_optimized` uses with MVS_COLLECTIVE=rccl; `TorchCollectives` wraps a
torch.distributed process group (gloo in the CPU tests, nccl = RCCL otherwise).
"""
import os

import numpy as np

from . import _capi


class TorchCollectives:
    """the collectives of the exchange over torch.distributed.  stream: a torch.cuda.Stream the collectives are issued
    on (None: the caller's current stream) -- a side stream lets an exchange run beside the compute stream's kernels"""
    kind = "torch.distributed"

    def __init__(self, dist, rank, world, stream=None):
        self.dist, self.rank, self.world, self.stream = dist, rank, world, stream

    def _on_stream(self):
        import contextlib
        if self.stream is None:
            return contextlib.nullcontext()
        import torch
        return torch.cuda.stream(self.stream)

    def allreduce_max(self, value, like):
        import torch
        with self._on_stream():
            t = torch.tensor([int(value)], dtype=torch.int64, device=getattr(like, "device", "cpu"))
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            return int(t.cpu()[0])

    def allgather_blocks(self, buf, block_elems):
        """buf: 1-D tensor of world * block_elems elements whose block `rank` is filled in"""
        with self._on_stream():
            send = buf[self.rank * block_elems:(self.rank + 1) * block_elems]
            if self._send is None or self._send.shape != send.shape or self._send.dtype != send.dtype:
                self._send = send.new_empty(send.shape)
            self._send.copy_(send)     # a separate send block: NCCL / gloo need not support aliased in-place gathers
            self.dist.all_gather_into_tensor(buf[:self.world * block_elems], self._send)

    def allgather_rows(self, planes, rows_per_rank, row_first, row_count, limbs, d_pad):
        """rows [row_first, row_first + row_count) of every rank's block of the plane buffer (mvs_allgather_rows)"""
        if row_count == rows_per_rank:
            return self.allgather_blocks(planes, rows_per_rank * (limbs & 0xff) * d_pad)
        if row_count == 0:
            return
        with self._on_stream():
            rb = (limbs & 0xff) * d_pad
            view = planes[:self.world * rows_per_rank * rb].view(self.world, rows_per_rank * rb)[:, row_first * rb:(row_first + row_count) * rb]
            tmp = planes.new_empty(self.world * row_count * rb)
            self.dist.all_gather_into_tensor(tmp, view[self.rank].contiguous())
            view.copy_(tmp.view(self.world, row_count * rb))

    _send = None


class NativeCollectives:
    """the same over the communicator of the C ABI (in place on the device, on the stream of the communicator's
    context).  stream: that stream as a torch.cuda.Stream when it is NOT the compute stream (bench.py gives the
    communicator a context of its own on a side stream, so that an exchange can run beside the projection kernel);
    ShardedComparison then orders the two streams around every exchange."""

    def __init__(self, comm, stream=None):
        self.comm, self.rank, self.world, self.stream = comm, comm.rank, comm.world, stream
        self.kind = "libmvs_hip mvs_comm (%s)" % ("RCCL" if comm.is_rccl else "file transport")

    def allreduce_max(self, value, like):
        return self.comm.allreduce_max(value)

    def allgather_blocks(self, buf, block_elems):
        self.comm.allgather_bytes(buf, block_elems * buf.element_size())

    def allgather_rows(self, planes, rows_per_rank, row_first, row_count, limbs, d_pad):
        self.comm.allgather_rows(planes, rows_per_rank, row_first, row_count, limbs, d_pad)


def shard_rows(n_total, world, rank):
    """src/pairwise_comp_optimized.cpp:938-940"""
    rps = (n_total + world - 1) // world
    b = min(rank * rps, n_total)
    return b, min(b + rps, n_total)


def block_plan(n_total, world, rank):
    """Blocks (row_begin, row_end, col_begin, col_end, flags) rank `rank` compares in the symmetric schedule.
    Over all ranks every unordered pair of samples is covered exactly once (diagonal blocks: both orders)."""
    rb, re = shard_rows(n_total, world, rank)
    plan = []
    if re > rb:
        plan.append((rb, re, rb, re, _capi.BLOCK_SYMMETRIC))
    for d in range(1, (world - 1) // 2 + 1):
        cb, ce = shard_rows(n_total, world, (rank + d) % world)
        if re > rb and ce > cb:
            plan.append((rb, re, cb, ce, _capi.BLOCK_MIRROR_ALL))
    if world > 1 and world % 2 == 0:
        p = (rank + world // 2) % world
        pb, pe = shard_rows(n_total, world, p)
        if rank < p:                      # the lower rank takes the first half of ITS rows against all of p's
            mid = rb + (re - rb + 1) // 2
            if mid > rb and pe > pb:
                plan.append((rb, mid, pb, pe, _capi.BLOCK_MIRROR_ALL))
        else:                             # the higher rank takes all of its rows against the second half of p's
            mid = pb + (pe - pb + 1) // 2
            if re > rb and pe > mid:
                plan.append((rb, re, mid, pe, _capi.BLOCK_MIRROR_ALL))
    return plan


class GpuOps:
    """numeric back end on one MI355X: everything is a call into libmvs_hip.so"""

    def __init__(self, ctx, device):
        import torch
        self.ctx, self.device = ctx, device
        self.k2_ms = 0.0
        # buffers here are torch tensors (zero fills, copies and slices run on torch's stream): the library
        # must issue its kernels on that same stream or nothing orders them against each other
        ctx.set_stream(torch.cuda.current_stream(torch.device(device)))

    def new_cells(self, capacity):
        import torch
        return torch.empty((capacity, 4), dtype=torch.int32, device=self.device)

    def open_set(self, planes, n, n_alloc, d, d_pad, limbs):
        return self.ctx.sketch_set_from_planes(planes, n, n_alloc, d, d_pad, limbs)

    def close_set(self, sset):
        sset.close()

    def compare_block(self, sset, norms_sq, rb, re, cb, ce, flags, keep_mode, raw, n_raw):
        n = self.ctx.pairwise_block(sset, norms_sq, rb, re, cb, ce, flags, raw, n_raw, keep_mode=keep_mode)
        try:
            self.k2_ms += self.ctx.kernel_ms(1)
        except _capi.MvsError:
            pass
        return n

    def sort_cells(self, cells_in, n, cells_out):
        self.ctx.cells_sort(cells_in, n, cells_out)

    def max_abs(self, sketches):
        return self.ctx.max_abs(sketches)

    def limbs_for(self, max_abs):
        return _capi.limbs_for_max_abs(max_abs)

    def limb_geometry(self, n, d, limbs):
        return self.ctx.limb_geometry(n, d, limbs)

    def new_planes(self, nbytes):
        import torch
        return torch.zeros(nbytes, dtype=torch.int8, device=self.device)

    def limb_split(self, sketches, limbs, planes, d_pad, row_offset):
        self.ctx.limb_split(sketches, limbs, planes, d_pad, row_offset)

    def to_device(self, host_array):
        import torch
        return torch.from_numpy(np.ascontiguousarray(host_array)).to(self.device)

    def compare(self, planes, n, n_alloc, d, d_pad, limbs, norms_sq, row_begin, row_end, keep_mode, cells_out):
        sset = self.ctx.sketch_set_from_planes(planes, n, n_alloc, d, d_pad, limbs)
        try:
            return self.ctx.pairwise_rows(sset, norms_sq, row_begin=row_begin, row_end=row_end,
                                          keep_mode=keep_mode, cells_out=cells_out)
        finally:
            sset.close()


class ShardedComparison:
    """State that survives between steps (the gathered plane buffer is reused while its geometry holds).

    run() is the whole exchange + comparison for local rows that are complete.  begin() / feed() / finish() do the same
    for local rows that become final in PARTS: the limb planes of a finished part are handed to the all-gather at once
    (mvs_allgather_rows, on the collectives' stream), so the exchange of part k runs beside the projection of part
    k + 1.  The limb code has to be fixed before the first part is known in full, so the parts are coded with
    `limbs_guess` (two base-256 limbs: |v| <= 32639, what sketches of up to tens of millions of hashes need) and the
    all-reduce of max|v| at the end verifies the guess on all ranks; if it does not hold, finish() falls back to run()
    -- same result, no overlap."""

    def __init__(self, ops, rank=0, world=1, dist=None, collectives=None):
        """dist: a torch.distributed module with an initialised default group, or collectives: a
        TorchCollectives / NativeCollectives object (takes precedence)."""
        self.ops, self.rank, self.world = ops, rank, world
        self.coll = collectives if collectives is not None else (TorchCollectives(dist, rank, world) if dist is not None else None)
        self._planes = None
        self._key = None
        self._raw = self._tmp = self._xraw = None
        self._n2 = None
        self.time_gather = False         # bench: torch events around the all-gathers (read with last_gather_ms())
        self._ev = None
        self._step = None
        self.symmetric = os.environ.get("MVS_SHARDED_SYMMETRIC", "1") != "0"
        if world > 1 and self.coll is None:
            raise ValueError("world > 1 needs torch.distributed or a communicator")

    def last_gather_ms(self):
        """span from the first to the last all-gather of the last step on the exchange stream (0 with one rank);
        synchronises on the end event"""
        if self._ev is None:
            return 0.0
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])

    # ---- stream order: compute stream <-> the collectives' stream (when they differ) ----
    def _side(self):
        return getattr(self.coll, "stream", None) if self.coll is not None else None

    def _exchange(self, fn):
        """fn() issues collectives: they see everything the compute stream has queued so far, and the compute stream
        sees their result"""
        side = self._side()
        if side is None:
            return fn()
        import torch
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        out = fn()
        main.wait_stream(side)
        return out

    def _mark(self, which):
        if not self.time_gather or self.world == 1:
            return
        import torch
        if which == 0:
            self._ev = [torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)]
        side = self._side()
        self._ev[which].record(side if side is not None else torch.cuda.current_stream())

    def _agree(self, status):
        """every rank learns whether any rank failed (a rank that raises alone would leave the others inside the
        next collective until the launcher's timeout): returns the largest status"""
        if self.world == 1:
            return status
        return self.coll.allreduce_max(status, self._planes)

    def _prepare(self, n_total, d, limbs):
        """plane buffer + norm buffer of the global geometry -> (rps, n_alloc, d_pad)"""
        ops, world = self.ops, self.world
        rps = (n_total + world - 1) // world            # rows per shard = block size of the all-gather
        n_rows_global = rps * world                      # >= n_total; the tail rows stay zero
        n_alloc, d_pad, nbytes = ops.limb_geometry(n_rows_global, d, limbs)
        key = (limbs, n_alloc, d_pad, n_rows_global)
        if self._key != key:
            self._planes = ops.new_planes(nbytes)
            self._n2 = ops.to_device(np.zeros(n_rows_global, dtype=np.float64))
            self._key = key
        return rps, n_alloc, d_pad

    def _put_norms(self, norms_sq_part, first, count, rps):
        """norms of local rows [first, first + count) into this rank's block of the gathered norm buffer"""
        n2_all, base = self._n2, self.rank * rps + first
        if _capi._is_torch(norms_sq_part) and not _capi._is_torch(n2_all):
            norms_sq_part = norms_sq_part.cpu().numpy()      # host back end (tests): plain arrays
        if _capi._is_torch(norms_sq_part):
            n2_all[base:base + count].copy_(norms_sq_part)   # already on the device: no host round trip
        else:
            n2_all[base:base + count] = self.ops.to_device(np.ascontiguousarray(norms_sq_part, dtype=np.float64))

    def run(self, sketches_local, norms_sq_local, n_total, keep_mode=_capi.KEEP_INT32, cells_out=None,
            max_abs_local=None):
        """sketches_local: this rank's rows (int32/int16 [n_local, d]); norms_sq_local: float64 [n_local], a host
        array or a device tensor; max_abs_local: largest |v| of sketches_local if the caller already has it
        (Context.stats).  Returns (cells, n_cells, info) for this rank's shard.
        cells_out (device [capacity, 4] int32): capacity only has to hold this shard's cells; what the symmetric
        schedule has in flight on top of them (mirrored cells on their way to other ranks) lives in internal buffers."""
        ops, rank, world = self.ops, self.rank, self.world
        n_local, d = sketches_local.shape
        rb, re = shard_rows(n_total, world, rank)
        if re - rb != n_local:
            raise ValueError("rank %d holds %d rows but its shard is [%d,%d)" % (rank, n_local, rb, re))
        if max_abs_local is not None:
            max_abs = int(max_abs_local)
        else:
            max_abs = ops.max_abs(sketches_local) if n_local else 0
        if world > 1:
            max_abs = self.coll.allreduce_max(max_abs, sketches_local)
        limbs = ops.limbs_for(max_abs)
        rps, n_alloc, d_pad = self._prepare(n_total, d, limbs)
        planes = self._planes
        if n_local:
            ops.limb_split(sketches_local, limbs, planes, d_pad, rank * rps)
            self._put_norms(norms_sq_local, 0, n_local, rps)
        if n_local < rps:
            self._n2[rank * rps + n_local:(rank + 1) * rps] = 0      # the block's tail rows: zero sketches, zero norms
        if world > 1:
            def gather():
                self._mark(0)
                self.coll.allgather_rows(planes, rps, 0, rps, limbs, d_pad)   # int8 row blocks: 2 B per entry at two limbs instead of 4
                self.coll.allgather_blocks(self._n2, rps)
                self._mark(1)
            self._exchange(gather)
        return self._compare(n_total, d, limbs, rps, n_alloc, d_pad, rb, re, keep_mode, cells_out, overlapped=False)

    # ---- local rows arriving in parts ----
    def begin(self, sketches_local, norms_sq_local, n_total, limbs_guess=2):
        """sketches_local / norms_sq_local: the rank's FULL buffers ([n_local, d] / [n_local]); their rows become valid
        part by part (feed).  Nothing is read here."""
        n_local, d = sketches_local.shape
        rb, re = shard_rows(n_total, self.world, self.rank)
        if re - rb != n_local:
            raise ValueError("rank %d holds %d rows but its shard is [%d,%d)" % (self.rank, n_local, rb, re))
        rps, n_alloc, d_pad = self._prepare(n_total, d, limbs_guess)
        if n_local < rps:
            self._n2[self.rank * rps + n_local:(self.rank + 1) * rps] = 0
        self._step = {"sk": sketches_local, "n2": norms_sq_local, "n_total": n_total, "limbs": limbs_guess, "rps": rps,
                      "n_alloc": n_alloc, "d_pad": d_pad, "max_abs": 0, "fed": 0, "parts": 0}

    def part_bounds(self, n_total, parts):
        """[(row_begin, row_end)] cutting a rank's block of ceil(n_total / world) rows into `parts` pieces: the SAME
        bounds on every rank (a collective's sizes must agree); a rank whose shard is shorter than the block simply has
        fewer -- or no -- rows of its own inside the later pieces"""
        rps = (n_total + self.world - 1) // self.world
        step = max(1, (rps + parts - 1) // parts)
        return [(b, min(b + step, rps)) for b in range(0, max(rps, 1), step)]

    def feed(self, row_begin, row_end, max_abs_part):
        """rows [row_begin, row_end) of this rank's BLOCK (block coordinates: 0 .. ceil(n_total / world), the same
        bounds on every rank, in order -- part_bounds()) are final: those of them the rank owns (below its n_local) have
        been written to the buffers given to begin(); max_abs_part: their largest |v|.  Their planes are coded and, with
        more than one rank, the exchange of exactly this row range of every rank's block starts now."""
        st = self._step
        if st is None or row_begin != st["fed"] or row_end < row_begin or row_end > st["rps"]:
            raise ValueError("feed(%d, %d) out of order" % (row_begin, row_end))
        st["fed"], st["parts"] = row_end, st["parts"] + 1
        st["max_abs"] = max(st["max_abs"], int(max_abs_part))
        rps, limbs, d_pad = st["rps"], st["limbs"], st["d_pad"]
        n_local = st["sk"].shape[0]
        own_b, own_e = min(row_begin, n_local), min(row_end, n_local)
        if own_e > own_b:
            self.ops.limb_split(st["sk"][own_b:own_e], limbs, self._planes, d_pad, self.rank * rps + own_b)
            self._put_norms(st["n2"][own_b:own_e], own_b, own_e - own_b, rps)
        if self.world > 1 and row_end > row_begin:
            def gather():
                if st["parts"] == 1:
                    self._mark(0)
                self.coll.allgather_rows(self._planes, rps, row_begin, row_end - row_begin, limbs, d_pad)
            self._exchange_begin(gather)

    def _exchange_begin(self, fn):
        """like _exchange, but the compute stream does NOT wait: what it queues next (the next part's projection) runs
        beside the exchange; finish() joins the streams"""
        side = self._side()
        if side is None:
            return fn()
        import torch
        side.wait_stream(torch.cuda.current_stream())
        return fn()

    def finish(self, keep_mode=_capi.KEEP_INT32, cells_out=None):
        st, self._step = self._step, None
        if st is None or st["fed"] != st["rps"]:
            raise ValueError("finish() before the whole block was fed")
        world, rps, limbs = self.world, st["rps"], st["limbs"]
        max_abs = st["max_abs"]
        if world > 1:
            max_abs = self.coll.allreduce_max(max_abs, st["sk"])
        need = self.ops.limbs_for(max_abs)
        # planes coded with MORE base-256 limbs than the data needs are still exact (the high limbs are zero); only a guess
        # that is too small -- or a guess / need in the three-plane Karatsuba code, which is a different encoding -- forces
        # the step to be redone (max|v| <= 127 against the guess of two limbs used to redo split and exchange)
        plain = need <= 4 and limbs <= 4
        if (need > limbs) if plain else (need != limbs):
            # the guess does not hold (on some rank): every rank takes this branch and redoes the step the plain way
            if self._side() is not None:
                import torch
                torch.cuda.current_stream().wait_stream(self._side())
            cells, cnt, info = self.run(st["sk"], st["n2"], st["n_total"], keep_mode=keep_mode, cells_out=cells_out,
                                        max_abs_local=st["max_abs"])
            info["overlap"] = "limb guess %d did not hold: plain exchange" % limbs
            return cells, cnt, info
        if world > 1:
            def gather():
                self.coll.allgather_blocks(self._n2, rps)
                self._mark(1)
            self._exchange(gather)           # joins the exchange stream: every part's rows are in place after this
        rb, re = shard_rows(st["n_total"], world, self.rank)
        return self._compare(st["n_total"], st["sk"].shape[1], limbs, rps, st["n_alloc"], st["d_pad"], rb, re, keep_mode,
                             cells_out, overlapped=world > 1 and st["parts"] > 1)

    def _compare(self, n_total, d, limbs, rps, n_alloc, d_pad, rb, re, keep_mode, cells_out, overlapped):
        ops, world, planes = self.ops, self.world, self._planes
        # rows beyond n_total are zero sketches with zero norms: they can never be kept
        n2_dev = self._n2[:n_total]
        info = {"limbs": limbs, "rows": (rb, re), "allgather_bytes_per_rank": rps * (limbs & 0xff) * d_pad if world > 1 else 0,
                "collectives": self.coll.kind if world > 1 else "none",
                "overlap": "exchange of a part beside the projection of the next" if overlapped else "none"}
        if world > 1 and self.symmetric and cells_out is not None:
            cells, cnt = self._run_symmetric(planes, n_total, n_alloc, d, d_pad, limbs, n2_dev, rb, re, keep_mode,
                                             cells_out, info)
        else:
            status, err = 0, None
            cells, cnt = None, 0
            try:
                cells, cnt = ops.compare(planes, n_total, n_alloc, d, d_pad, limbs, n2_dev, rb, re, keep_mode, cells_out)
            except _capi.MvsError as e:
                status, err = e.code, e
            if self._agree(status):
                raise err if err is not None else _capi.MvsError(_capi.MVS_E_HIP, "another rank failed in the comparison")
        return cells, cnt, info

    def _run_symmetric(self, planes, n_total, n_alloc, d, d_pad, limbs, n2_dev, rb, re, keep_mode, cells_out, info):
        """every unordered pair of row blocks once + exchange of the mirrored cells (module docstring)"""
        import torch
        ops, rank, world = self.ops, self.rank, self.world
        cap = cells_out.shape[0]
        # raw / tmp hold what this rank's blocks produce: its own cells AND the mirrored ones on their way to other
        # ranks (about as many again), so they are sized from the caller's capacity with that in mind and regrown on
        # demand -- only the final shard has to fit cells_out
        want = 2 * cap + 1024
        status, err, n_raw = 0, None, 0
        plan = block_plan(n_total, world, rank)
        for attempt in range(3):
            if self._raw is None or self._raw.shape[0] < want:
                self._raw, self._tmp = ops.new_cells(want), ops.new_cells(want)
            raw, tmp = self._raw, self._tmp
            status, err, n_raw = 0, None, 0
            sset = ops.open_set(planes, n_total, n_alloc, d, d_pad, limbs)
            try:
                for (b0, b1, c0, c1, flags) in plan:
                    n_raw = ops.compare_block(sset, n2_dev, b0, b1, c0, c1, flags, keep_mode, raw, n_raw)
            except _capi.MvsError as e:          # tell the others before anybody enters the exchange
                status, err = e.code, e
            finally:
                ops.close_set(sset)
            needed = getattr(err, "needed", None)
            if status == _capi.MVS_E_CAPACITY and needed and attempt < 2:
                want = int(needed) + int(needed) // 4 + 1024      # the library reported the count so far: retry, local to this rank
                continue
            break
        if self._agree(status):
            raise err if err is not None else _capi.MvsError(_capi.MVS_E_HIP, "another rank failed in its block comparisons")
        ops.sort_cells(raw, n_raw, tmp)                     # (row, col) order: own rows form one contiguous run
        rows = tmp[:n_raw, 0].contiguous()
        lo, hi = torch.searchsorted(rows, torch.tensor([rb, re], dtype=rows.dtype, device=rows.device)).tolist()   # sync 1
        n_local, n_foreign = hi - lo, n_raw - (hi - lo)     # mirrored cells of rows other ranks own: before lo / after hi
        max_f = max(self.coll.allreduce_max(n_foreign, rows), 1)
        # exchange: all-gather of the padded foreign lists; every rank keeps the cells of its own rows
        recv = torch.full((world * max_f, 4), 2147483647, dtype=tmp.dtype, device=tmp.device)
        mine = recv[rank * max_f:(rank + 1) * max_f]
        mine[:lo] = tmp[:lo]
        mine[lo:n_foreign] = tmp[hi:n_raw]
        self._exchange(lambda: self.coll.allgather_blocks(recv.view(-1), max_f * 4))
        wanted = (recv[:, 0] >= rb) & (recv[:, 0] < re)
        got = recv[wanted]                                  # sync 2: compacted, only the cells of this shard's rows
        n_out = n_local + got.shape[0]
        status = _capi.MVS_E_CAPACITY if n_out > cap else 0
        if self._agree(status):
            raise _capi.MvsError(_capi.MVS_E_CAPACITY, "%d cells for this shard but capacity is %d" % (n_out, cap))
        if self._xraw is None or self._xraw.shape[0] < n_out:
            self._xraw = ops.new_cells(n_out + n_out // 4 + 1024)
        self._xraw[:n_local] = tmp[lo:hi]
        self._xraw[n_local:n_out] = got
        ops.sort_cells(self._xraw, n_out, cells_out)
        info.update({"blocks": len(plan), "exchanged_cells": int(n_foreign), "schedule": "symmetric"})
        return cells_out, n_out
