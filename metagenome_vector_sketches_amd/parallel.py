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
    """the three collectives of the exchange over torch.distributed"""
    kind = "torch.distributed"

    def __init__(self, dist, rank, world):
        self.dist, self.rank, self.world = dist, rank, world

    def allreduce_max(self, value, like):
        import torch
        t = torch.tensor([int(value)], dtype=torch.int64, device=getattr(like, "device", "cpu"))
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return int(t.cpu()[0])

    def allgather_blocks(self, buf, block_elems):
        """buf: 1-D tensor of world * block_elems elements whose block `rank` is filled in"""
        send = buf[self.rank * block_elems:(self.rank + 1) * block_elems]
        if self._send is None or self._send.shape != send.shape or self._send.dtype != send.dtype:
            self._send = send.new_empty(send.shape)
        self._send.copy_(send)     # a separate send block: NCCL / gloo need not support aliased in-place gathers
        self.dist.all_gather_into_tensor(buf[:self.world * block_elems], self._send)

    _send = None


class NativeCollectives:
    """the same over the communicator of the C ABI (in place on the device, on the context's stream)"""

    def __init__(self, comm):
        self.comm, self.rank, self.world = comm, comm.rank, comm.world
        self.kind = "libmvs_hip mvs_comm (%s)" % ("RCCL" if comm.is_rccl else "file transport")

    def allreduce_max(self, value, like):
        return self.comm.allreduce_max(value)

    def allgather_blocks(self, buf, block_elems):
        self.comm.allgather_bytes(buf, block_elems * buf.element_size())


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
    """State that survives between steps (the gathered plane buffer is reused while its geometry holds)."""

    def __init__(self, ops, rank=0, world=1, dist=None, collectives=None):
        """dist: a torch.distributed module with an initialised default group, or collectives: a
        TorchCollectives / NativeCollectives object (takes precedence)."""
        self.ops, self.rank, self.world = ops, rank, world
        self.coll = collectives if collectives is not None else (TorchCollectives(dist, rank, world) if dist is not None else None)
        self._planes = None
        self._key = None
        self._raw = self._tmp = None
        self._n2 = None
        self.time_gather = False         # bench: torch events around the all-gathers (read with last_gather_ms())
        self._ev = None
        self.symmetric = os.environ.get("MVS_SHARDED_SYMMETRIC", "1") != "0"
        if world > 1 and self.coll is None:
            raise ValueError("world > 1 needs torch.distributed or a communicator")

    def last_gather_ms(self):
        """duration of the last step's all-gathers on the stream (0 with one rank); synchronises on the end event"""
        if self._ev is None:
            return 0.0
        self._ev[1].synchronize()
        return self._ev[0].elapsed_time(self._ev[1])

    def _agree(self, status):
        """every rank learns whether any rank failed (a rank that raises alone would leave the others inside the
        next collective until the launcher's timeout): returns the largest status"""
        if self.world == 1:
            return status
        return self.coll.allreduce_max(status, self._planes)

    def run(self, sketches_local, norms_sq_local, n_total, keep_mode=_capi.KEEP_INT32, cells_out=None,
            max_abs_local=None):
        """sketches_local: this rank's rows (int32/int16 [n_local, d]); norms_sq_local: float64 [n_local], a host
        array or a device tensor; max_abs_local: largest |v| of sketches_local if the caller already has it
        (Context.stats).  Returns (cells, n_cells, info) for this rank's shard."""
        ops, rank, world = self.ops, self.rank, self.world
        n_local, d = sketches_local.shape
        rb, re = shard_rows(n_total, world, rank)
        if re - rb != n_local:
            raise ValueError("rank %d holds %d rows but its shard is [%d,%d)" % (rank, n_local, rb, re))
        rps = (n_total + world - 1) // world            # rows per shard = block size of the all-gather
        if max_abs_local is not None:
            max_abs = int(max_abs_local)
        else:
            max_abs = ops.max_abs(sketches_local) if n_local else 0
        if world > 1:
            max_abs = self.coll.allreduce_max(max_abs, sketches_local)
        limbs = ops.limbs_for(max_abs)
        n_rows_global = rps * world                      # >= n_total; the tail rows stay zero
        n_alloc, d_pad, nbytes = ops.limb_geometry(n_rows_global, d, limbs)
        key = (limbs, n_alloc, d_pad)
        if self._key != key:
            self._planes = ops.new_planes(nbytes)
            self._n2 = ops.to_device(np.zeros(rps * world, dtype=np.float64))
            self._key = key
        planes, n2_all = self._planes, self._n2
        blk = rps * limbs * d_pad
        if n_local:
            ops.limb_split(sketches_local, limbs, planes, d_pad, rank * rps)
        if _capi._is_torch(norms_sq_local) and not _capi._is_torch(n2_all):
            norms_sq_local = norms_sq_local.cpu().numpy()    # host back end (tests): plain arrays
        if _capi._is_torch(norms_sq_local):
            # already on the device (Context.norms_sq_text): no host round trip; the block's tail rows stay zero
            n2_all[rank * rps:rank * rps + n_local].copy_(norms_sq_local)
            if n_local < rps:
                n2_all[rank * rps + n_local:(rank + 1) * rps].zero_()
        else:
            n2_pad = np.zeros(rps, dtype=np.float64)
            n2_pad[:n_local] = norms_sq_local
            n2_all[rank * rps:(rank + 1) * rps] = ops.to_device(n2_pad)
        if world > 1:
            if self.time_gather:
                import torch
                self._ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
                self._ev[0].record()
            self.coll.allgather_blocks(planes, blk)     # int8 row blocks: 2 B per entry at two limbs instead of 4
            self.coll.allgather_blocks(n2_all, rps)
            if self.time_gather:
                self._ev[1].record()
        # rows beyond n_total are zero sketches with zero norms: they can never be kept
        n2_dev = n2_all[:n_total]
        info = {"limbs": limbs, "rows": (rb, re), "allgather_bytes_per_rank": blk if world > 1 else 0,
                "collectives": self.coll.kind if world > 1 else "none"}
        if world > 1 and self.symmetric and cells_out is not None:
            cells, cnt = self._run_symmetric(planes, n_total, n_alloc, d, d_pad, limbs, n2_dev, rb, re, keep_mode,
                                             cells_out, info)
        else:
            status, err = 0, None
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
        if self._raw is None or self._raw.shape[0] != cap:
            self._raw, self._tmp = ops.new_cells(cap), ops.new_cells(cap)
        raw, tmp = self._raw, self._tmp
        status, err = 0, None
        n_raw = 0
        plan = block_plan(n_total, world, rank)
        sset = ops.open_set(planes, n_total, n_alloc, d, d_pad, limbs)
        try:
            for (b0, b1, c0, c1, flags) in plan:
                n_raw = ops.compare_block(sset, n2_dev, b0, b1, c0, c1, flags, keep_mode, raw, n_raw)
        except _capi.MvsError as e:          # e.g. capacity: tell the others before anybody enters the exchange
            status, err = e.code, e
        finally:
            ops.close_set(sset)
        if self._agree(status):
            raise err if err is not None else _capi.MvsError(_capi.MVS_E_HIP, "another rank failed in its block comparisons")
        ops.sort_cells(raw, n_raw, tmp)                     # (row, col) order: own rows form one contiguous run
        rows = tmp[:n_raw, 0].contiguous()
        lo, hi = torch.searchsorted(rows, torch.tensor([rb, re], dtype=rows.dtype, device=rows.device)).tolist()   # sync 1
        n_local, n_foreign = hi - lo, n_raw - (hi - lo)     # mirrored cells of rows other ranks own: before lo / after hi
        max_f = max(self.coll.allreduce_max(n_foreign, rows), 1)
        # exchange: all-gather of the padded foreign lists; every rank keeps the cells of its own rows.  Unwanted
        # entries get row = INT32_MAX so that the final (row, col) sort pushes them behind the shard's cells.
        recv = torch.full((world * max_f, 4), 2147483647, dtype=tmp.dtype, device=tmp.device)
        mine = recv[rank * max_f:(rank + 1) * max_f]
        mine[:lo] = tmp[:lo]
        mine[lo:n_foreign] = tmp[hi:n_raw]
        self.coll.allgather_blocks(recv.view(-1), max_f * 4)
        wanted = (recv[:, 0] >= rb) & (recv[:, 0] < re)
        recv[:, 0] = torch.where(wanted, recv[:, 0], torch.full_like(recv[:, 0], 2147483647))
        n_mine = int(wanted.sum())                          # sync 2
        n_out = n_local + n_mine
        status = _capi.MVS_E_CAPACITY if (n_out > cap or n_local + world * max_f > cap) else 0
        if self._agree(status):
            raise _capi.MvsError(_capi.MVS_E_CAPACITY, "%d cells for this shard (%d in flight) but capacity is %d"
                                 % (n_out, n_local + world * max_f, cap))
        raw[:n_local] = tmp[lo:hi]
        raw[n_local:n_local + world * max_f] = recv
        ops.sort_cells(raw, n_local + world * max_f, cells_out)
        info.update({"blocks": len(plan), "exchanged_cells": int(n_foreign), "schedule": "symmetric"})
        return cells_out, n_out
