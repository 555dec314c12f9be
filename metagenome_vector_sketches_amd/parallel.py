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
"""
import numpy as np

from . import _capi


def shard_rows(n_total, world, rank):
    """src/pairwise_comp_optimized.cpp:938-940"""
    rps = (n_total + world - 1) // world
    b = min(rank * rps, n_total)
    return b, min(b + rps, n_total)


class GpuOps:
    """numeric back end on one MI355X: everything is a call into libmvs_hip.so"""

    def __init__(self, ctx, device):
        self.ctx, self.device = ctx, device

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

    def __init__(self, ops, rank=0, world=1, dist=None):
        self.ops, self.rank, self.world, self.dist = ops, rank, world, dist
        self._planes = None
        self._key = None
        if world > 1 and dist is None:
            raise ValueError("world > 1 needs torch.distributed")

    def run(self, sketches_local, norms_sq_local, n_total, keep_mode=_capi.KEEP_INT32, cells_out=None,
            max_abs_local=None):
        """sketches_local: this rank's rows (int32/int16 [n_local, d]); norms_sq_local: host float64
        [n_local]; max_abs_local: largest |v| of sketches_local if the caller already has it
        (Context.stats).  Returns (cells, n_cells, info) for this rank's shard."""
        ops, dist, rank, world = self.ops, self.dist, self.rank, self.world
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
            import torch
            t = ops.to_device(np.array([max_abs], dtype=np.int64))
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            max_abs = int(t.cpu()[0])
        limbs = ops.limbs_for(max_abs)
        n_rows_global = rps * world                      # >= n_total; the tail rows stay zero
        n_alloc, d_pad, nbytes = ops.limb_geometry(n_rows_global, d, limbs)
        key = (limbs, n_alloc, d_pad)
        if self._key != key:
            self._planes = ops.new_planes(nbytes)
            self._key = key
        planes = self._planes
        blk = rps * limbs * d_pad
        if n_local:
            ops.limb_split(sketches_local, limbs, planes, d_pad, rank * rps)
        n2_pad = np.zeros(rps, dtype=np.float64)
        n2_pad[:n_local] = norms_sq_local
        if world > 1:
            mine = planes[rank * blk:(rank + 1) * blk].clone()
            dist.all_gather_into_tensor(planes[:world * blk], mine)
            n2_all = ops.to_device(np.zeros(rps * world, dtype=np.float64))
            dist.all_gather_into_tensor(n2_all, ops.to_device(n2_pad))
        else:
            n2_all = ops.to_device(n2_pad)
        # rows beyond n_total are zero sketches with zero norms: they can never be kept
        cells, cnt = ops.compare(planes, n_total, n_alloc, d, d_pad, limbs, n2_all[:n_total].contiguous()
                                 if hasattr(n2_all, "contiguous") else n2_all[:n_total], rb, re, keep_mode, cells_out)
        return cells, cnt, {"limbs": limbs, "rows": (rb, re), "allgather_bytes_per_rank": blk if world > 1 else 0}
