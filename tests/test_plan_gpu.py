"""GPU: block plans (include/mvs_hip.h "block plans") -- a rank's share of the symmetric multi-rank schedule as one
two-stage comparison in storage coordinates.  All G ranks of a split are played by ONE process on the one card: the
storage buffers hold every rank's block (what the all-gathers would have assembled), each rank's plan runs in turn, its
kept cells are routed, the mirrored ones are handed to their owners through the same send-buffer format the exchange uses.
The union must equal mvs_pairwise_rows / the oracle cell for cell (src/pairwise_comp_optimized.cpp:135-147, :654-665)."""
import numpy as np
import pytest
import torch

from metagenome_vector_sketches_amd import _capi, parallel, synth
from oracle import pyoracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("restore_options")]
DEV = "cuda:0"


@pytest.fixture(autouse=True)
def _on_torchs_stream(ctx):
    """the buffers here are torch tensors: the library must issue its kernels on the stream torch fills and reads them on"""
    ctx.set_stream(torch.cuda.current_stream())
    yield
    torch.cuda.synchronize()
    ctx.set_stream(None)


def _n2(sk):
    return np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(row))) for row in sk.astype(np.int32)])


def _want(sk, n2, keep_mode):
    cells = orc.pairwise_rows(sk, n2, chunk=192, threads=8) if keep_mode == _capi.KEEP_INT32 else None
    if cells is None:
        pytest.skip("oracle keep mode")
    cells = cells[np.lexsort((cells["col"], cells["row"]))]
    return np.stack([cells[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)


class Split:
    """storage buffers of a G-way split on one device, filled for ALL ranks"""

    def __init__(self, ctx, sk, n2, world, limbs=2):
        self.ctx, self.world = ctx, world
        n, d = sk.shape
        self.n, self.d = n, d
        self.rps, self.P = _capi.shard_layout(n, world)
        n_st = self.P * world
        self.n_alloc, self.d_pad, nbytes = ctx.limb_geometry(n_st, d, limbs)
        self.planes = torch.zeros(nbytes, dtype=torch.int8, device=DEV)
        self.coarse = torch.zeros(self.n_alloc * self.d_pad, dtype=torch.uint8, device=DEV)
        self.stats = torch.zeros(self.n_alloc * 16, dtype=torch.uint8, device=DEV)
        self.n2 = torch.zeros(self.n_alloc, dtype=torch.float64, device=DEV)
        self.sset = ctx.sketch_set_from_planes(self.planes, n_st, self.n_alloc, d, self.d_pad, limbs)
        ctx.attach_derived(self.sset, self.coarse, self.stats)
        for r in range(world):
            b, e = parallel.shard_rows(n, world, r)
            if e > b:
                ctx.limb_split(torch.from_numpy(sk[b:e].copy()).to(DEV), limbs, self.planes, self.d_pad, r * self.P)
                self.n2[r * self.P:r * self.P + (e - b)] = torch.from_numpy(n2[b:e].copy()).to(DEV)
            ctx.prepare_rows(self.sset, r * self.P, self.P)

    def rank_cells(self, rank, symmetric=True, chunks=1, keep_mode=_capi.KEEP_INT32, cap=None, cap_f=None, wire=False):
        """-> (own cells [k, 4] in sample indices, the send buffer as the exchange would carry it, plan statistics).
        wire: the other ranks' limb planes are NOT there (poisoned) -- the plan rebuilds the rows it reads from their low limbs"""
        ctx, P, world = self.ctx, self.P, self.world
        saved = None
        if wire:
            lo = torch.zeros(self.n_alloc * self.d_pad, dtype=torch.int8, device=DEV)
            parallel.GpuOps(ctx, DEV).wire_rows(self.planes, lo, self.d_pad, 0, P * world)
            saved = self.planes.clone()
            rows = self.planes.view(-1, 2, self.d_pad)
            rows[:rank * P] = 0x55
            rows[(rank + 1) * P:] = 0x55
        cap = cap or max(4096, 400 * self.n)
        raw = torch.empty((2 * cap, 4), dtype=torch.int32, device=DEV)
        own = torch.empty((cap, 4), dtype=torch.int32, device=DEV)
        b, e = parallel.shard_rows(self.n, world, rank)
        d_own = torch.full((2 + (e - b + 2) // 2,), -1, dtype=torch.int64, device=DEV)    # the state block: route clears it
        plan = parallel.block_plan(world, rank, P, symmetric=symmetric)
        mirror = symmetric and world > 1
        ctx.plan_begin(self.sset, self.n2, rank * P, (rank + 1) * P, mirror, raw, keep_mode=keep_mode)
        if wire:
            ctx.plan_wire(lo)
        ctx.plan_filter(plan[:1])
        if rank % 2 == 1:                   # odd ranks announce every row at once (one launch derives the peers' filter constants:
            ctx.plan_rows_ready(0, P * world)   # mvs_plan_rows_ready), even ranks leave it to mvs_plan_filter, block by block
        for (c0, c1) in parallel.chunk_bounds(P, chunks):
            blocks = parallel.clip_blocks(plan[1:], P, c0, c1)
            if blocks:
                ctx.plan_filter(blocks)
        d_cnt = ctx.plan_finish()
        cap_f = cap if cap_f is None else cap_f
        send = torch.zeros(_capi.CELLS_HEADER_BYTES + 16 * cap_f, dtype=torch.uint8, device=DEV)
        ctx.cells_route(raw, d_cnt, P, self.rps, self.n, b, e, own, d_own, send, cap_f, status=0, max_abs=123 + rank)
        n_own, heads = ctx.cells_report(send, 1, cap_f, e - b, d_own)[:2]
        if saved is not None:
            self.planes.copy_(saved)
        assert heads[0][1] == 0 and heads[0][2] == 123 + rank and (heads[0][3] <= heads[0][4] or cap < 4096)
        return own[:n_own].cpu().numpy(), send, ctx.plan_stats(), heads[0], own, d_own


def _union(split, symmetric=True, chunks=1, keep_mode=_capi.KEEP_INT32, wire=False):
    """every rank's shard = its own cells + what the other ranks' send buffers hold for its rows (mvs_cells_collect)"""
    ctx, world = split.ctx, split.world
    per_rank = [split.rank_cells(r, symmetric, chunks, keep_mode, wire=wire) for r in range(world)]
    cap_f = (per_rank[0][1].numel() - _capi.CELLS_HEADER_BYTES) // 16
    recv = torch.cat([x[1] for x in per_rank])
    shards = []
    for r in range(world):
        b, e = parallel.shard_rows(split.n, world, r)
        own, d_own = per_rank[r][4], per_rank[r][5]                  # the rank's own cells and its state block, as routed
        ctx.cells_collect(recv, world, r, cap_f, b, e, own, d_own)
        n_out, heads, max_row = ctx.cells_report(recv, world, cap_f, e - b, d_own)
        assert [h[2] for h in heads] == [123 + k for k in range(world)] and n_out <= own.shape[0]
        got = own[:n_out].cpu().numpy()
        assert np.all((got[:, 0] >= b) & (got[:, 0] < e))
        assert max_row == (np.bincount(got[:, 0] - b).max() if n_out else 0)
        out = torch.empty((max(n_out, 1), 4), dtype=torch.int32, device=DEV)
        if n_out:
            ctx.cells_sort(own, n_out, out)
            if max_row <= 64:                                            # the row-bucket sort: the same order
                out2 = torch.zeros_like(out)
                ctx.cells_sort_rows(own, n_out, b, e, d_own, out2)
                assert torch.equal(out2[:n_out], out[:n_out])
                # ... and queued before the host knows the count (the count is read on the device, the buffers' sizes bound it)
                out3 = torch.zeros_like(out)
                ctx.cells_sort_rows_ahead(own, b, e, d_own, out3)
                assert torch.equal(out3[:n_out], out[:n_out])
                if n_out > 8:                                            # an output buffer that is too small: whole rows or nothing
                    small = torch.full((n_out // 2, 4), -7, dtype=torch.int32, device=DEV)
                    ctx.cells_sort_rows_ahead(own, b, e, d_own, small)
                    ends = np.cumsum(np.bincount(out[:n_out, 0].cpu().numpy() - b, minlength=e - b))
                    fit = int(ends[ends <= n_out // 2].max()) if np.any(ends <= n_out // 2) else 0
                    assert torch.equal(small[:fit], out[:fit])
        shards.append(out[:n_out].cpu().numpy())
    return np.concatenate(shards), per_rank


@pytest.mark.parametrize("n,d,world", [(700, 512, 1), (700, 512, 2), (1300, 256, 3), (2100, 256, 4), (2100, 128, 8),
                                       (5000, 256, 5), (3000, 2048, 4)])
@pytest.mark.parametrize("mode", ["two_stage", "exact"])
def test_plans_of_all_ranks_give_the_whole_matrix(ctx, n, d, world, mode):
    ctx.set_option("pairwise_filter", 2 if mode == "two_stage" else 0)
    sk = synth.make_sketches_numpy(n, d, 3000, seed=n + world, cluster=8)
    n2 = _n2(sk)
    want = _want(sk, n2, _capi.KEEP_INT32)
    split = Split(ctx, sk, n2, world)
    got, per_rank = _union(split, symmetric=True, chunks=1)
    assert len(want) > 4 * n and np.array_equal(got, want)
    st = per_rank[0][2]
    assert st["exact_mode"] == (mode == "exact")
    if mode == "two_stage":
        assert st["filter_launches"] >= 1 and st["filter_tiles"] > 0 and st["candidates"] + st["flagged_tiles"] > 0
    # the peers' columns arriving in chunks (one filter launch per chunk), and the plain rows x all-columns schedule
    got2, _ = _union(split, symmetric=True, chunks=3)
    assert np.array_equal(got2, want)
    got3, per3 = _union(split, symmetric=False, chunks=2)
    assert np.array_equal(got3, want)
    assert all(x[3][0] == 0 for x in per3)                       # nothing to send anywhere
    # the static super-patch map instead of the balanced tile order (option plan_order, default 1): placement only -- the same
    # tiles, candidates and cells
    with ctx.options(plan_order=0):
        got4, per4 = _union(split, symmetric=True, chunks=1)
    assert np.array_equal(got4, want)
    for a, b in zip(per_rank, per4):
        assert (a[2]["filter_tiles"], a[2]["candidates"], a[2]["flagged_tiles"]) == (b[2]["filter_tiles"], b[2]["candidates"], b[2]["flagged_tiles"])


def test_plan_on_dense_clusters_goes_through_flagged_tiles(ctx):
    """clusters of 600 related samples: whole 256 x 256 tiles are dense, on the diagonal blocks AND in the peers' blocks --
    those tiles go to the exact kernel (flagged), the rest through the re-check; mirror rules inside / outside the square"""
    n, d, world = 2600, 256, 4
    sk = synth.make_sketches_numpy(n, d, 3000, seed=77, cluster=600)
    n2 = _n2(sk)
    want = _want(sk, n2, _capi.KEEP_INT32)
    assert len(want) > 100 * n
    ctx.set_option("pairwise_filter", 2)
    split = Split(ctx, sk, n2, world)
    got, per_rank = _union(split)
    assert sum(x[2]["flagged_tiles"] for x in per_rank) > world       # off-diagonal tiles too
    assert np.array_equal(got, want)
    ctx.set_option("tile_dense_thr", 0)                               # nothing is flagged: every candidate re-checked
    got2, per2 = _union(split)
    assert sum(x[2]["flagged_tiles"] for x in per2) == 0 and np.array_equal(got2, want)


def test_plan_int16_keep_mode_and_overflow_reporting(ctx):
    n, d, world = 900, 256, 2
    sk = synth.make_sketches_numpy(n, d, 3000, seed=5, cluster=8)
    n2 = _n2(sk)
    ctx.set_option("pairwise_filter", 2)
    split = Split(ctx, sk, n2, world)
    # the floating keep test of the int16 path (src/pairwise_comp_optimized_16bits.cpp:218) against mvs_pairwise_rows
    plain = ctx.sketch_set(sk)
    ref, cnt = ctx.pairwise_rows(plain, n2, keep_mode=_capi.KEEP_INT16)
    plain.close()
    ref = ref[np.lexsort((ref["col"], ref["row"]))]
    want = np.stack([ref[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    got, _ = _union(split, keep_mode=_capi.KEEP_INT16)
    assert np.array_equal(got, want)
    # a raw list and a send buffer that are too small: nothing is written out of bounds, the header says how much there was
    own, send, st, head = split.rank_cells(0, cap=64, cap_f=16)[:4]
    assert head[3] > head[4] == 128                                    # raw cells > raw capacity
    own, send, st, head = split.rank_cells(0, cap_f=16)[:4]
    assert head[0] > 16 and head[3] <= head[4]                         # more mirror images than the buffer holds


def test_plan_rejects_what_it_cannot_run(ctx):
    n, d = 600, 128
    sk = synth.make_sketches_numpy(n, d, 3000, seed=1, cluster=8)
    split = Split(ctx, sk, _n2(sk), 2)
    ctx.set_option("pairwise_filter", 2)
    raw = torch.empty((1024, 4), dtype=torch.int32, device=DEV)
    with pytest.raises(_capi.MvsError):
        ctx.plan_filter([(0, 256, 0, 256)])                            # no plan in progress
    with pytest.raises(_capi.MvsError):
        ctx.plan_rows_ready(0, split.P)                                # likewise
    ctx.plan_begin(split.sset, split.n2, 0, split.P, True, raw)
    with pytest.raises(_capi.MvsError):
        ctx.plan_rows_ready(0, 2 * split.P + 1)                        # rows beyond the sketch set
    with pytest.raises(_capi.MvsError):
        ctx.plan_rows_ready(split.P, 0)                                # an inverted range
    ctx.plan_rows_ready(0, 2 * split.P)                                # every row (the frame's own part is skipped); twice is fine
    ctx.plan_rows_ready(split.P, 2 * split.P)
    with pytest.raises(_capi.MvsError):
        ctx.plan_filter([(0, split.P, 128, split.P)])                  # not on the tile grid
    with pytest.raises(_capi.MvsError):
        ctx.plan_filter([(split.P, 2 * split.P, 0, split.P)])          # rows outside the frame
    with pytest.raises(_capi.MvsError):
        ctx.plan_filter([(0, split.P, 0, 2 * split.P)])                # columns straddle the frame's square
    ctx.plan_filter([(0, split.P, 0, split.P)])
    ctx.plan_finish()
    with pytest.raises(_capi.MvsError):
        ctx.plan_finish()


def test_sharded_comparison_single_rank_equals_pairwise_rows(ctx):
    """world 1 through parallel.ShardedComparison (what `bench.py --config 3 --gpus 1` times) against mvs_pairwise_rows"""
    n, d = 6000, 512
    sk = synth.make_sketches_numpy(n, d, 3000, seed=11, cluster=16)
    n2 = _n2(sk)
    plain = ctx.sketch_set(sk)
    ref, cnt = ctx.pairwise_rows(plain, n2)
    plain.close()
    want = np.stack([ref[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, torch.device(DEV)), 0, 1)
    out = torch.empty((cnt + 10, 4), dtype=torch.int32, device=DEV)
    for _ in range(2):
        _, got_n, info = sc.run(torch.from_numpy(sk).to(DEV), n2, n, cells_out=out)
        torch.cuda.synchronize()
        assert got_n == cnt and np.array_equal(out[:cnt].cpu().numpy(), want)
    host, got_n, _ = sc.run(torch.from_numpy(sk).to(DEV), n2, n)
    assert got_n == cnt and np.array_equal(host, ref[np.lexsort((ref["col"], ref["row"]))])


@pytest.mark.parametrize("d", [64, 100, 512, 1100, 2048, 2100, 4096, 5000])
@pytest.mark.parametrize("dtype", [np.int32, np.int16])
def test_recode_rows_equals_the_separate_passes(ctx, d, dtype):
    """k_recode_rows (one pass over the sketches) against k_limb_split + k_coarse_build + k_coarse_fm: byte-identical limb
    planes, fragment-major coarse plane and row statistics, including the zero rows behind the samples and sketch lengths that
    are not multiples of 16 / that have no fused kernel (d_pad > 4096: the call takes the separate passes itself)"""
    rng = np.random.default_rng(d)
    n, rows = 37, 48                                            # 37 samples in a range of 48 rows starting at row 32
    hi = 30000 if dtype == np.int16 else 32639
    sk = rng.integers(-hi, hi + 1, size=(n, d)).astype(dtype)
    sk[3] = 0
    sk[5] = rng.integers(-100, 101, size=d).astype(dtype)      # a one-limb row
    sk[7, :] = hi
    n_st = 128
    n_alloc, d_pad, nbytes = ctx.limb_geometry(n_st, d, 2)
    out = []
    for fused in (8, 16, 0):                                    # rows per workgroup of the fused kernel; 0: separate passes
        planes = torch.zeros(nbytes, dtype=torch.int8, device=DEV)
        coarse = torch.full((n_alloc * d_pad,), 0x55, dtype=torch.uint8, device=DEV)
        stats = torch.full((n_alloc * 16,), 0x55, dtype=torch.uint8, device=DEV)
        sset = ctx.sketch_set_from_planes(planes, n_st, n_alloc, d, d_pad, 2)
        ctx.attach_derived(sset, coarse, stats)
        dev_sk = torch.from_numpy(sk).to(DEV)
        if fused:
            with ctx.options(recode_rows_wg=fused):
                ctx.recode_rows(sset, dev_sk, 32, rows)
        else:
            ctx.limb_split(dev_sk, 2, planes, d_pad, 32)
            ctx.prepare_rows(sset, 32, rows)
        torch.cuda.synchronize()
        out.append((planes.cpu().numpy().copy(), coarse.cpu().numpy().copy(), stats.cpu().numpy().copy()))
        sset.close()
    for got in out[:2]:
        for a, b in zip(got, out[2]):
            assert np.array_equal(a, b)
    assert np.all(out[0][1][:32 * d_pad] == 0x55) and np.all(out[0][1][80 * d_pad:] == 0x55)     # nothing outside the range
    st = out[0][2][32 * 16:80 * 16].view(np.int32).reshape(rows, 4)
    assert np.all(st[n:] == [1, 0, 0, 0]) and st[3].tolist() == [1, 0, 0, 0] and st[5, 0] == 1 and st[7, 0] > 1


def test_plan_rectangle_beyond_one_dispatch_is_cut_into_strips(ctx):
    """A rectangle whose padded tile grid holds more workgroups than one dispatch takes (BASELINE configs[4] on one GPU: the
    1M x 1M diagonal block is 15 M workgroups) is cut into column strips of whole super-patch columns, launched in groups that
    fit.  With the limit lowered (option plan_strip_wgs) the same happens at test size: same cells, same tile count, more launches."""
    n, d = 9000, 512                                           # (d = 128 keeps 5 M chance pairs: more than the test's buffers)
    sk = synth.make_sketches_numpy(n, d, 3000, seed=9, cluster=6)
    n2 = _n2(sk)
    ctx.set_option("pairwise_filter", 2)
    split = Split(ctx, sk, n2, 2)
    want, per_rank = _union(split)
    launches = [x[2]["filter_launches"] for x in per_rank]
    tiles = [x[2]["filter_tiles"] for x in per_rank]
    try:
        ctx.set_option("plan_strip_wgs", 512)                  # 2 patch rows x 256 -> one patch column per strip
        got, per2 = _union(split)
    finally:
        ctx.set_option("plan_strip_wgs", 1 << 22)
    assert np.array_equal(got, want) and len(want) > 4 * n
    assert [x[2]["filter_tiles"] for x in per2] == tiles
    assert all(b > a for a, b in zip(launches, [x[2]["filter_launches"] for x in per2]))


def _plan_once(ctx, split, cap, speculate):
    """rank 0's plan of a one-rank split -> (cell count as the plan left it, sorted cells or None, plan statistics)"""
    raw = torch.empty((cap, 4), dtype=torch.int32, device=DEV)
    own = torch.empty((cap, 4), dtype=torch.int32, device=DEV)
    P, n = split.P, split.n
    d_own = torch.zeros(2 + (n + 2) // 2, dtype=torch.int64, device=DEV)
    send = torch.zeros(_capi.CELLS_HEADER_BYTES, dtype=torch.uint8, device=DEV)
    with ctx.options(plan_speculate=1 if speculate else 0):
        ctx.plan_begin(split.sset, split.n2, 0, P, False, raw)
    ctx.plan_filter(parallel.block_plan(1, 0, P))
    d_cnt = ctx.plan_finish()
    ctx.cells_route(raw, d_cnt, P, split.rps, n, 0, n, own, d_own, send, 0)
    n_own, heads, _ = ctx.cells_report(send, 1, 0, n, d_own)          # (brings the plan's counts along)
    count = int(heads[0][3])
    st = ctx.plan_stats()
    cells = None
    if count < _capi.PLAN_STALE:
        assert n_own == count <= cap
        cells = own[:n_own]
        order = torch.argsort(cells[:, 0].to(torch.int64) * (1 << 32) + cells[:, 1].to(torch.int64))
        cells = cells[order]                                             # (stays on the device: the dense case has 17 M cells)
    return count, cells, st


def test_plan_runs_ahead_of_its_read_backs_and_says_when_its_sizes_were_stale(ctx):
    """Option plan_speculate: the second plan of a shape takes its sizes from the first and does not synchronise; the same
    cells.  A third plan of the same shape on data with many more flagged tiles and candidates finds its sizes too small:
    the cell count reads MVS_PLAN_STALE, the plan after it does not speculate and is right."""
    n, d = 8192, 1024
    sparse = synth.make_sketches_numpy(n, d, 3000, seed=3, cluster=2)          # flags its 32 diagonal tiles
    dense = synth.make_sketches_numpy(n, d, 3000, seed=4, cluster=2048)        # 4 clusters of 8 x 8 tiles: 144 on and above the diagonal
    ctx.set_option("pairwise_filter", 2)
    cap = 2500 * n
    a = Split(ctx, sparse, _n2(sparse), 1)
    c1, cells1, s1 = _plan_once(ctx, a, cap, True)
    assert not s1["speculated"] and 0 < c1 < cap
    c2, cells2, s2 = _plan_once(ctx, a, cap, True)
    assert s2["speculated"] and not s2["stale"] and c2 == c1 and torch.equal(cells2, cells1)
    assert (s2["candidates"], s2["flagged_tiles"], s2["filter_tiles"]) == (s1["candidates"], s1["flagged_tiles"], s1["filter_tiles"])
    b = Split(ctx, dense, _n2(dense), 1)
    assert (b.P, b.d_pad, b.n_alloc) == (a.P, a.d_pad, a.n_alloc)
    c3, cells3, s3 = _plan_once(ctx, b, cap, True)
    assert s3["speculated"] and s3["stale"] and c3 >= _capi.PLAN_STALE and s3["flagged_tiles"] > 2 * s1["flagged_tiles"] + 64
    c4, cells4, s4 = _plan_once(ctx, b, cap, True)
    assert not s4["speculated"] and not s4["stale"] and 0 < c4 < cap
    c5, cells5, s5 = _plan_once(ctx, b, cap, False)
    assert c5 == c4 and torch.equal(cells5, cells4)
    c6, cells6, s6 = _plan_once(ctx, b, cap, True)          # and from here on the dense shape speculates on its own counts
    assert s6["speculated"] and not s6["stale"] and c6 == c4 and torch.equal(cells6, cells4)
    a.sset.close()
    b.sset.close()


def test_sharded_comparison_repeats_a_step_whose_plan_was_stale(ctx):
    """parallel.ShardedComparison on one rank: sparse data, then same-shaped dense data -- the second step's plan runs on the
    first step's sizes, reports MVS_PLAN_STALE in its header, the step is repeated and equals mvs_pairwise_rows"""
    n, d = 8192, 1024
    ctx.set_option("pairwise_filter", 2)
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, DEV), 0, 1)
    cells_out = torch.empty((2500 * n, 4), dtype=torch.int32, device=DEV)
    for rep, (seed, cluster) in enumerate(((3, 2), (3, 2), (4, 2048), (4, 2048))):
        sk = synth.make_sketches_numpy(n, d, 3000, seed=seed, cluster=cluster)
        n2 = _n2(sk)
        plain = ctx.sketch_set(sk)
        ref, n_want = ctx.pairwise_rows(plain, n2)
        plain.close()
        want = np.stack([ref[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
        out, cnt, info = sc.run(torch.from_numpy(sk).to(DEV), n2, n, cells_out=cells_out)
        torch.cuda.synchronize()
        assert info.get("plan_respeculated", 0) == (1 if rep == 2 else 0)
        assert cnt == n_want and np.array_equal(out[:cnt].cpu().numpy(), want)


@pytest.mark.parametrize("d", [64, 100, 512, 2048, 2100, 4096, 5000])
@pytest.mark.parametrize("dtype", [np.int32, np.int16])
def test_limb_planes_rebuilt_from_low_limbs_and_the_coarse_plane(ctx, d, dtype):
    """mvs_sketch_set_planes_from_wire: what a rank receives in a multi-rank step -- low limbs, fragment-major coarse plane,
    row statistics -- gives back both limb planes byte for byte, for rows at the largest |v| the rule covers (32 004), rows whose
    radix search settled below the clamping-free radix (bell-shaped entries with a few outliers), one-limb rows, zero rows and
    the padding rows behind the samples; rows outside the range are not touched"""
    rng = np.random.default_rng(1000 + d)
    n, rows = 150, 160                                           # 150 samples in a range of 160 rows starting at row 32
    scale = 10.0 ** rng.uniform(0.0, 3.9, size=(n, 1))           # rows of every magnitude: radices from 1 to 252
    sk = np.rint(rng.normal(0.0, 1.0, size=(n, d)) * scale).astype(np.int64)
    for r in range(0, n, 3):                                     # outliers of 1.5 .. 12 sigma: the radix search clamps some of them
        k = rng.integers(0, d, size=3)
        sk[r, k] = (np.rint(scale[r, 0] * rng.uniform(1.5, 12.0, size=3)) * rng.choice([-1, 1], size=3)).astype(np.int64)
    sk[1] = np.rint(rng.normal(0.0, 220.0, size=d))
    sk[1, rng.integers(0, d)] = 1500
    sk[3] = 0
    sk[5] = rng.integers(-100, 101, size=d)
    sk[7] = rng.integers(-_capi.WIRE_MAX_ABS, _capi.WIRE_MAX_ABS + 1, size=d)
    sk[7, 0] = _capi.WIRE_MAX_ABS
    sk[8] = -sk[7]
    sk[9] = rng.integers(-4000, 4001, size=d)
    sk[9, d // 2] = 31000                                        # one far outlier in a small row
    sk = np.clip(sk, -32004, 32004).astype(dtype)
    n_st = 256
    n_alloc, d_pad, nbytes = ctx.limb_geometry(n_st, d, 2)
    planes = torch.zeros(nbytes, dtype=torch.int8, device=DEV)
    coarse = torch.zeros(n_alloc * d_pad, dtype=torch.uint8, device=DEV)
    stats = torch.zeros(n_alloc * 16, dtype=torch.uint8, device=DEV)
    sset = ctx.sketch_set_from_planes(planes, n_st, n_alloc, d, d_pad, 2)
    ctx.attach_derived(sset, coarse, stats)
    ctx.recode_rows(sset, torch.from_numpy(sk).to(DEV), 32, rows)
    lo = torch.zeros(n_alloc * d_pad, dtype=torch.int8, device=DEV)
    parallel.GpuOps(ctx, DEV).wire_rows(planes, lo, d_pad, 32, rows)
    rebuilt = torch.full((nbytes,), 0x33, dtype=torch.int8, device=DEV)
    other = ctx.sketch_set_from_planes(rebuilt, n_st, n_alloc, d, d_pad, 2)
    ctx.attach_derived(other, coarse, stats)
    ctx.planes_from_wire(other, lo, 32, rows)
    torch.cuda.synchronize()
    a = planes.cpu().numpy().reshape(-1, 2, d_pad)
    b = rebuilt.cpu().numpy().reshape(-1, 2, d_pad)
    assert np.array_equal(b[32:32 + rows], a[32:32 + rows])
    assert np.all(b[:32] == 0x33) and np.all(b[32 + rows:] == 0x33)
    radix = stats.cpu().numpy().reshape(-1, 16)[32:32 + n, :4].copy().view("<i4").reshape(n)
    clampfree = np.maximum(1, (np.abs(sk.astype(np.int64)).max(axis=1) + 126) // 127)
    assert np.all(radix <= clampfree) and np.any(radix < clampfree)          # the search did go below it somewhere
    sset.close()
    other.close()


@pytest.mark.parametrize("world,cluster", [(2, 8), (4, 8), (4, 600), (3, 300)])
def test_plan_rebuilds_the_rows_it_reads_from_their_low_limbs(ctx, world, cluster):
    """mvs_plan_wire: with the other ranks' limb planes poisoned, a plan still finds every cell -- its finish rebuilds the
    columns of its candidates and of its flagged tiles (dense clusters that span ranks) from low limbs + coarse plane, both when it
    waits for its counts and when it runs ahead of them"""
    n, d = 3000, 512
    sk = synth.make_sketches_numpy(n, d, 3000, seed=world + cluster, cluster=cluster)
    ctx.set_option("pairwise_filter", 2)
    split = Split(ctx, sk, _n2(sk), world)
    want, per = _union(split)
    assert len(want) > 4 * n
    for speculate in (0, 1, 1):
        ctx.set_option("plan_speculate", speculate)
        got, per_w = _union(split, wire=True)
        assert np.array_equal(got, want)
    assert any(x[2]["flagged_tiles"] > 0 for x in per_w) or cluster < 100
    split.sset.close()
