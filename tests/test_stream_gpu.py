"""GPU: mvs_pairwise_stream -- the comparison with its result streamed out as CSR pieces of whole rows -- against
mvs_pairwise_rows (the cell list) and the oracle, on every path the library can take for a row range: the two-stage
comparison with its output sized from the candidate count, the exact kernel on one block, the exact kernel on several row
blocks planned against a small device budget, dense results, the int16 keep test, q values beyond 8 bits."""
import numpy as np
import pytest

from metagenome_vector_sketches_amd import _capi, synth
from oracle import pyoracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("restore_options")]


def _n2(sk):
    return np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(row))) for row in sk.astype(np.int32)])


def _triples(row_ptr, col, q, row_begin=0):
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64) + row_begin, np.diff(row_ptr))
    return np.stack([rows, col.astype(np.int64), q.astype(np.int64)], axis=1)


def _cells_triples(cells):
    return np.stack([cells["row"].astype(np.int64), cells["col"].astype(np.int64), cells["q"].astype(np.int64)], axis=1)


@pytest.mark.parametrize("filt", [0, 2])
@pytest.mark.parametrize("keep", [_capi.KEEP_INT32, _capi.KEEP_INT16])
def test_stream_equals_cell_list_and_oracle(ctx, filt, keep):
    sk = synth.make_sketches_numpy(900, 512, 3000, seed=5, cluster=8)
    n2 = _n2(sk)
    ctx.set_option("pairwise_filter", filt)
    ss = ctx.sketch_set(sk)
    cells, cnt = ctx.pairwise_rows(ss, n2, keep_mode=keep)
    if keep == _capi.KEEP_INT32:            # (the oracle applies the floating keep test to int16 sketches only)
        want = orc.pairwise_rows(sk, n2, chunk=192, threads=8)
        want = want[np.lexsort((want["col"], want["row"]))]
        assert np.array_equal(_cells_triples(cells), _cells_triples(want))
    assert cnt > 900 * 6
    row_ptr, col, q, n = ctx.pairwise_stream(ss, n2, keep_mode=keep)
    assert n == cnt and q.dtype == np.uint8 and (ctx.pairwise_candidates() > 0) == (filt == 2)
    assert np.array_equal(_triples(row_ptr, col, q), _cells_triples(cells))
    # a shard of the rows, pieces seen one by one
    seen = []
    n_part = ctx.pairwise_stream(ss, n2, on_block=lambda b, e, rp, c, qq: seen.append((b, e, rp, c, qq)) and None,
                                 row_begin=300, row_end=777, keep_mode=keep)
    assert seen[0][0] == 300 and seen[-1][1] == 777 and all(a[1] == b[0] for a, b in zip(seen, seen[1:]))
    got = np.concatenate([_triples(rp, c, qq, b) for (b, e, rp, c, qq) in seen])
    sel = (cells["row"] >= 300) & (cells["row"] < 777)
    assert n_part == int(sel.sum()) and np.array_equal(got, _cells_triples(cells[sel]))
    ss.close()


def test_stream_toy_db(ctx, gold):
    """the reference's toy DB (61 samples, 35 % of the cells kept): exact kernel, one block"""
    n2 = np.array([orc.norm_sq_from_text(l.split()[1]) for l in gold.norm_lines()])
    ss = ctx.sketch_set(gold.vectors)
    row_ptr, col, q, n = ctx.pairwise_stream(ss, n2)
    want = sorted((r, c, qq) for r, c, _, qq in gold.cells())
    assert n == 1291 and [tuple(int(x) for x in t) for t in _triples(row_ptr, col, q)] == want
    ss.close()


@pytest.mark.parametrize("filt", [0, 1])
@pytest.mark.parametrize("mode", ["dense-per-block", "dense-whole-square", "dense-whole-square-one-stream", "packed-list"])
def test_stream_dense_result_in_row_blocks(ctx, filt, mode):
    """clusters of 1000 samples: a third of all cells are kept.  The filter (when it runs) gives up past 1/128 of the
    cells in its list and the exact kernel does the rows in blocks -- nothing is compared twice and the pieces still
    cover every row once, in order.  The three ways a block's cells leave the exact kernel:
      dense-per-block     one byte per cell, the matrix holds one block (budget 2 MB: blocks of 512 rows)
      dense-whole-square  one matrix for all rows, blocks of 256 rows, mirror images land in later blocks' rows; block k
                          becomes CSR on a side stream while launch k + 1 runs (-one-stream: stream_dense = 2, in series)
      packed-list         64-bit words + radix sort (what other limb codes get), blocks sized for the worst case"""
    n, d = 3000, 256
    sk = synth.make_sketches_numpy(n, d, 3000, seed=77, cluster=1000, shared=0.6)
    n2 = _n2(sk)
    ctx.set_option("pairwise_filter", filt)
    ss = ctx.sketch_set(sk)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    assert cnt > n * n // 4
    budget = 0
    if mode == "dense-per-block":
        budget = 2 << 20
    elif mode.startswith("dense-whole-square"):
        ctx.set_option("stream_block_rows", 256)
        if mode.endswith("one-stream"):
            ctx.set_option("stream_dense", 2)
    else:
        ctx.set_option("stream_dense", 0)
        budget = 16 << 20
    try:
        pieces = []
        n_s = ctx.pairwise_stream(ss, n2, on_block=lambda b, e, rp, c, qq: pieces.append((b, e, rp, c, qq)) and None,
                                  device_budget_bytes=budget)
        st = ctx.stream_stats()
        assert n_s == cnt and pieces[0][0] == 0 and pieces[-1][1] == n
        assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:]))
        assert st["row_blocks"] >= {"dense-per-block": 5, "dense-whole-square": 11, "dense-whole-square-one-stream": 11, "packed-list": 3}[mode] and not st["two_stage"]
        got = np.concatenate([_triples(rp, c, qq, b) for (b, e, rp, c, qq) in pieces])
        assert np.array_equal(got, _cells_triples(cells))
        # a shard that does not start on a tile border
        row_ptr, col, q, n_sh = ctx.pairwise_stream(ss, n2, row_begin=777, row_end=2222, device_budget_bytes=budget)
        sel = (cells["row"] >= 777) & (cells["row"] < 2222)
        assert n_sh == int(sel.sum()) and np.array_equal(_triples(row_ptr, col, q, 777), _cells_triples(cells[sel]))
    finally:
        ctx.set_option("stream_block_rows", 0)
        ctx.set_option("stream_dense", 1)
    # and with the defaults: one block
    row_ptr, col, q, n_one = ctx.pairwise_stream(ss, n2)
    assert n_one == cnt and np.array_equal(_triples(row_ptr, col, q), _cells_triples(cells))
    ss.close()


def _mixed_density(n_dense, n_sparse, d, seed):
    """clusters of n_dense / 2 samples (every pair inside one is kept) followed by clusters of 8"""
    a = synth.make_sketches_numpy(n_dense, d, 3000, seed=seed, cluster=max(2, n_dense // 2), shared=0.6)
    b = synth.make_sketches_numpy(n_sparse, d, 3000, seed=seed + 1, cluster=8)
    return np.concatenate([a, b])


@pytest.mark.parametrize("mode", ["list", "matrix", "matrix-side-stream", "matrix-one-block", "pipeline", "pipeline-side-stream",
                                  "pipeline-big-blocks"])
@pytest.mark.parametrize("keep", [_capi.KEEP_INT32, _capi.KEEP_INT16])
def test_stream_tile_granular_mixed_density(ctx, mode, keep):
    """Dense regions next to sparse ones (the reference's cost is flat in the density, src/pairwise_comp_optimized.cpp:
    135-147; ours must not fall off a cliff between the two): the ping-pong filter flags the 256 x 256 tiles whose waves
    hold more than tile_dense_thr candidates, the exact kernel computes those tiles only, everything else is re-checked
    pair by pair.  The kept cells leave as ONE packed list (few) or through the dense byte matrix (many; stream_list_cells
    lowered to force it): flagged tiles written whole, the re-check's cells scattered as bytes into tiles cleared on first
    touch, the row passes reading only tiles that can hold something.  "matrix": one filter pass over all rows feeds it
    (stream_pipeline = 0), "pipeline": the filter itself runs row block by row block (the default where the first tile row
    looks dense).  Every way: the cells of the exact kernel, bit for bit."""
    n, d = 2900, 256
    sk = _mixed_density(1300, 1600, d, seed=31)
    n2 = _n2(sk)
    ss = ctx.sketch_set(sk)
    ctx.set_option("pairwise_filter", 0)
    cells, cnt = ctx.pairwise_rows(ss, n2, keep_mode=keep)              # the exact kernel on every cell
    if keep == _capi.KEEP_INT32:
        want = orc.pairwise_rows(sk, n2, chunk=192, threads=8)
        want = want[np.lexsort((want["col"], want["row"]))]
        assert np.array_equal(_cells_triples(cells), _cells_triples(want)) and np.array_equal(cells["dot"], want["dot"])
    ctx.set_option("pairwise_filter", 2)
    ctx.set_option("filter_variant", 8)
    cells2, cnt2 = ctx.pairwise_rows(ss, n2, keep_mode=keep)            # cell list through filter + flagged tiles
    cand, flagged, tiles = ctx.pairwise_stats()
    assert cnt2 == cnt and np.array_equal(cells2, cells)
    assert 0 < flagged < tiles and cand > 0                             # both mechanisms had work
    if mode == "list":
        ctx.set_option("stream_list_cells", 1 << 26)
    else:
        ctx.set_option("stream_list_cells", 1000)
        ctx.set_option("stream_pipeline", 1 if mode.startswith("pipeline") else 0)
        ctx.set_option("stream_block_rows", 0 if mode in ("matrix-one-block", "pipeline-big-blocks") else 256)
        if mode.endswith("side-stream"):
            ctx.set_option("stream_dense", 3)
    pieces = []
    n_s = ctx.pairwise_stream(ss, n2, on_block=lambda b, e, rp, c, qq: pieces.append((b, e, rp, c, qq)) and None, keep_mode=keep)
    st = ctx.stream_stats()
    assert st["two_stage"] == (1 if mode == "list" else 3 if mode.startswith("pipeline") else 2)
    assert st["row_blocks"] >= (1 if mode in ("list", "matrix-one-block") else 3 if mode == "pipeline-big-blocks" else 11)
    got = np.concatenate([_triples(rp, c, qq, b) for (b, e, rp, c, qq) in pieces])
    assert n_s == cnt and np.array_equal(got, _cells_triples(cells))
    # rows that start on a 256-row border (the matrix can take them) and rows that do not (list)
    for rb, re in ((512, 2000), (777, 2222)):
        row_ptr, col, q, n_sh = ctx.pairwise_stream(ss, n2, row_begin=rb, row_end=re, keep_mode=keep)
        sel = (cells["row"] >= rb) & (cells["row"] < re)
        assert n_sh == int(sel.sum()) and np.array_equal(_triples(row_ptr, col, q, rb), _cells_triples(cells[sel]))
    # device-encoded rows ride on the same blocks
    from test_encode_gpu import _decode
    enc = ctx.pairwise_stream_encoded(ss, n2, keep_mode=keep)
    assert enc["n_cells"] == cnt and _decode(enc) == [tuple(int(x) for x in t) for t in _cells_triples(cells)]
    ss.close()


@pytest.mark.parametrize("thr", [1, 8, 64, 500, 8192, 0])
def test_tile_density_threshold_changes_no_cell(ctx, thr):
    """tile_dense_thr only moves work between the pair-by-pair re-check and the exact kernel on whole tiles (0: no tile is
    ever flagged, the behaviour up to round 3): same cells at every setting"""
    sk = _mixed_density(700, 900, 512, seed=5)
    n2 = _n2(sk)
    ss = ctx.sketch_set(sk)
    ctx.set_option("pairwise_filter", 0)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    ctx.set_option("pairwise_filter", 2)
    ctx.set_option("filter_variant", 8)
    ctx.set_option("tile_dense_thr", thr)
    cells2, cnt2 = ctx.pairwise_rows(ss, n2)
    cand, flagged, tiles = ctx.pairwise_stats()
    assert cnt2 == cnt and np.array_equal(cells2, cells)
    assert (flagged == 0) == (thr in (0, 8192)) and cand > 0
    row_ptr, col, q, n_s = ctx.pairwise_stream(ss, n2)
    assert n_s == cnt and np.array_equal(_triples(row_ptr, col, q), _cells_triples(cells))
    # a rectangular block with every cell mirrored (the sharded schedule's kind of block)
    import torch
    out = torch.empty((1 << 20, 4), dtype=torch.int32, device="cuda")
    n2_t = torch.from_numpy(n2).to("cuda")
    ctx.set_option("pairwise_filter", 0)
    n_ref = ctx.pairwise_block(ss, n2_t, 0, 512, 512, 1600, _capi.BLOCK_MIRROR_ALL, out, 0)
    ctx.synchronize()
    ref = out[:n_ref].cpu().numpy()
    ctx.set_option("pairwise_filter", 2)
    n_got = ctx.pairwise_block(ss, n2_t, 0, 512, 512, 1600, _capi.BLOCK_MIRROR_ALL, out, 0)
    ctx.synchronize()
    got = out[:n_got].cpu().numpy()
    assert n_got == n_ref and sorted(map(tuple, got.tolist())) == sorted(map(tuple, ref.tolist()))
    ss.close()


def test_stream_q_beyond_8_bits_and_empty_rows(ctx):
    """norms that do not belong to the vectors make the Jaccard estimate negative for the pairs of sample 3 with the
    large samples: the reference's uint16 cast gives q = 65536 + round(255 J) there; such a piece
    comes with 16-bit q values.  Rows 10..19 have huge norms: they keep nothing and appear as empty rows."""
    rng = np.random.default_rng(11)
    sk = rng.integers(-400, 401, size=(40, 256), dtype=np.int32)
    sk[20:] = sk[:20] * 3 // 2                      # related pairs (i, i + 20)
    n2 = _n2(sk)
    n2[3] = 1e-3
    n2[23] *= 1.05
    n2[10:20] = 1e12
    n2[30:40] = 1e12
    ss = ctx.sketch_set(sk)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    want = orc.pairwise_rows(sk, n2, chunk=192, threads=4)
    want = want[np.lexsort((want["col"], want["row"]))]
    assert np.array_equal(_cells_triples(cells), _cells_triples(want))
    assert cnt > 0 and int(cells["q"].max()) > 255
    row_ptr, col, q, n = ctx.pairwise_stream(ss, n2)
    assert n == cnt and q.dtype == np.uint16
    assert np.array_equal(_triples(row_ptr, col, q), _cells_triples(cells))
    assert np.all(np.diff(row_ptr)[10:20] == 0)
    ss.close()


def test_stream_callback_can_stop_and_errors_surface(ctx):
    sk = synth.make_sketches_numpy(600, 256, 3000, seed=9, cluster=8)
    n2 = _n2(sk)
    ss = ctx.sketch_set(sk)
    with pytest.raises(_capi.MvsError) as ei:
        ctx.pairwise_stream(ss, n2, on_block=lambda *a: True)
    assert ei.value.code == _capi.MVS_E_ABORTED

    def boom(*a):
        raise KeyError("from the callback")
    with pytest.raises(KeyError):
        ctx.pairwise_stream(ss, n2, on_block=boom)
    row_ptr, col, q, n = ctx.pairwise_stream(ss, n2)           # the context is fine afterwards
    cells, cnt = ctx.pairwise_rows(ss, n2)
    assert n == cnt and np.array_equal(_triples(row_ptr, col, q), _cells_triples(cells))
    ss.close()


def test_stream_100k_two_stage_matches_cell_list(ctx):
    """BASELINE.json configs[2] size: the two-stage comparison with its output sized between the stages; pieces of 32 MiB"""
    import torch
    n, d = 100_000, 2048
    sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
    ssq = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        ctx.sumsq(sk, out=ssq)
        n2 = np.array([orc.norm_sq_from_text(orc.format_norm(float(x))) for x in np.sqrt(ssq.cpu().numpy() / d)])
        n2_t = torch.from_numpy(n2).to("cuda")
        ss = ctx.sketch_set(sk)
        del sk
        cells_t = torch.empty((n * 24, 4), dtype=torch.int32, device="cuda")
        _, cnt = ctx.pairwise_rows(ss, n2_t, cells_out=cells_t)
        torch.cuda.synchronize()
        cells = cells_t[:cnt].cpu().numpy()
        row_ptr, col, q, n_s = ctx.pairwise_stream(ss, n2_t)
        assert ctx.pairwise_candidates() > 0
        ss.close()
    finally:
        ctx.set_stream(None)
    assert n_s == cnt and cnt >= 16 * n
    got = _triples(row_ptr, col, q)
    assert np.array_equal(got, cells[:, [0, 1, 3]].astype(np.int64))


@pytest.mark.parametrize("seed", range(8))
def test_stream_random_shapes(ctx, seed):
    """random sizes, dimensions that are not multiples of the tile edges, cluster sizes from sparse to dense, a random row
    range, a random budget (so that dense blocks, whole-square symmetry and the packed list all come up): the streamed
    pieces always equal the cell list"""
    rng = np.random.default_rng(1000 + seed)
    n = int(rng.integers(40, 900))
    d = int(rng.choice([64, 100, 192, 256, 520]))
    cluster = int(rng.choice([4, 16, max(4, n // 5), max(4, n // 2)]))
    sk = synth.make_sketches_numpy(n, d, int(rng.choice([300, 3000, 40000])), seed=seed, cluster=cluster,
                                   shared=float(rng.choice([0.3, 0.6])))
    n2 = _n2(sk)
    ctx.set_option("pairwise_filter", int(rng.choice([0, 1, 2])))
    ctx.set_option("stream_dense", int(rng.choice([0, 1, 1])))
    ctx.set_option("stream_block_rows", int(rng.choice([0, 256, 512])))
    try:
        ss = ctx.sketch_set(sk)
        cells, cnt = ctx.pairwise_rows(ss, n2)
        rb = int(rng.integers(0, n // 2))
        re = int(rng.integers(rb, n + 1))
        budget = int(rng.choice([0, 1 << 18, 1 << 20, 1 << 22]))
        row_ptr, col, q, n_s = ctx.pairwise_stream(ss, n2, row_begin=rb, row_end=re, device_budget_bytes=budget)
        sel = (cells["row"] >= rb) & (cells["row"] < re)
        assert n_s == int(sel.sum())
        assert np.array_equal(_triples(row_ptr, col, q, rb), _cells_triples(cells[sel]))
        ss.close()
    finally:
        ctx.set_option("stream_dense", 1)
        ctx.set_option("stream_block_rows", 0)


@pytest.mark.parametrize("seed", range(10))
def test_stream_fuzz_seeded(ctx, seed):
    """a fixed handful of tests/fuzz_stream.py's random cases (sketch set x request x library switches): cell list vs the
    oracle, CSR pieces and decoded device-encoded rows vs the cell list"""
    import fuzz_stream
    rng = np.random.default_rng(424242 + seed)
    info = fuzz_stream.run_case(ctx, rng, max_n=900)
    assert info["cells"] >= 0


@pytest.mark.parametrize("clusters", [1, 3])
def test_stream_large_block_dense_everywhere_or_a_third(ctx, clusters):
    """a block large enough for the ping-pong filter by size (24 576 samples: > 4096 tiles), default options.  One cluster:
    every cell is kept, every tile would be flagged -- the filter stops once 70 % of its tiles are, and the exact kernel
    does the shard (path 0; with the segmented filter the first tile row alone says so).  Three clusters: a third of the
    tiles is flagged, below the stop mark -- the tile-granular comparison feeds the dense byte matrix (path 3: filter per
    segment of row blocks; path 2 with stream_pipeline = 0: one filter pass).  Either way the stream equals the exact kernel's (pairwise_filter = 0):
    cell count, per-piece order, and checksums over (row, col, q) of all 3-6 * 10^8 cells."""
    import ctypes
    import torch
    n, d = 24_576, 256
    sk = synth.make_sketches_torch(n, d, 50_000, seed=7, device="cuda", cluster=n // clusters, shared=0.8)
    ssq = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        ctx.sumsq(sk, out=ssq)
        n2 = np.array([orc.norm_sq_from_text(orc.format_norm(float(x))) for x in np.sqrt(ssq.cpu().numpy() / d)])
        n2_t = torch.from_numpy(n2).to("cuda")
        ss = ctx.sketch_set(sk)
        del sk

        def run():
            seen = {"cells": 0, "rows": 0, "sum_col": 0, "sum_rcq": 0, "diag": 0, "next_row": 0, "ordered": True}

            def count(_user, bp):
                b = bp.contents
                seen["ordered"] = seen["ordered"] and b.row_begin == seen["next_row"]
                seen["next_row"] = b.row_end
                seen["cells"] += b.n_cells
                seen["rows"] += b.row_end - b.row_begin
                rp = np.ctypeslib.as_array(b.row_ptr, shape=(b.row_end - b.row_begin + 1,))
                col = np.ctypeslib.as_array(b.col, shape=(b.n_cells,)).astype(np.int64)
                q = np.ctypeslib.as_array(b.q, shape=(b.n_cells,)).astype(np.int64)
                rows = np.repeat(np.arange(b.row_begin, b.row_end, dtype=np.int64), np.diff(rp))
                seen["sum_col"] += int(col.sum())
                seen["sum_rcq"] += int(((rows * 31 + col * 7 + 1) * q % 1000003).sum())
                seen["diag"] += int((q[rows == col] == 255).sum())
                return 0
            cb = _capi.ROW_BLOCK_CB(count)
            cnt = ctypes.c_int64()
            rc = ctx.lib.mvs_pairwise_stream(ctx._h, ss._h, n2_t.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, cb, None,
                                             ctypes.byref(cnt))
            assert rc == 0, ctx.lib.mvs_last_error()
            assert cnt.value == seen["cells"] and seen["rows"] == n and seen["ordered"]
            return seen, ctx.stream_stats()["two_stage"]
        got, path = run()
        ctx.set_option("stream_pipeline", 0)
        got2, path2 = run()
        ctx.set_option("pairwise_filter", 0)
        want, path0 = run()
        ss.close()
    finally:
        ctx.set_stream(None)
    m = n // clusters
    assert path0 == 0 and path == (0 if clusters == 1 else 3) and path2 == (0 if clusters == 1 else 2)
    assert got == want and got2 == want and got["cells"] >= clusters * m * m and got["diag"] == n
