"""GPU: the pairwise kernels (through the C ABI) against the oracle and the reference fixtures.
Bit-exact: dot products are int32 mod 2^32, the keep test and the 8-bit Jaccard are evaluated in the
reference's own operation order in fp64."""
import os

import numpy as np
import pytest

from metagenome_vector_sketches_amd import _capi, synth
from oracle import pyoracle as orc

pytestmark = [pytest.mark.gpu, pytest.mark.usefixtures("restore_options")]


def _n2_from_sketches(sk):
    """norms as the DB would carry them: '%g' text of sqrt(sumsq/d), squared (pairwise_comp_optimized.cpp:893-901)"""
    return np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(row))) for row in sk.astype(np.int32)])


def _cells_tuple(cells):
    return [(int(c["row"]), int(c["col"]), int(c["dot"]), int(c["q"])) for c in cells]


def _oracle_sorted(sk, n2, **kw):
    return sorted(_cells_tuple(orc.pairwise_rows(sk, n2, threads=8, **kw)))


K3 = 0x103   # MVS_LIMBS_K3: three planes of base-128 digits, 3 matrix-core passes per cell


@pytest.fixture(params=["exact", "exact_ring", "two_stage", "two_stage_pp"])
def pw_filter(request, ctx):
    """run a comparison test on every comparison path: the exact kernel on every cell (option pairwise_filter = 0;
    ping-pong kernel = default, and the ring kernel it replaced), and the coarse filter + exact re-check of the
    candidates forced on even for small blocks (= 2; 128 x 128 ring tiles = default for small blocks, and the
    256 x 256 ping-pong kernel that large blocks get)"""
    ctx.set_option("pairwise_filter", 0 if request.param.startswith("exact") else 2)
    if request.param == "exact_ring":
        ctx.set_option("pairwise_variant", 6)
    if request.param == "two_stage_pp":
        ctx.set_option("filter_variant", 8)
    return request.param


@pytest.mark.parametrize("n,d,hi,code", [(61, 2048, 1500, 2), (61, 2048, 1500, K3), (300, 2048, 8127, K3),
                                         (300, 2048, 8128, 2), (257, 100, 1500, K3), (257, 100, 32639, 2),
                                         (130, 4096, 1500, 2), (130, 4096, 1500, K3), (5, 64, 20000, 2),
                                         (129, 2048 + 64, 8000, K3), (129, 2048 + 64, 8000, 2)])
def test_dots_mfma_and_valu_vs_oracle(ctx, n, d, hi, code):
    """both exact limb schemes: two base-256 limbs (default) and the opt-in 3-plane Karatsuba scheme"""
    rng = np.random.default_rng(n * 7 + d)
    sk = rng.integers(-hi, hi + 1, size=(n, d), dtype=np.int32)   # asymmetric
    sk[0] = 0
    sk[1, 0], sk[2, 1] = hi, -hi
    ss = ctx.sketch_set(sk, limbs=code)
    assert ss.limbs == code
    auto = ctx.sketch_set(sk)
    assert auto.limbs == 2          # library default for 128 <= max|v| <= 32639
    auto.close()
    want = orc.dots_dense(sk, 0, n, 0, n, threads=8)
    assert np.array_equal(ctx.pairwise_dots(ss, 0, n, 0, n, algo=0), want)
    assert np.array_equal(ctx.pairwise_dots(ss, 0, n, 0, n, algo=1), want)
    # off-diagonal rectangular block (row/col tiles differ; catches a transposed C/D mapping)
    r0, r1, c0, c1 = n // 3, n, 0, n // 2 + 1
    assert np.array_equal(ctx.pairwise_dots(ss, r0, r1, c0, c1, algo=0), want[r0:r1, c0:c1])
    ss.close()


def test_dots_single_limb(ctx):
    rng = np.random.default_rng(3)
    sk = rng.integers(-128, 128, size=(200, 2048), dtype=np.int32)
    sk[sk == -128] = -127
    ss = ctx.sketch_set(sk)
    assert ss.limbs == 1
    assert np.array_equal(ctx.pairwise_dots(ss, 0, 200, 0, 200), orc.dots_dense(sk, 0, 200, 0, 200, threads=8))
    ss.close()


@pytest.mark.parametrize("hi,limbs", [(8127, K3), (32639, 2), (40000, 3), (8355711, 3), (9_000_000, 4), (2**31 - 1, 4)])
def test_dots_wrap_and_many_limbs(ctx, hi, limbs):
    """large entries: products overflow int32 and must wrap exactly like the reference's MatrixXi product"""
    rng = np.random.default_rng(hi % 1000)
    sk = rng.integers(-hi, hi, size=(70, 256), dtype=np.int64).astype(np.int32)
    sk[0, 0], sk[1, 1] = hi, -hi
    if hi == 2**31 - 1:
        sk[2, 2] = -2**31
    ss = ctx.sketch_set(sk, limbs=limbs if limbs == K3 else None)
    assert ss.limbs == limbs
    want = orc.dots_dense(sk, 0, 70, 0, 70, threads=8)
    assert np.array_equal(ctx.pairwise_dots(ss, 0, 70, 0, 70, algo=0), want)
    assert np.array_equal(ctx.pairwise_dots(ss, 0, 70, 0, 70, algo=1), want)
    ss.close()


def test_toy_cells_reference_db(ctx, gold, pw_filter):
    """config 1: the reference-built toy DB (vectors.bin + vector_norms.txt) -> 1291 kept cells, the rows
    SURVEY.md recorded, every (row, col, dot, q) equal to the fixture"""
    n2 = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in gold.norm_lines()])
    ss = ctx.sketch_set(gold.vectors)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    got = _cells_tuple(cells)
    assert cnt == 1291 == gold.kat["survey_kept_cells"]["int32"]
    ssk = ctx.sketch_set(gold.vectors, limbs=K3)          # toy max |v| = 1263: Karatsuba planes are exact too
    cells_k, cnt_k = ctx.pairwise_rows(ssk, n2)
    assert cnt_k == cnt and _cells_tuple(cells_k) == got
    ssk.close()
    assert got == sorted(gold.cells())            # library order is (row, col)
    by_row = {}
    for r, c, dot, q in got:
        by_row.setdefault(gold.names[r], {})[gold.names[c]] = q
    for rname, pin in gold.kat["survey_pairwise_pins"].items():
        for cname, q in zip(pin["cols"], pin["q"]):
            assert by_row[rname][cname] == q
    # int16 DB + floating keep test -> 1293 cells
    ss16 = ctx.sketch_set(gold.vectors.astype(np.int16))
    cells16, cnt16 = ctx.pairwise_rows(ss16, n2, keep_mode=_capi.KEEP_INT16)
    assert cnt16 == 1293 and _cells_tuple(cells16) == sorted(gold.cells(int16=True))
    ss.close()
    ss16.close()


def test_kept_cells_equal_the_references_own_functions(ctx, gold, pw_filter):
    """a5/a6/a10/a11 against REFERENCE output: every run recorded from oracle/_ref/ref_pairwise32 / ref_pairwise16 (the
    reference's own loaders, product and keep test of both dtypes, compiled from line ranges of its sources:
    tests/golden/make_golden_pairwise.py) -- toy DB, cells on the keep threshold, products that wrap mod 2^32, NaN /
    infinite / negative norms, d = 100, shards -- reproduced by the HIP path: the same (row, col, dot), bit for bit, on
    every comparison path.  The library's order is (row, col); the reference's is its tile loop's (:949-982)."""
    for name, c in gold.ref_pairwise_cases().items():
        ss = ctx.sketch_set(c["vectors"])
        keep = _capi.KEEP_INT32 if c["elem"] == 4 else _capi.KEEP_INT16
        for run in c["runs"]:
            b, e = orc.shard_rows(len(c["vectors"]), run["num_shards"], run["shard_idx"])
            cells, cnt = ctx.pairwise_rows(ss, c["norms_sq"], row_begin=b, row_end=e, keep_mode=keep)
            got = [(int(x["row"]), int(x["col"]), int(x["dot"])) for x in cells]
            assert cnt == run["kept"], (name, run["num_shards"], run["shard_idx"])
            assert got == sorted(tuple(x) for x in run["cells"].tolist()), (name, run["num_shards"], run["shard_idx"])
        # the streamed form the executable uses carries (row, col, q): the same kept SET
        row_ptr, col, q, n_stream = ctx.pairwise_stream(ss, c["norms_sq"], keep_mode=keep)
        whole = [r for r in c["runs"] if r["num_shards"] == 1][0]
        rows = np.repeat(np.arange(len(row_ptr) - 1), np.diff(row_ptr))
        assert sorted(zip(rows.tolist(), col.tolist())) == sorted((x[0], x[1]) for x in whole["cells"].tolist()), name
        # a9 arithmetic against REFERENCE output: q of every cell == the reference's own quantiser lines (:654-672, the
        # binaries' rows mode), from the cell list's epilogue and from the streamed form; 16-bit values included
        head = c["writer_head"]
        if c["elem"] == 4:
            ref_q = {(r, cc): v for r, cc, v in head["cells"].tolist() if (r, cc) not in head["undefined"]}
            cells, _ = ctx.pairwise_rows(ss, c["norms_sq"], keep_mode=keep)
            assert {(int(x["row"]), int(x["col"])): int(x["q"]) for x in cells
                    if (int(x["row"]), int(x["col"])) not in head["undefined"]} == ref_q, name
            assert {k: v for k, v in zip(zip(rows.tolist(), col.tolist()), q.tolist())
                    if k not in head["undefined"]} == ref_q, name
        ss.close()


def test_clustered_synthetic_vs_oracle(ctx, pw_filter):
    """sketches of real (synthetic) hash sets: clusters of 16 with Jaccard ~0.25 -> ~16 kept cells per row"""
    hashes, offsets = synth.make_csr_numpy(400, 3000, seed=5, cluster=16, shared=0.4, lognormal_sigma=0.8)
    sk = ctx.project_csr(hashes, offsets, 2048)
    n2 = _n2_from_sketches(sk)
    ss = ctx.sketch_set(sk)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    want = _oracle_sorted(sk, n2, chunk=192)
    assert cnt == len(want) and cnt > 400 * 8
    assert _cells_tuple(cells) == want
    # shards = row ranges; union of shards == whole (src/pairwise_comp_optimized.cpp:938-940)
    parts = []
    for k in range(3):
        b, e = _capi.shard_rows(400, 3, k)
        c, _ = ctx.pairwise_rows(ss, n2, row_begin=b, row_end=e)
        assert all(b <= x["row"] < e for x in c)
        parts += _cells_tuple(c)
    assert parts == want
    ss.close()


def test_keep_threshold_edges(ctx, pw_filter):
    """cells sitting exactly on the keep threshold: truncating vs floating division"""
    d = 64
    sk = np.zeros((4, d), dtype=np.int32)
    sk[0, :] = 1                      # dot(0,0) = 64 -> 64/64 = 1
    sk[1, :63] = 1                    # dot(0,1) = 63 -> trunc 0 / float 0.98
    sk[2, :] = -1                     # negative dots: trunc toward zero
    sk[3, 0] = 1
    n2 = np.array([1.0, 0.5, 1.0, 0.0])
    ss = ctx.sketch_set(sk)
    for mode in (_capi.KEEP_INT32, _capi.KEEP_INT16):
        cells, _ = ctx.pairwise_rows(ss, n2, keep_mode=mode)
        skx = sk if mode == _capi.KEEP_INT32 else sk.astype(np.int16)
        want = sorted(_cells_tuple(orc.pairwise_rows(skx, n2, chunk=192)))
        assert _cells_tuple(cells) == want
    ss.close()


def test_capacity_error_reports_needed(ctx, gold):
    n2 = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in gold.norm_lines()])
    ss = ctx.sketch_set(gold.vectors)
    with pytest.raises(_capi.MvsError) as ei:
        ctx.pairwise_rows(ss, n2, capacity=100)
    assert ei.value.code == _capi.MVS_E_CAPACITY and "1291" in str(ei.value)
    ss.close()


def test_full_width_properties(ctx, pw_filter):
    """N = 4096, d = 2048 (sketch magnitudes of 50k-hash samples): no oracle for the whole matrix --
    symmetry of the kept set, diagonal = 255, checksum of dots vs the VALU path on a stripe, and an
    oracle check of 64 rows."""
    import torch
    n, d = 4096, 2048
    sk_t = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
    sk = sk_t.cpu().numpy()
    n2 = _n2_from_sketches(sk)
    ss = ctx.sketch_set(sk_t)
    assert ss.limbs in (2, K3)
    cells, cnt = ctx.pairwise_rows(ss, n2)
    got = _cells_tuple(cells)
    s = set((r, c) for r, c, _, _ in got)
    assert all((c, r) in s for r, c in s)                         # symmetric
    dq = {(r, c): (dot, q) for r, c, dot, q in got}
    assert all(dq[(i, i)][1] == 255 for i in range(n))            # self pairs kept with J = 1
    assert all(dq[(r, c)] == dq[(c, r)] for r, c in s)
    assert cnt >= n * 12                                          # ~16 cluster mates per row
    rows = slice(1000, 1064)
    want = _oracle_sorted(sk, n2, row_begin=rows.start, row_end=rows.stop, chunk=192)
    assert [t for t in got if rows.start <= t[0] < rows.stop] == want
    a = ctx.pairwise_dots(ss, 2048, 2048 + 256, 0, n, algo=0)
    b = ctx.pairwise_dots(ss, 2048, 2048 + 256, 0, n, algo=1)
    assert np.array_equal(a, b)
    ss.close()


@pytest.mark.parametrize("n,d,clu", [(2321, 2048, 16), (2600, 100, 600), (2305, 4096, 2305)])
def test_fragment_major_planes_change_no_cell(ctx, n, d, clu):
    """The ping-pong exact kernel and the ping-pong tile filter copy their LDS pieces from fragment-major copies of the limb
    planes / the coarse plane (one contiguous KiB per copy instruction; option fragment_major, on by default) when the
    block's origin is a multiple of 16 samples, otherwise from the row-major planes, and load the B operand's fragments
    straight from those copies into registers (option pairwise_bdirect, on by default): every combination gives the cells of
    the row-major path -- exact kernel on every cell, two-stage with the ping-pong filter (flagged tiles go to the exact
    kernel on a tile list), sparse and dense results, blocks that start off the 16-sample grid, shards -- and an oracle
    stripe pins them."""
    sk = synth.make_sketches_numpy(n, d, 3000, seed=n + d, cluster=clu)
    n2 = _n2_from_sketches(sk)
    ss = ctx.sketch_set(sk)
    assert ss.limbs == 2
    cap = 1 << 23
    results = {}
    # (fragment_major, pairwise_bdirect): B operand straight from the fragment-major plane into registers / both operands
    # through LDS copied from the fragment-major planes / the row-major planes
    for fmaj, bdir in ((1, 1), (1, 0), (0, 0)):
        ctx.set_option("fragment_major", fmaj)
        ctx.set_option("pairwise_bdirect", bdir)
        for filt, variant in ((0, -1), (2, 8)):
            ctx.set_option("pairwise_filter", filt)
            ctx.set_option("filter_variant", variant)
            cells, cnt = ctx.pairwise_rows(ss, n2, capacity=cap)
            whole = _cells_tuple(cells)
            parts = []
            # (16, 2016): origin on the 16-sample grid but off the tile grid, large enough for the fragment-major copy;
            # (2016, 2023) / (2023, n): small blocks, the last one starting off the 16-sample grid (row-major planes)
            for b, e in ((0, 16), (16, 2016), (2016, 2023), (2023, n)):
                c, _ = ctx.pairwise_rows(ss, n2, row_begin=b, row_end=e, capacity=cap)
                parts += _cells_tuple(c)
            assert parts == whole
            results[(fmaj, bdir, filt)] = whole
    ref = results[(0, 0, 0)]
    assert all(v == ref for v in results.values()) and len(ref) >= n
    want = _oracle_sorted(sk, n2, row_begin=1030, row_end=1060, chunk=192)
    assert [t for t in ref if 1030 <= t[0] < 1060] == want
    ss.close()


def test_sharded_comparison_single_rank(ctx, pw_filter):
    """metagenome_vector_sketches_amd.parallel with the real GPU back end, world = 1 (the multi-rank
    orchestration is covered on CPU with gloo in tests/test_distributed_cpu.py)"""
    import torch
    from metagenome_vector_sketches_amd import parallel
    sk = synth.make_sketches_numpy(300, 2048, 3000, seed=4, cluster=8)
    n2 = _n2_from_sketches(sk)
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, "cuda:0"), 0, 1, None)   # binds ctx to torch's stream
    try:
        cells_dev = torch.empty((1 << 16, 4), dtype=torch.int32, device="cuda:0")
        _, cnt, info = sc.run(torch.from_numpy(sk).to("cuda:0"), n2, 300, cells_out=cells_dev)
        ctx.synchronize()
        got = [tuple(int(x) for x in row) for row in cells_dev[:cnt].cpu().numpy()]
    finally:
        ctx.set_stream(None)
    assert got == _oracle_sorted(sk, n2, chunk=192) and info["limbs"] in (2, K3)


def test_baseline_config2_full_size(ctx):
    """BASELINE.json configs[1] at full size (10 000 samples x 50 000 hashes, d = 2048) through the same
    calls bench.py makes: too big for the oracle as a whole, so size-independent properties + oracle spot
    checks.  Projection: parity/bounds of every entry, cluster structure visible in the norms; pairwise:
    symmetric kept set, diagonal q = 255, ~16 mates per row, 48 rows and 3 sketches against the oracle."""
    import torch
    from metagenome_vector_sketches_amd import parallel
    S, NH, D = 10_000, 50_000, 2048
    hashes, offsets = synth.make_csr_torch(S, NH, seed=1234, device="cuda", cluster=16, shared=0.4)
    sk_t = torch.empty((S, D), dtype=torch.int32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        ctx.project_csr(hashes, offsets, D, out=sk_t)
        sumsq_t = torch.empty(S, dtype=torch.int64, device="cuda")
        _, max_abs = ctx.stats(sk_t, out=sumsq_t)
        sk = sk_t.cpu().numpy()
        assert np.all((sk - NH) % 2 == 0) and max_abs == int(np.abs(sk).max()) < 8 * NH ** 0.5
        hh = hashes.cpu().numpy().view(np.uint64)
        for s in (0, 4999, 9999):
            assert np.array_equal(sk[s], orc.project(hh[offsets[s]:offsets[s + 1]], D))
        sumsq = sumsq_t.cpu().numpy()
        assert np.array_equal(sumsq[:64], (sk[:64].astype(np.int64) ** 2).sum(1))
        n2 = _n2_from_sketches(sk)
        assert np.allclose(n2, NH, rtol=0.2)                        # E[sum v^2 / d] = n
        sc = parallel.ShardedComparison(parallel.GpuOps(ctx, "cuda:0"), 0, 1, None)
        cells_dev = torch.empty((1 << 20, 4), dtype=torch.int32, device="cuda")
        _, cnt, info = sc.run(sk_t, n2, S, cells_out=cells_dev, max_abs_local=max_abs)
        ctx.synchronize()
        cells = cells_dev[:cnt].cpu().numpy()
    finally:
        ctx.set_stream(None)
    rows, cols, q = cells[:, 0], cells[:, 1], cells[:, 3]
    key = rows.astype(np.int64) * S + cols
    assert np.all(np.diff(key) > 0)                                  # sorted by (row, col), no duplicates
    assert np.array_equal(np.sort(cols.astype(np.int64) * S + rows), key)   # symmetric kept set
    diag = cells[rows == cols]
    assert len(diag) == S and np.all(diag[:, 3] == 255)
    per_row = np.bincount(rows, minlength=S)
    assert per_row.min() >= 16 and per_row.mean() < 17.5             # 16 cluster members (incl. self) + rare chance hits
    same_cluster = (rows // 16) == (cols // 16)
    assert same_cluster.sum() == S * 16
    want = _oracle_sorted(sk, n2, row_begin=5000, row_end=5048, chunk=192)
    got = [tuple(int(x) for x in c) for c in cells[(rows >= 5000) & (rows < 5048)]]
    assert got == want


def test_half_million_samples_grid_limits(ctx):
    """N = 500 000 (2.5e11 cells, 15.6 M workgroups): beyond what a one-dimensional dispatch can address
    (2^32 work-items); exercises the 2-D tile grid.  Properties only: every row keeps its 16 cluster mates
    (incl. itself, q = 255 on the diagonal), the kept set is symmetric, sorted, within bounds."""
    import torch
    n, d = 500_000, 2048
    sk_t = synth.make_sketches_torch(n, d, 50_000, seed=31, device="cuda")
    ss_t = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        _, max_abs = ctx.stats(sk_t, out=ss_t)
        n2 = torch.sqrt(ss_t.double() / d) ** 2
        sset = ctx.sketch_set(sk_t)
        cells_t = torch.empty((n * 20, 4), dtype=torch.int32, device="cuda")
        _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells_t)
        torch.cuda.synchronize()
        cells = cells_t[:cnt].cpu().numpy()
        sset.close()
    finally:
        ctx.set_stream(None)
    rows, cols = cells[:, 0].astype(np.int64), cells[:, 1].astype(np.int64)
    assert rows.min() == 0 and rows.max() == n - 1 and cols.min() == 0 and cols.max() == n - 1
    key = rows * n + cols
    assert np.all(np.diff(key) > 0)
    assert np.array_equal(np.sort(cols * n + rows), key)
    assert int(((rows // 16) == (cols // 16)).sum()) == n * 16
    per_row = np.bincount(rows, minlength=n)
    assert per_row.min() >= 16 and np.all(cells[rows == cols][:, 3] == 255) and int((rows == cols).sum()) == n


def _random_rows(rng, n, d, kind):
    """sketch-like rows of very different shapes, all within two base-256 limbs"""
    if kind == "mixed":       # set sizes over three decades, clusters of related rows
        sizes = (10 ** rng.uniform(1.5, 5.0, n)).astype(np.int64)
        base = rng.standard_normal((n // 8 + 1, d))
        sk = np.empty((n, d), dtype=np.int64)
        for i in range(n):
            k = int(0.5 * sizes[i])
            x = base[i // 8] * np.sqrt(k) + rng.standard_normal(d) * np.sqrt(sizes[i] - k)
            sk[i] = np.round((x - (sizes[i] & 1)) / 2) * 2 + (sizes[i] & 1)
        return np.clip(sk, -32639, 32639).astype(np.int32)
    if kind == "peaky":       # a few huge coordinates on small noise: the coarse plane loses almost everything
        sk = rng.integers(-3, 4, (n, d))
        for i in range(n):
            idx = rng.integers(0, d, 3)
            sk[i, idx] = rng.integers(-18000, 18000, 3)   # sums of squares stay below 2^31
        sk[::7] = sk[3]       # exact duplicates
        sk[5] = -sk[3]        # and an anti-correlated row
        return sk.astype(np.int32)
    if kind == "flat":        # constant and near-constant rows, empty rows
        sk = np.tile(rng.integers(-1500, 1500, (n, 1)), (1, d)) + rng.integers(-1, 2, (n, d))
        sk[::5] = 0
        return sk.astype(np.int32)
    raise ValueError(kind)


@pytest.mark.parametrize("kind,n,d", [("mixed", 700, 2048), ("mixed", 500, 4096), ("mixed", 900, 100),
                                      ("peaky", 600, 2048), ("flat", 400, 512), ("mixed", 1500, 1024),
                                      ("mixed", 300, 8192), ("mixed", 260, 32768)])
@pytest.mark.parametrize("mode", ["int32", "int16"])
@pytest.mark.parametrize("fv", [-1, 8])
def test_two_stage_equals_exact_and_oracle(ctx, monkeypatch, kind, n, d, mode, fv):
    """the coarse filter may only drop pairs the keep test rejects: same cells as the exact kernel and the
    oracle on adversarial row shapes, both keep tests"""
    rng = np.random.default_rng(hash((kind, n, d)) % 2 ** 32)
    sk = _random_rows(rng, n, d, kind)
    n2 = _n2_from_sketches(sk)
    n2[::11] *= 0.3          # norms that do not match the rows (the filter must not rely on them)
    keep = _capi.KEEP_INT32 if mode == "int32" else _capi.KEEP_INT16
    ss = ctx.sketch_set(sk)
    assert ss.limbs == 2
    ctx.set_option("pairwise_filter", 2)
    ctx.set_option("filter_variant", fv)                             # -1: 128 x 128 ring tiles here; 8: ping-pong kernel
    two, cnt_two = ctx.pairwise_rows(ss, n2, keep_mode=keep)
    n_cand, n_flagged, _ = ctx.pairwise_stats()
    assert fv == 8 or n_flagged == 0                                 # only the ping-pong filter flags dense tiles
    ctx.set_option("pairwise_filter", 0)
    exact, cnt_exact = ctx.pairwise_rows(ss, n2, keep_mode=keep)
    assert ctx.pairwise_candidates() == 0
    assert cnt_two == cnt_exact and np.array_equal(two, exact)
    # every kept pair was a candidate (upper triangle) or lies in a tile the filter handed to the exact kernel whole
    assert n_cand + n_flagged * 65536 >= cnt_two // 2
    skx = sk if mode == "int32" else sk.astype(np.int16)
    assert _cells_tuple(two) == _oracle_sorted(skx, n2, chunk=192)
    ss.close()


def test_two_stage_threshold_knife_edge(ctx, monkeypatch):
    """pairs whose dot sits exactly on, one below and one above d * 0.05 * (n2_i + n2_j)"""
    d = 2048
    rng = np.random.default_rng(99)
    base = rng.integers(-900, 900, d)
    sk = np.stack([base + rng.integers(-40, 40, d) for _ in range(64)]).astype(np.int32)
    P = sk.astype(np.int64) @ sk.astype(np.int64).T
    n2 = np.empty(64)
    # choose n2 so that 0.05 * (n2_0 + n2_j) lands on trunc(P_0j / d) + {-1, 0, +1} / 3
    n2[0] = 1000.0
    for j in range(1, 64):
        q = P[0, j] // d
        n2[j] = (q + ((j % 3) - 1) / 3.0) / 0.05 - n2[0]
    ss = ctx.sketch_set(sk)
    for keep in (_capi.KEEP_INT32, _capi.KEEP_INT16):
        ctx.set_option("pairwise_filter", 2)
        two, _ = ctx.pairwise_rows(ss, n2, keep_mode=keep)
        assert ctx.pairwise_candidates() > 0
        skx = sk if keep == _capi.KEEP_INT32 else sk.astype(np.int16)
        assert _cells_tuple(two) == _oracle_sorted(skx, n2, chunk=192)
    ss.close()


def test_two_stage_wrap_guard(ctx, monkeypatch):
    """rows whose sum of squares reaches 2^31: their int32 dots may wrap, the bound says nothing about them --
    every pair of such a row goes to the exact re-check, which reproduces the wrapped value"""
    d = 2048
    rng = np.random.default_rng(5)
    sk = rng.integers(-700, 700, (300, d)).astype(np.int32)
    for i in (17, 18, 40):
        sk[i] = rng.integers(0, 2, d) * 2200 - 1100      # 2048 * 1100^2 = 2.48e9 >= 2^31
    sk[18] = sk[17]                                       # dot = 2.48e9 -> wraps negative: not kept
    sk[41] = -sk[40]                                      # dot = -2.48e9 -> wraps positive: kept by the reference
    n2 = _n2_from_sketches(sk)
    P = (sk[17].astype(np.int64) * sk[18]).sum()
    assert P >= 2 ** 31 and (sk[40].astype(np.int64) * sk[41]).sum() <= -2 ** 31
    ss = ctx.sketch_set(sk)
    for f in ("2", "0"):
        ctx.set_option("pairwise_filter", int(f))
        cells, _ = ctx.pairwise_rows(ss, n2)
        assert (ctx.pairwise_candidates() >= 4 * 300 - 16) == (f == "2")
        got = _cells_tuple(cells)
        assert got == _oracle_sorted(sk, n2, chunk=192)
    assert (17, 18) not in {(r, c) for r, c, _, _ in got}
    ss.close()


def test_two_stage_gives_way_when_everything_is_a_candidate(ctx, monkeypatch):
    """all rows equal: every pair is kept; when the candidate list overflows and holds more than 1/128 of the
    block the exact kernel takes over (and stays in charge for this set) -- same cells either way"""
    import torch
    n, d = 3000, 256
    row = np.random.default_rng(3).integers(-900, 900, d).astype(np.int32)
    sk = np.tile(row, (n, 1))
    n2 = _n2_from_sketches(sk[:1]).repeat(n)
    ctx.set_option("pairwise_filter", 2)
    ss = ctx.sketch_set(sk)
    with pytest.raises(_capi.MvsError) as ei:
        ctx.pairwise_rows(ss, n2, capacity=1 << 20)
    assert ei.value.code == _capi.MVS_E_CAPACITY and str(n * n) in str(ei.value)
    cells = torch.empty((n * n, 4), dtype=torch.int32, device="cuda")
    _, cnt = ctx.pairwise_rows(ss, torch.from_numpy(n2).to("cuda"), cells_out=cells)
    ctx.synchronize()
    # exact kernel (candidates() == 0) if the candidate list overflowed above; if an earlier, larger comparison
    # left the context with a list big enough, the filter ran and passed every pair of the upper triangle
    assert cnt == n * n and ctx.pairwise_candidates() in (0, n * (n + 1) // 2)
    got = cells.cpu().numpy()
    assert np.array_equal(got[:, 0], np.repeat(np.arange(n), n)) and np.array_equal(got[:, 1], np.tile(np.arange(n), n))
    assert (got[:, 2] == int((row.astype(np.int64) ** 2).sum())).all() and (got[:, 3] == 255).all()
    ss.close()
    # another set goes through the filter again
    sk2 = synth.make_sketches_numpy(300, 2048, 3000, seed=8, cluster=8)
    n22 = _n2_from_sketches(sk2)
    ss2 = ctx.sketch_set(sk2)
    c2, _ = ctx.pairwise_rows(ss2, n22)
    assert ctx.pairwise_candidates() > 0 and _cells_tuple(c2) == _oracle_sorted(sk2, n22, chunk=192)
    ss2.close()


@pytest.mark.parametrize("filt", ["0", "2"])
def test_row_chunked_shard(ctx, monkeypatch, filt):
    """a shard cut into row chunks (the bound that applies beyond ~1M x 1M cells, lowered here): same cells"""
    sk = synth.make_sketches_numpy(1100, 512, 3000, seed=12, cluster=8)
    n2 = _n2_from_sketches(sk)
    ss = ctx.sketch_set(sk)
    ctx.set_option("pairwise_filter", int(filt))
    ctx.set_option("pairwise_block_cells", 300 * 1100)     # chunks of 256 rows
    whole, _ = ctx.pairwise_rows(ss, n2)
    part, _ = ctx.pairwise_rows(ss, n2, row_begin=130, row_end=901)
    ctx.set_option("pairwise_block_cells", 1 << 40)
    want = _oracle_sorted(sk, n2, chunk=192)
    assert _cells_tuple(whole) == want
    assert _cells_tuple(part) == [t for t in want if 130 <= t[0] < 901]
    ss.close()


@pytest.mark.parametrize("d", [2048, 4096])
def test_baseline_config3_and_4_full_size(ctx, monkeypatch, d):
    """BASELINE.json configs[2] and the size of configs[3] (100 000 sketches, d = 2048 / 4096, magnitudes of
    50k-hash samples): the two-stage comparison and the exact kernel give the same 1.6 M cells bit for bit;
    size-independent properties; 24 rows against the oracle."""
    import torch
    n = 100_000
    sk_t = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
    ss_t = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        ctx.sumsq(sk_t, out=ss_t)
        ss_host = ss_t.cpu().numpy()
        n2 = np.array([orc.norm_sq_from_text(orc.format_norm(float(np.sqrt(v / d)))) for v in ss_host])
        n2_t = torch.from_numpy(n2).to("cuda")
        sset = ctx.sketch_set(sk_t)
        assert sset.limbs == 2
        out = []
        for f in ("1", "0"):
            ctx.set_option("pairwise_filter", int(f))
            cells_t = torch.empty((1 << 22, 4), dtype=torch.int32, device="cuda")
            _, cnt = ctx.pairwise_rows(sset, n2_t, cells_out=cells_t)
            assert (ctx.pairwise_candidates() > 0) == (f == "1")
            ctx.synchronize()
            out.append(cells_t[:cnt].cpu().numpy())
        sset.close()
        rows_chk = slice(77_000, 77_024)
        sk_rows = sk_t[rows_chk].cpu().numpy()
        sk_all = sk_t.cpu().numpy()
    finally:
        ctx.set_stream(None)
    two, exact = out
    assert np.array_equal(two, exact) and len(two) >= 16 * n
    rows, cols = two[:, 0].astype(np.int64), two[:, 1].astype(np.int64)
    key = rows * n + cols
    assert np.all(np.diff(key) > 0)                                          # sorted, no duplicates
    assert np.array_equal(np.sort(cols * n + rows), key)                     # symmetric
    diag = two[rows == cols]
    assert len(diag) == n and np.all(diag[:, 3] == 255)
    assert ((rows // 16) == (cols // 16)).sum() == 16 * n                    # every cluster pair kept
    want = _oracle_sorted(sk_all, n2, row_begin=rows_chk.start, row_end=rows_chk.stop, chunk=192)
    got = [tuple(int(x) for x in c) for c in two[(rows >= rows_chk.start) & (rows < rows_chk.stop)]]
    assert got == want and np.array_equal(sk_rows, sk_all[rows_chk])


def test_nan_and_infinite_norms(ctx, monkeypatch):
    """a norm that parsed as nan or inf ("nan" in vector_norms.txt) makes every threshold of that row nan / inf:
    the reference keeps nothing in its row and column; both comparison paths agree with the oracle"""
    d = 256
    rng = np.random.default_rng(41)
    sk = rng.integers(-300, 300, (200, d)).astype(np.int32)
    sk[100:] = sk[:100]
    n2 = _n2_from_sketches(sk)
    n2[[3, 50]] = np.nan
    n2[[51, 120]] = np.inf
    ss = ctx.sketch_set(sk)
    for keep in (_capi.KEEP_INT32, _capi.KEEP_INT16):
        skx = sk if keep == _capi.KEEP_INT32 else sk.astype(np.int16)
        want = _oracle_sorted(skx, n2, chunk=192)
        assert not any(r in (3, 50, 51, 120) or c in (3, 50, 51, 120) for r, c, _, _ in want) and len(want) > 300
        for f in ("2", "0"):
            ctx.set_option("pairwise_filter", int(f))
            got, _ = ctx.pairwise_rows(ss, n2, keep_mode=keep)
            assert _cells_tuple(got) == want
    ss.close()


def test_baseline_config5_full_size(ctx, tmp_path):
    """BASELINE.json configs[4] on one card: 1 000 000 sketches, d = 2048 (1e12 cells; 8.2 GB of int32 sketches,
    4.1 GB of limb planes + 2 GB coarse plane, all resident).  The oracle cannot do the whole, so:
      * size-independent properties of the two-stage result (sorted, symmetric, every row keeps its 16 cluster
        mates incl. itself with q = 255, indices in range);
      * the exact kernel on a 65 536-row stripe gives the stripe's cells of the two-stage run bit for bit;
      * 24 rows against the oracle, column slab by column slab (the sketches go through the host 100k rows at a time);
      * the executable on the same DB (written slab-wise, 8.2 GB vectors.bin) with --num_shards 8 (the shard's ~2.4 M kept
        cells streamed out as device-encoded rows): same cells as the library call."""
    import subprocess
    import torch
    n, d = 1_000_000, 2048
    sk_t = synth.make_sketches_torch(n, d, 50_000, seed=4567, device="cuda")
    ss_t = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    try:
        ctx.sumsq(sk_t, out=ss_t)
        norms = np.sqrt(ss_t.cpu().numpy().astype(np.float64) / d)
        norm_txt = [orc.format_norm(float(x)) for x in norms]
        n2 = np.array([float(t) ** 2 for t in norm_txt])
        n2_t = torch.from_numpy(n2).to("cuda")
        sset = ctx.sketch_set(sk_t)
        assert sset.limbs == 2
        cells_t = torch.empty((n * 24, 4), dtype=torch.int32, device="cuda")
        ctx.set_option("pairwise_filter", 1)
        _, cnt = ctx.pairwise_rows(sset, n2_t, cells_out=cells_t)
        assert ctx.pairwise_candidates() > 0
        torch.cuda.synchronize()
        two = cells_t[:cnt].cpu().numpy()
        rb, re = 458_752, 458_752 + 65_536
        ctx.set_option("pairwise_filter", 0)
        _, cnt_x = ctx.pairwise_rows(sset, n2_t, row_begin=rb, row_end=re, cells_out=cells_t)
        assert ctx.pairwise_candidates() == 0
        torch.cuda.synchronize()
        exact = cells_t[:cnt_x].cpu().numpy()
        sset.close()
        del cells_t
    finally:
        ctx.set_stream(None)
    rows, cols = two[:, 0].astype(np.int64), two[:, 1].astype(np.int64)
    assert rows.min() == 0 and rows.max() == n - 1 and cols.min() == 0 and cols.max() == n - 1
    key = rows * n + cols
    assert np.all(np.diff(key) > 0)                                          # sorted by (row, col), no duplicates
    assert np.array_equal(np.sort(cols * n + rows), key)                     # symmetric kept set
    assert int(((rows // 16) == (cols // 16)).sum()) == 16 * n               # every cluster pair kept
    diag = two[rows == cols]
    assert len(diag) == n and np.all(diag[:, 3] == 255)
    lo, hi = np.searchsorted(rows, [rb, re])
    assert np.array_equal(two[lo:hi], exact)                                 # exact kernel == two-stage on the stripe

    # 24 rows against the oracle, slab by slab
    q0 = 777_000
    qrows = sk_t[q0:q0 + 24].cpu().numpy()
    want = []
    for s0 in range(0, n, 100_000):
        slab = sk_t[s0:s0 + 100_000].cpu().numpy()
        stack = np.concatenate([qrows, slab])
        n2s = np.concatenate([n2[q0:q0 + 24], n2[s0:s0 + 100_000]])
        for c in orc.pairwise_rows(stack, n2s, row_begin=0, row_end=24, chunk=192, threads=8):
            if c["col"] >= 24:
                want.append((q0 + int(c["row"]), s0 + int(c["col"]) - 24, int(c["dot"]), int(c["q"])))
    lo, hi = np.searchsorted(rows, [q0, q0 + 24])
    assert sorted(want) == [tuple(int(x) for x in c) for c in two[lo:hi]]

    # the executable, one of 8 shards: ~2.4M kept cells streamed out through the pinned buffers
    db = str(tmp_path / "db1m") + "/"
    os.makedirs(db)
    with open(db + "vectors.bin", "wb") as f:
        for s0 in range(0, n, 100_000):
            sk_t[s0:s0 + 100_000].cpu().numpy().tofile(f)
    del sk_t
    torch.cuda.empty_cache()
    with open(db + "vector_norms.txt", "w") as f:
        f.write("".join("s%d %s\n" % (i, t) for i, t in enumerate(norm_txt)))
    open(db + "dimension.txt", "w").write("%d\n" % d)
    open(db + "dtype.txt", "w").write("int32\n")
    out = str(tmp_path / "idx1m")
    bin_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "metagenome_vector_sketches_amd", "bin")
    r = subprocess.run([os.path.join(bin_dir, "pairwise_comp_optimized"), "--db", db, "--max_memory_gb", "0.02",
                        "--num_threads", "8", "--output_folder", out, "--num_shards", "8", "--shard_idx", "3"],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Total vectors: 1000000" in r.stdout and "Shard 3 processing rows 375000 to 500000" in r.stdout
    os.remove(db + "vectors.bin")
    r = subprocess.run([os.path.join(bin_dir, "mvs_dump_matrix"), os.path.join(out, "shard_3")], capture_output=True,
                       text=True)
    assert r.returncode == 0, r.stderr
    got = np.array([[int(t) for t in l.split()] for l in r.stdout.strip().split("\n") if l], dtype=np.int64)
    lo, hi = np.searchsorted(rows, [375_000, 500_000])
    assert hi - lo > 1_000_000 and len(got) == hi - lo
    assert np.array_equal(got, two[lo:hi][:, [0, 1, 3]].astype(np.int64))


def test_dots_match_the_references_eigen_product(ctx, gold):
    """the dense dot kernels (matrix-core and vector-ALU paths) against the int32 products the reference's own vendored
    Eigen computed (tests/golden/kat.json: eigen_gemm_cases; src/pairwise_comp_optimized.cpp:135): same bits, wrap-around
    included -- the HIP path pinned against reference code directly, not through the oracle"""
    for case in gold.kat["eigen_gemm_cases"]:
        bi, bj = gold.eigen_blocks(case)
        sk = np.concatenate([bi, bj])
        ss = ctx.sketch_set(sk)
        ci, cj = case["c_i"], case["c_j"]
        want = np.array(case["dots"], dtype=np.int32).reshape(ci, cj)
        for algo in (0, 1):
            got = ctx.pairwise_dots(ss, 0, ci, ci, ci + cj, algo=algo)
            assert np.array_equal(got, want), (case["d"], case["magnitude"], algo, ss.limbs)
        ss.close()


@pytest.mark.parametrize("ev", [0, 1, 2, 3])
def test_recheck_kernel_variants(ctx, ev):
    """every re-check kernel (tree reduction = default, one butterfly per pair, quarter wave per pair, 16 per round) on a
    round structure with a partial last round: the oracle's cells"""
    sk = synth.make_sketches_numpy(777, 1000, 3000, seed=21, cluster=8)
    n2 = _n2_from_sketches(sk)
    ss = ctx.sketch_set(sk)
    ctx.set_option("pairwise_filter", 2)
    ctx.set_option("exact_variant", ev)
    got, _ = ctx.pairwise_rows(ss, n2)
    assert ctx.pairwise_candidates() > 777 * 4
    assert _cells_tuple(got) == _oracle_sorted(sk, n2, chunk=192)
    ss.close()
