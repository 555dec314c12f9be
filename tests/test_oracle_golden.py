"""CPU: the oracle (oracle/mvs_oracle.c) against the committed reference fixtures.

Fixtures come from the reference binaries (oracle/_ref, built from /root/reference/src) run by
tests/golden/make_golden.py, and from the values SURVEY.md section 4 recorded."""
import hashlib

import numpy as np

from oracle import pyoracle as orc


def test_splitmix_kat(gold):
    # splitmix64 finaliser as used by src/random_projection.cpp:13-17, x = 1
    assert orc.splitmix64(1) == int(gold.kat["splitmix64_of_1"], 16)


def _parse_sp(stdout):
    return [np.array([float(t) for t in line.split(" ")], dtype=np.float64) if line else np.zeros(0)
            for line in stdout.split("\n")[:-1]]


def test_standalone_projection_cases(gold):
    """src/standalone_projection.cpp:28-43: one line of d floats per input line."""
    for key, case in gold.kat["standalone_projection"].items():
        d = case["d"]
        lines = case["input"].split("\n")[:-1]
        want = _parse_sp(case["stdout"])
        assert len(want) == len(lines), key
        for line, w in zip(lines, want):
            hashes = sorted(set(int(t) for t in line.split()))  # unordered_set dedups
            got = orc.project(np.array(hashes, dtype=np.uint64), d)
            assert len(w) == d, key
            assert np.array_equal(got.astype(np.float64), w), key


def test_toy_projection_bit_exact(gold):
    """all 61 toy samples, d=2048, against the reference's vectors.bin"""
    got = orc.project_csr(gold.hashes, gold.offsets, 2048, threads=4)
    assert np.array_equal(got, gold.vectors)
    fast = orc.project_csr(gold.hashes, gold.offsets, 2048, threads=4, fast=True)
    assert np.array_equal(fast, gold.vectors)
    assert hashlib.sha256(got.tobytes()).hexdigest() == gold.kat["toy_vectors_sha256_sorted_by_name"]
    for i, n in enumerate(gold.names):
        dg = gold.digests[n]
        assert hashlib.sha256(got[i].tobytes()).hexdigest() == dg["sha256"]
        assert int(got[i].sum()) == dg["sum"] and orc.sumsq(got[i]) == dg["sumsq"]
        assert gold.offsets[i + 1] - gold.offsets[i] == dg["n_hashes"]


def test_toy_norms_within_tolerance(gold):
    """vector_norms.txt is float32 + -ffast-math in the reference and not bit-reproducible
    (SURVEY.md 8c): tolerance 1e-5 relative, i.e. <= 1 unit in the 6th significant digit."""
    exact_match = 0
    for i, line in enumerate(gold.norm_lines()):
        name, txt = line.split(" ")
        assert name == gold.names[i]
        ref = float(txt)
        for mine in (orc.norm(gold.vectors[i]), orc.norm_f32path(gold.vectors[i])):
            assert abs(mine - ref) <= 1e-5 * max(ref, 1e-30) + 1e-12
        exact_match += orc.format_norm(orc.norm(gold.vectors[i])) == txt
    assert exact_match >= len(gold.names) - 3


def test_toy_pairwise_cells(gold):
    """int32 path, chunk 192, one shard: 1291 of 3721 cells."""
    n2 = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in gold.norm_lines()])
    cells = orc.pairwise_rows(gold.vectors, n2, chunk=192, threads=4)
    want = gold.cells()
    assert len(cells) == gold.kat["survey_kept_cells"]["int32"] == len(want)
    got = [(int(c["row"]), int(c["col"]), int(c["dot"]), int(c["q"])) for c in cells]
    assert got == want
    # the rows SURVEY.md recorded from the reference binary itself
    by_row = {}
    for r, c, dot, q in got:
        by_row.setdefault(gold.names[r], []).append((gold.names[c], q))
    # SURVEY listed them in readdir order; compare as sets of (col, q) prefixes by name
    for rname, pin in gold.kat["survey_pairwise_pins"].items():
        have = dict(by_row[rname])
        for cname, q in zip(pin["cols"], pin["q"]):
            assert have[cname] == q


def test_toy_pairwise_cells_int16(gold):
    n2 = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in gold.norm_lines()])
    v16 = orc.saturate_i16(gold.vectors)
    assert np.array_equal(v16.astype(np.int32), gold.vectors)  # toy max |v| = 1263
    cells = orc.pairwise_rows(v16, n2, chunk=2048, threads=4)
    got = sorted((int(c["row"]), int(c["col"]), int(c["dot"]), int(c["q"])) for c in cells)
    assert len(got) == gold.kat["survey_kept_cells"]["int16"]
    assert got == sorted(gold.cells(int16=True))


def test_pairwise_order_and_shards(gold):
    """order = (i-chunk, j-chunk, i, j); shards = row ranges (src/pairwise_comp_optimized.cpp:938-980)"""
    n2 = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in gold.norm_lines()])
    full = orc.pairwise_rows(gold.vectors, n2, chunk=192)
    parts = []
    for k in range(2):
        b, e = orc.shard_rows(len(gold.names), 2, k)
        parts.append(orc.pairwise_rows(gold.vectors, n2, row_begin=b, row_end=e, chunk=192))
    assert (b, e) == (31, 61)
    assert np.array_equal(np.concatenate(parts), full)
    small = orc.pairwise_rows(gold.vectors, n2, chunk=7)  # different tiling -> different order, same set
    assert sorted(map(tuple, small.tolist())) == sorted(map(tuple, full.tolist()))
    assert not np.array_equal(small, full)


def test_chunk_formula():
    assert orc.chunk_size(12, 2048) == 192
    assert orc.chunk_size(12, 4096) == 48
    assert orc.chunk_size(64, 2048) == 1024


def test_keep_and_quantize_edge_cases():
    lib = orc.load()
    # truncation toward zero: -2047/2048 -> 0, not -1; 0 > 0 is false
    assert lib.mvs_oracle_keep_i32(-2047, 2048, 0.0, 0.0) == 0
    assert lib.mvs_oracle_keep_i32(2047, 2048, 0.0, 0.0) == 0          # trunc(0.9995) = 0
    assert lib.mvs_oracle_keep_i16(2047, 2048, 0.0, 0.0) == 1          # floating division keeps it
    assert lib.mvs_oracle_keep_i32(2048, 2048, 0.0, 0.0) == 1
    # self pair: J = 1 -> 255
    assert lib.mvs_oracle_quantize(2048 * 100, 2048, 100.0, 100.0) == 255
    # J slightly above 1 clamps
    assert lib.mvs_oracle_quantize(2048 * 101, 2048, 100.0, 100.0) == 255
    # round half away from zero: J*255 = 0.5 -> 1
    assert lib.mvs_oracle_quantize(1, 1, 255.0, 255.0 + 1.0) == 1


def test_dot_wraps_mod_2_32():
    a = np.full(2048, 70000, dtype=np.int32)
    lib = orc.load()
    got = lib.mvs_oracle_dot_i32(a.ctypes.data, a.ctypes.data, 2048)
    want = (70000 * 70000 * 2048) % (1 << 32)
    want = want - (1 << 32) if want >= (1 << 31) else want
    assert got == want


def test_saturate_i16():
    v = np.array([0, 1, -1, 32767, 32768, -32768, -32769, 2**31 - 1, -2**31], dtype=np.int32)
    assert orc.saturate_i16(v).tolist() == [0, 1, -1, 32767, 32767, -32768, -32768, 32767, -32768]


def test_toy_other_dimensions(gold):
    """d = 4096 (BASELINE config 4) and d = 100 (tail block, d % 64 != 0): reference digests"""
    for d in (4096, 100):
        got = orc.project_csr(gold.hashes, gold.offsets, d, threads=4, fast=True)
        assert hashlib.sha256(got.tobytes()).hexdigest() == gold.kat["toy_vectors_sha256_d%d" % d]


def test_int32_product_matches_the_references_eigen(gold):
    """src/pairwise_comp_optimized.cpp:135 evaluated by the reference's vendored Eigen (fixture made by
    tests/golden/make_golden_eigen.py): the oracle's wrapping int32 dot gives the same bits, wrap-around included"""
    cases = gold.kat["eigen_gemm_cases"]
    assert len(cases) >= 5
    lib = orc.load()
    wrapped = 0
    for case in cases:
        bi, bj = gold.eigen_blocks(case)
        got = [lib.mvs_oracle_dot_i32(bi[i].ctypes.data, bj[j].ctypes.data, case["d"])
               for i in range(case["c_i"]) for j in range(case["c_j"])]
        assert got == case["dots"], (case["d"], case["magnitude"])
        exact = bi.astype(object) @ bj.astype(object).T
        wrapped += int(sum(abs(int(x)) >= 2**31 for x in exact.ravel()))
    assert wrapped >= 20          # the fixture does exercise the modulo-2^32 behaviour


def test_own_norm_definition_changes_no_toy_cell(gold):
    """The reference writes vector_norms.txt through a float32 / -ffast-math path that is not bit-reproducible
    (SURVEY.md 8c); this build writes sqrt(double(sum v^2) / d) with %g.  What that costs end to end, measured on the
    toy DB: a few norm lines differ in the 6th digit, and the all-vs-all result built on OUR norms is compared cell by
    cell with the one built on the REFERENCE's norms -- the kept set must be identical, q may move by at most one level
    on the few cells whose rows' norms differ."""
    ref_lines = gold.norm_lines()
    own = [orc.format_norm(orc.norm(v)) for v in gold.vectors]
    differing = [i for i, l in enumerate(ref_lines) if l.split(" ")[1] != own[i]]
    assert len(differing) <= 4
    for i in differing:
        a, b = float(ref_lines[i].split(" ")[1]), float(own[i])
        assert abs(a - b) <= 1e-5 * a
    n2_ref = np.array([orc.norm_sq_from_text(l.split(" ")[1]) for l in ref_lines])
    n2_own = np.array([orc.norm_sq_from_text(t) for t in own])
    c_ref = orc.pairwise_rows(gold.vectors, n2_ref, chunk=192, threads=2)
    c_own = orc.pairwise_rows(gold.vectors, n2_own, chunk=192, threads=2)
    k_ref = {(int(c["row"]), int(c["col"])): int(c["q"]) for c in c_ref}
    k_own = {(int(c["row"]), int(c["col"])): int(c["q"]) for c in c_own}
    assert set(k_ref) == set(k_own) and len(k_ref) == 1291                    # same kept cells
    moved = [k for k in k_ref if k_ref[k] != k_own[k]]
    assert all(abs(k_ref[k] - k_own[k]) <= 1 and (k[0] in differing or k[1] in differing) for k in moved)
    assert len(moved) <= 8


def test_pairwise_restatement_equals_the_references_own_functions(gold):
    """a5/a6/a10/a11 pinned: oracle/_ref/ref_pairwise32 / ref_pairwise16 are the reference's OWN load_matrix_block,
    compute_sparse_dot_products_optimized (src/pairwise_comp_optimized.cpp:33-160), Matrix16, load_matrix_block_int16 and
    compute_sparse_dot_products_optimized_16 (_16bits.cpp:40-244), compiled from line ranges (oracle/Makefile
    ref_pairwise).  Every recorded run -- toy DB, threshold edges, wrapping products, odd norms, d = 100, many tiles,
    several shards -- is reproduced by the restatement cell for cell IN THE REFERENCE'S ORDER (:949-982)."""
    cases = gold.ref_pairwise_cases()
    assert len(cases) >= 13 and sum(len(c["runs"]) for c in cases.values()) >= 23
    for name, c in cases.items():
        for run in c["runs"]:
            b, e = orc.shard_rows(len(c["vectors"]), run["num_shards"], run["shard_idx"])
            if c["elem"] == 4:
                assert orc.chunk_size(run["max_memory_gb"], c["d"]) == run["chunk"], name
            got = orc.pairwise_rows(c["vectors"], c["norms_sq"], row_begin=b, row_end=e, chunk=run["chunk"], threads=4)
            got3 = np.stack([got["row"], got["col"], got["dot"]], axis=1).astype(np.int64).reshape(-1, 3)
            assert np.array_equal(got3, run["cells"]), (name, run["num_shards"], run["shard_idx"])
            text = "".join("%d %d %d\n" % tuple(r) for r in got3.tolist())
            assert hashlib.sha256(text.encode()).hexdigest() == run["cells_sha256"], name


def test_quantiser_restatement_equals_the_references_writer_head(gold):
    """a9 / a12 arithmetic pinned: the same binaries in `rows` mode pass the kept cells through the bits-free first lines
    of the reference's WRITERS, included as reference text by the recipe -- src/pairwise_comp_optimized.cpp:654-672
    (inter = P / d, J = inter / (n2r + n2c - inter), min(J, 1), q = uint16(round(255 J)), grouped by row) and
    _16bits.cpp:260-264 + :274-280 (grouping, round(dot / d)).  The restatement's q / rounded dot equal the reference's for
    every cell of every case; the columns of a row come out ascending (what :720 relies on).  What follows those lines in
    the writers (bits:: codecs, file bytes) stays unpinned."""
    cases = gold.ref_pairwise_cases()
    n_cells = 0
    for name, c in cases.items():
        head = c["writer_head"]
        whole = [r for r in c["runs"] if r["num_shards"] == 1][0]
        got = orc.pairwise_rows(c["vectors"], c["norms_sq"], chunk=whole["chunk"], threads=4)
        if c["elem"] == 4:
            mine = {(int(x["row"]), int(x["col"])): int(x["q"]) for x in got}
        else:                                       # std::round: half away from zero (_16bits.cpp:277)
            d = c["d"]
            mine = {(int(x["row"]), int(x["col"])): int(np.sign(int(x["dot"])) * np.floor(abs(int(x["dot"])) / d + 0.5))
                    for x in got}
        ref = {(r, col): v for r, col, v in head["cells"].tolist() if (r, col) not in head["undefined"]}
        assert set(mine) - head["undefined"] == set(ref), name
        assert all(mine[k] == v for k, v in ref.items()), name
        rows = head["cells"][:, 0].tolist()
        assert list(dict.fromkeys(rows)) == head["row_order"] and len(set(rows)) == len(head["row_order"])   # rows contiguous
        for r in head["row_order"]:
            cols = head["cells"][head["cells"][:, 0] == r, 1]
            assert np.all(np.diff(cols) > 0), (name, r)
        n_cells += len(ref)
    assert n_cells > 40000
    # the pin spans the scale: the threshold's q = 13 ... the diagonal's 255, and 16-bit values where wrapped products or
    # odd norms make J negative (uint16(round(255 J)) = 65536 - ..., which the 8-bit-or-16-bit row containers of :718-736 carry)
    qs = np.concatenate([c["writer_head"]["cells"][:, 2] for c in cases.values() if c["elem"] == 4])
    assert qs.min() == 13 and 255 in qs and qs.max() > 65000 and len(np.unique(qs)) > 100


def test_toy_cell_fixture_carries_reference_provenance(gold):
    """toy_pairwise_cells*.txt: (row, col, dot) are the reference functions' output (ref_pairwise.json toy runs); the int32
    list's q column is the reference's own quantiser lines (writer head); the int16 list's q is this repository's extension
    (the reference's int16 writer stores round(dot / d))"""
    cases = gold.ref_pairwise_cases()
    for int16, case in ((False, "toy_int32"), (True, "toy_int16")):
        ref = cases[case]["runs"][0]["cells"]
        assert [(r, c, dot) for r, c, dot, _ in gold.cells(int16=int16)] == [tuple(x) for x in ref.tolist()]
    wq = {(r, c): q for r, c, q in cases["toy_int32"]["writer_head"]["cells"].tolist()}
    assert all(wq[(r, c)] == q for r, c, _, q in gold.cells()) and len(wq) == 1291
    prov = gold.kat["provenance"]
    assert any(k.startswith("reference functions compiled from line ranges") for k in prov)
    assert any("654-672" in k and "toy_pairwise_cells.txt: column q" in v for k, v in prov.items())
    # SURVEY section 4's 14 (col, q) pins (stand-in build) agree with the reference-text pin
    by_name = {(gold.names[r], gold.names[c]): q for (r, c), q in wq.items()}
    for rname, pin in gold.kat["survey_pairwise_pins"].items():
        for cname, q in zip(pin["cols"], pin["q"]):
            assert by_name[(rname, cname)] == q


def test_low_limb_and_coarse_value_pin_every_value_the_radix_rule_admits():
    """csrc/mvs_pairwise.hip "The high limb on the wire": for every radix m <= 252 and every v with |v| <= 127 m - ceil(m / 2) + 254
    (what radix_keeps_high_limb admits; |v| <= 32 896 = two signed base-256 digits), the coarse value c = clamp(rint(v / m)) and
    the low digit of v give v back -- exhaustively, with the float arithmetic the kernels use."""
    def wrap8(x):
        return ((x + 128) % 256) - 128
    for m in range(1, 253):
        h = (m + 1) // 2
        edge = 127 * m - h
        top = min(edge + 254, 32896)
        v = np.arange(-top, top + 1, dtype=np.int64)
        c = np.clip(np.rint(v.astype(np.float32) * (np.float32(1.0) / np.float32(m))).astype(np.int64), -127, 127)
        l0 = wrap8(v)
        t = np.where(c == 127, edge + 127, np.where(c == -127, -(edge + 127), m * c))
        assert np.array_equal(t + wrap8(l0 - t), v), m
        hi = (v - l0) // 256
        assert hi.min() >= -128 and hi.max() <= 127
    # the radix that just avoids clamping always qualifies, whatever the row's largest |v|
    mx = np.arange(1, 32005, dtype=np.int64)
    m = np.maximum(1, (mx + 126) // 127)
    assert np.all(mx <= 127 * m - (m + 1) // 2 + 254) and m.max() == 252
