"""GPU: query-by-hashes search (metagenome_vector_sketches_amd/search.py, mvs_search_block) against the oracle's
restatement of the reference's scoring (oracle/pyoracle.py:search_scores, src/jaccard.py:117-200).  Tolerance 1e-5
relative on the Jaccard estimates (the reference path itself is float32; FAISS is not installed, so no reference
output exists to pin this against: parity unpinned, semantics restated)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _write_db(folder, gold):
    os.makedirs(folder, exist_ok=True)
    gold.vectors.astype("<i4").tofile(folder + "vectors.bin")
    open(folder + "vector_norms.txt", "w").write(gold.norms_txt)
    open(folder + "dimension.txt", "w").write("2048\n")
    open(folder + "dtype.txt", "w").write("int32\n")


def test_search_matches_float64_restatement(ctx, gold, tmp_path):
    from metagenome_vector_sketches_amd import search
    from oracle import pyoracle as orc
    db = str(tmp_path / "db") + "/"
    _write_db(db, gold)
    # queries: two database samples verbatim, a half-subsample, a disjoint set, an empty set
    rng = np.random.default_rng(1)
    pick = [gold.names.index("DRR000821"), 6]
    qlists = [gold.hashes[gold.offsets[i]:gold.offsets[i + 1]] for i in pick]
    big = gold.hashes[gold.offsets[6]:gold.offsets[7]]
    qlists.append(big[rng.random(len(big)) < 0.5])
    qlists.append(rng.integers(0, 2**62, size=500, dtype=np.uint64))
    qlists.append(np.zeros(0, dtype=np.uint64))
    qf = tmp_path / "queries.txt"
    with open(qf, "w") as f:
        for k, h in enumerate(qlists):
            f.write("q%d:" % k + "".join(" %d" % int(x) for x in h) + "\n")
        f.write("\n")
    j = 0.1
    got = search.search_index(db, str(qf), j, ctx=ctx, verbose=False)
    norms = np.array([float(l.split(" ")[1]) for l in gold.norm_lines()])
    want = []
    for qi, h in enumerate(qlists):
        v = orc.project(np.unique(h), 2048)
        want += [(qi, gold.names[k], jac) for k, jac in orc.search_scores(gold.vectors, norms, v, 2048, j)]
    assert [(a, b) for a, b, _ in got] == [(a, b) for a, b, _ in want]
    assert np.allclose([c for _, _, c in got], [c for _, _, c in want], rtol=1e-5, atol=0)
    first = {}
    for qi, nid, jac in got:
        first.setdefault(qi, (nid, jac))
    assert first[0][0] == "DRR000821" and abs(first[0][1] - 1.0) < 1e-4      # a database sample finds itself
    assert first[1][0] == gold.names[6] and abs(first[1][1] - 1.0) < 1e-4
    assert first[2][0] == gold.names[6] and 0.4 < first[2][1] < 0.6           # half subsample B of A: J = 0.5
    assert 3 not in first and 4 not in first                                  # disjoint and empty queries: nothing


@pytest.mark.parametrize("case", ["int32", "int16", "three-limbs"])
def test_search_larger_db(ctx, tmp_path, case):
    """1300 samples (few query rows x >= 1024 columns: the streaming kernel for two-limb sets), an int16 DB, and a DB whose
    entries need three limbs -- the loader allocates for two, learns the largest |v| from the upload itself and builds the set
    again.  vectors.bin is mapped and goes up in row chunks.  Scores against the float64 restatement."""
    from metagenome_vector_sketches_amd import search
    from oracle import pyoracle as orc
    rng = np.random.default_rng(77)
    n, d, per = 1300, 256, 400
    pools = [rng.integers(0, 2**62, size=per, dtype=np.uint64) for _ in range(n // 10 + 1)]
    lists = []
    for i in range(n):                                          # clusters of 10 share 60 % of their hashes
        own = rng.integers(0, 2**62, size=int(0.4 * per), dtype=np.uint64)
        lists.append(np.unique(np.concatenate([pools[i // 10][:int(0.6 * per)], own])))
    offs = np.zeros(n + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(x) for x in lists])
    vec = ctx.project_csr(np.concatenate(lists), offs, d)
    if case == "three-limbs":
        vec = vec * 700                                         # |v| up to ~60 000
        assert np.abs(vec).max() > 32639
    db = str(tmp_path / "db") + "/"
    os.makedirs(db)
    vec.astype("<i2" if case == "int16" else "<i4").tofile(db + "vectors.bin")
    norms_txt = "".join("s%d %s\n" % (i, orc.format_norm(orc.norm(vec[i]))) for i in range(n))
    open(db + "vector_norms.txt", "w").write(norms_txt)
    open(db + "dimension.txt", "w").write("%d\n" % d)
    open(db + "dtype.txt", "w").write("int16\n" if case == "int16" else "int32\n")
    qlists = [lists[5], lists[777], lists[1299][::2], rng.integers(0, 2**62, size=300, dtype=np.uint64)]
    qf = tmp_path / "queries.txt"
    with open(qf, "w") as f:
        for k, h in enumerate(qlists):
            f.write("q%d:" % k + "".join(" %d" % int(x) for x in h) + "\n")
    # (a DB scaled by 700 against unscaled queries: J(self) = 700 n / (700^2 n + n - 700 n) ~ 1 / 700)
    j = 0.0004 if case == "three-limbs" else 0.08
    got = search.search_index(db, str(qf), j, ctx=ctx, verbose=False)
    norms = np.array([float(l.split(" ")[1]) for l in norms_txt.split("\n") if l])
    want = []
    for qi, h in enumerate(qlists):
        v = orc.project(np.unique(h), d)
        want += [(qi, "s%d" % k, jac) for k, jac in orc.search_scores(vec, norms, v, d, j)]
    assert len(want) >= 10
    assert sorted((a, b) for a, b, _ in got) == sorted((a, b) for a, b, _ in want)
    gj = {(a, b): c for a, b, c in got}
    assert all(abs(gj[(a, b)] - c) <= 1e-5 * abs(c) for a, b, c in want)
    if case != "three-limbs":                                   # (scaled vectors are no projections of anything)
        assert [b for a, b, _ in got if a == 0][0] == "s5" and [b for a, b, _ in got if a == 1][0] == "s777"


def test_search_command_line_resident_index_and_small_hit_buffer(ctx, gold, tmp_path, capsys):
    """the reference's command line (src/jaccard.py:334-362: `search <index_folder> <query_file> -j J -t N`) end to end,
    a resident SearchIndex queried in chunks of two queries, and a hit buffer that is too small: the library reports how
    many hits there are and the block is compared once more with exactly that room -- the same neighbours every way"""
    import torch
    from metagenome_vector_sketches_amd import search
    db = str(tmp_path / "db") + "/"
    _write_db(db, gold)
    picks = [gold.names.index("DRR000821"), 6, 10, 20, 33]
    qf = tmp_path / "queries.txt"
    with open(qf, "w") as f:
        for k, i in enumerate(picks):
            f.write("q%d:" % k + "".join(" %d" % int(x) for x in gold.hashes[gold.offsets[i]:gold.offsets[i + 1]]) + "\n")
    want = search.search_index(db, str(qf), 0.1, ctx=ctx, verbose=False)
    assert {a for a, _, _ in want} == set(range(5)) and len(want) > 20
    assert search.main(["search", db.rstrip("/"), str(qf), "-j", "0.1", "-t", "4"]) == 0
    out = capsys.readouterr().out.split("\n")
    assert out[0].startswith("Version: ") and out[1].startswith("Command line:") and "Query 0:" in out
    first = out[out.index("Query 0:") + 1]
    assert first.startswith("  Neighbor 0: DRR000821 (jaccard: 1.0000), inner_product: 1.0000 ")
    assert sum(l.startswith("  Neighbor ") for l in out) == len(want)
    with search.SearchIndex(db, ctx=ctx, max_queries=2) as idx:
        assert idx.search(str(qf), 0.1, verbose=False) == want          # three chunks: 2 + 2 + 1 queries
        idx._hits = torch.empty((4, 4), dtype=torch.int32, device="cuda")
        assert idx.search(str(qf), 0.1, verbose=False) == want          # too small -> sized from the reported count
        assert idx._hits.shape[0] > 4
    # a malformed query line ends the run with the reference's code 332 (:82-84)
    bad = tmp_path / "bad.txt"
    bad.write_text("a: 1 2 : 3\n")
    assert search.main(["search", db, str(bad)]) == 332


@pytest.mark.usefixtures("restore_options")
@pytest.mark.parametrize("d,n_db", [(2048, 6000), (2100, 4500), (4096, 4300), (9000, 4200), (512, 9000)])
def test_streaming_search_filter_equals_exact_kernel(ctx, d, n_db):
    """k_search_filter -- the query rows' coarse plane resident in LDS (64 / 32 / 16 rows by sketch length), the database
    columns streamed from global memory (fragment-major copy of the coarse plane: one coalesced KiB per wave instruction)
    straight into the matrix cores -- as the first stage of searches and of
    rectangular blocks with 1 .. 1023 rows against >= 4096 columns: the same hits as the exact kernel, for query counts
    around every group size, column ranges that start and end off the 512-column chunks, both keep tests, mirrored
    blocks.  (Reference semantics: src/jaccard.py:117-200; scores are checked against the restatement above.)"""
    import torch
    from metagenome_vector_sketches_amd import _capi, synth
    from oracle import pyoracle as orc
    nq_max = 1023
    sk = synth.make_sketches_numpy(n_db + nq_max, d, 3000, seed=d, cluster=12)
    # queries = the last rows; make them relatives of database rows so that there are hits
    rng = np.random.default_rng(d)
    for q in range(nq_max):
        src = int(rng.integers(0, n_db))
        sk[n_db + q] = sk[src] if q % 3 == 0 else (sk[src] + synth.make_sketches_numpy(1, d, 1500, seed=q)[0])
    sk[n_db + 7] = 0                                                   # an empty query
    n2 = np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(r))) for r in sk])
    ss = ctx.sketch_set(sk)
    assert ss.limbs == 2
    n2_t = torch.from_numpy(n2).to("cuda")
    cells = torch.empty((1 << 21, 4), dtype=torch.int32, device="cuda")

    def run(fn):
        out = []
        # streaming filter (rows resident), exact kernel, tile filter; last field: the streamed columns come from the
        # fragment-major copy of the coarse plane (1) or from the row-major plane (0)
        for filt, stream, variant, fm in ((2, 1, 50, 1), (0, 1, -1, 1), (2, 0, -1, 1), (2, 1, 50, 0), (2, 1, -1, 1)):
            ctx.set_option("pairwise_filter", filt)
            ctx.set_option("search_stream", stream)
            ctx.set_option("fragment_major", fm)
            ctx.set_option("filter_variant", variant)    # 50 by number: up to 1023 rows (by size: up to 640)
            cnt = fn()
            ctx.synchronize()
            out.append((sorted(map(tuple, cells[:cnt].cpu().numpy().tolist())), ctx.pairwise_candidates()))
        assert all(o[0] == out[1][0] for o in out)
        assert out[0][1] > 0 and out[1][1] == 0                      # the two-stage comparison ran / did not run
        assert out[3][1] == out[0][1]                                # either plane layout: the same candidates
        return out[0][0]

    total = 0
    for nq in (1, 5, 16, 17, 33, 64, 65, 128, 129, 200, 256, 300, 385, 512, 513, 1023):
        got = run(lambda: ctx.search_block(ss, n2_t, 0.1, n_db, n_db + nq, 0, n_db, cells))
        total += len(got)
        assert all(n_db <= r < n_db + nq and 0 <= c < n_db for r, c, _, _ in got)
    assert total > 300
    # a column range off the chunk grid, and a low bound (many hits per wave)
    run(lambda: ctx.search_block(ss, n2_t, 0.02, n_db + 3, n_db + 90, 117, n_db - 55, cells))
    # plain and mirrored rectangular blocks through mvs_pairwise_block (reference keep tests, :135-147 / _16bits.cpp:218)
    for flags in (0, _capi.BLOCK_MIRROR_ALL):
        for keep in (_capi.KEEP_INT32, _capi.KEEP_INT16):
            run(lambda: ctx.pairwise_block(ss, n2_t, n_db, n_db + 300, 0, n_db, flags, cells, 0, keep_mode=keep))
    ss.close()


def test_resident_index_keeps_its_coarse_plane_across_searches(tmp_path):
    """every search uploads NEW query sketches into the scratch rows behind the database.  That must not throw
    away what the context derived from the database rows: from the second search on the first stage is the streaming
    filter on the cached coarse plane (pairwise_candidates() > 0), only the query rows are re-derived, and the answers
    stay those of the exact kernel (a fresh index with the filter switched off)."""
    from metagenome_vector_sketches_amd import Context, search, synth
    n, d = 9000, 1024
    rng = np.random.default_rng(42)
    hashes, offsets = synth.make_csr_numpy(n, 400, seed=3, cluster=10, shared=0.5)
    ctx = Context(0)
    sk = ctx.project_csr(hashes, offsets, d)
    db = str(tmp_path / "db") + "/"
    os.makedirs(db)
    sk.astype("<i4").tofile(db + "vectors.bin")
    norms = np.sqrt((sk.astype(np.float64) ** 2).sum(axis=1) / d)
    open(db + "vector_norms.txt", "w").write("".join("s%d %g\n" % (i, x) for i, x in enumerate(norms)))
    open(db + "dimension.txt", "w").write("%d\n" % d)
    open(db + "dtype.txt", "w").write("int32\n")

    def query_file(k, count):
        qf = tmp_path / ("q%d.txt" % k)
        with open(qf, "w") as f:
            for q in range(count):
                i = int(rng.integers(0, n))
                h = hashes[offsets[i]:offsets[i + 1]]
                h = h[rng.random(len(h)) < 0.8] if q % 2 else h          # a subsample of a database sample, or the sample
                f.write("q%d:" % q + "".join(" %d" % int(x) for x in h) + "\n")
        return str(qf)
    files = [query_file(k, c) for k, c in enumerate((3, 40, 7, 300, 1))]
    with search.SearchIndex(db, ctx=ctx, max_queries=512) as exact:
        ctx.set_option("pairwise_filter", 0)
        want = [exact.search(f, 0.1, verbose=False) for f in files]
        ctx.set_option("pairwise_filter", 1)
    assert all(len(w) >= 1 for w in want)
    with search.SearchIndex(db, ctx=ctx, max_queries=512) as idx:
        cands = []
        for f, w in zip(files, want):
            assert idx.search(f, 0.1, verbose=False) == w
            cands.append(ctx.pairwise_candidates())
        # first search: the exact streaming kernel (no plane yet); second: builds the plane; from then on it is kept
        assert cands[0] == 0 and all(c > 0 for c in cands[1:]), cands
    ctx.close()
