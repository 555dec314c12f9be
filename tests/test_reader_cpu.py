"""CPU: the reader stack (query_pc_mat CLI, pc_mat:: library, pybind11 read_pc_mat_module) over shard
folders in this build's codec.  Shards are written from the toy fixture cells with mvs_write_matrix (the
same write_shard() pairwise_comp_optimized uses), so no device is needed.  Semantics follow
src/read_pc_mat_cmp.cpp:989-1171 and src/query_pc_mat.cpp; the codec bytes themselves are this build's own
(the reference's `bits` library is absent: byte parity unpinned)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "metagenome_vector_sketches_amd")
BIN = os.path.join(PKG, "bin")


def run(*args, cwd=None):
    return subprocess.run(list(args), capture_output=True, text=True, cwd=cwd)


@pytest.fixture(scope="module")
def toy_index(tmp_path_factory, gold):
    d = tmp_path_factory.mktemp("reader")
    db = d / "toy_db"
    db.mkdir()
    (db / "vector_norms.txt").write_text(gold.norms_txt)
    cells = gold.cells()
    txt = d / "cells.txt"
    txt.write_text("".join("%d %d %d\n" % (r, c, q) for r, c, _, q in cells))
    out = {}
    for shards in (1, 3):
        m = d / ("idx%d" % shards)
        r = run(os.path.join(BIN, "mvs_write_matrix"), str(txt), str(m), "61", str(shards))
        assert r.returncode == 0, r.stderr
        out[shards] = str(m)
    by_row = {}
    for r, c, _, q in cells:
        by_row.setdefault(r, []).append((c, q))
    return d, str(db) + "/", out, by_row


def _expected_neighbors(by_row, row, names):
    nb = sorted(by_row.get(row, []), key=lambda t: -t[1])          # stable: ties keep ascending column
    return [names[c] for c, _ in nb], [np.float32(q / 255.0) for _, q in nb]


def test_dump_roundtrip_all_shards(toy_index, gold):
    d, db, idx, by_row = toy_index
    want = sorted((r, c, q) for r, c, _, q in gold.cells())
    for shards, folder in idx.items():
        got = []
        for k in range(shards):
            r = run(os.path.join(BIN, "mvs_dump_matrix"), os.path.join(folder, "shard_%d" % k))
            assert r.returncode == 0, r.stderr
            got += [tuple(int(t) for t in l.split()) for l in r.stdout.strip().split("\n") if l]
        assert got == want


def test_pybind_module_query(toy_index, gold, tmp_path):
    d, db, idx, by_row = toy_index
    sys.path.insert(0, PKG)
    import read_pc_mat_module as rpc          # the reference's module name (src/bindings.cpp:110)
    qf = tmp_path / "q.txt"
    # ids, a numeric index (taken as a row index first, read_pc_mat_cmp.cpp:674-689), comment, blank, unknown id
    qf.write_text("DRR000821\n# comment\n\n  DRR000837  \n10\nNOT_THERE\n")
    for shards, folder in idx.items():
        res = rpc.query(folder, db, str(qf))
        assert [r["id"] for r in res] == ["DRR000821", "DRR000837", gold.names[10]]
        for r in res:
            row = gold.names.index(r["id"])
            ids, jac = _expected_neighbors(by_row, row, gold.names)
            assert list(r["neighbor_ids"]) == ids
            assert r["jaccard_similarities"].dtype == np.float32
            assert np.array_equal(r["jaccard_similarities"], np.array(jac, dtype=np.float32))
            assert r["neighbor_ids"][0] == r["id"] and r["jaccard_similarities"][0] == 1.0   # self pair, q = 255


def test_pybind_module_query_sliced(toy_index, gold, tmp_path):
    d, db, idx, by_row = toy_index
    sys.path.insert(0, PKG)
    import read_pc_mat_module as rpc
    rows = [gold.names[i] for i in (6, 20, 60, 0)]
    cols = [gold.names[i] for i in (6, 10, 22, 20, 59)]
    rf, cf = tmp_path / "rows.txt", tmp_path / "cols.txt"
    rf.write_text("\n".join(rows) + "\n")
    cf.write_text("\n".join(cols) + "\n")
    for shards, folder in idx.items():
        res = rpc.query_sliced(folder, db, str(rf), str(cf))
        assert list(res["row-list"]) == rows and list(res["col-list"]) == cols
        for rn in rows:
            have = dict(by_row.get(gold.names.index(rn), []))
            want = [np.float32(have.get(gold.names.index(cn), 0) / 255.0) for cn in cols]   # absent cell -> 0
            assert np.array_equal(np.array(res["jac-dict"][rn], dtype=np.float32), np.array(want, dtype=np.float32))


def test_query_pc_mat_cli_regular(toy_index, gold, tmp_path):
    d, db, idx, by_row = toy_index
    exe = os.path.join(BIN, "query_pc_mat")
    qf = tmp_path / "query_strs.txt"
    qf.write_text("DRR000821\nDRR000837\n")          # the reference's test/query_strs.txt
    out = tmp_path / "toy_neighbors.txt"
    r = run(exe, "--matrix", idx[3], "--db", db, "--query_file", str(qf), "--write_to_file", str(out), "--show_all")
    assert r.returncode == 0, r.stderr
    assert "Total vectors loaded: 61" in r.stdout and "Query completed in" in r.stdout
    for name in ("DRR000821", "DRR000837"):
        lines = (tmp_path / (name + "_toy_neighbors.txt")).read_text().strip().split("\n")
        assert lines[0] == "ID\tJaccard"
        ids, jac = _expected_neighbors(by_row, gold.names.index(name), gold.names)
        assert [l.split("\t")[0] for l in lines[1:]] == ids
        assert np.allclose([float(l.split("\t")[1]) for l in lines[1:]], jac, rtol=1e-5)
    # --query_ids with numeric indices (test/query_ids.txt: 10, 12), printed to screen, top 3
    r = run(exe, "--matrix", idx[1], "--db", db, "--query_ids", "10", "12", "--top", "3")
    assert r.returncode == 0, r.stderr
    assert ("Query: %s #Neighbors: %d" % (gold.names[10], len(by_row[10]))) in r.stdout
    assert "Top 3 neighbors:" in r.stdout and ("1. Neighbor: %s Jaccard Similarity: 1" % gold.names[10]) in r.stdout


def test_query_pc_mat_cli_sliced_and_errors(toy_index, gold, tmp_path):
    d, db, idx, by_row = toy_index
    exe = os.path.join(BIN, "query_pc_mat")
    rows = [gold.names[i] for i in (6, 20, 33)]
    cols = [gold.names[i] for i in (6, 10, 22, 20)]
    rf, cf = tmp_path / "row_file.txt", tmp_path / "col_file.txt"
    rf.write_text("\n".join(rows) + "\n")
    cf.write_text("\n".join(cols) + "\n")
    npy = tmp_path / "row_col.npy"
    r = run(exe, "--matrix", idx[3], "--db", db, "--row_file", str(rf), "--col_file", str(cf), "--write_to_file",
            str(npy), "--batch_size", "2")
    assert r.returncode == 0, r.stderr
    a = np.load(str(npy))
    assert a.shape == (3, 4) and a.dtype == np.float32
    for i, rn in enumerate(rows):
        have = dict(by_row.get(gold.names.index(rn), []))
        assert np.array_equal(a[i], np.array([have.get(gold.names.index(c), 0) / 255.0 for c in cols], dtype=np.float32))
    csv = tmp_path / "row_col.csv"
    r = run(exe, "--matrix", idx[1], "--db", db, "--row_file", str(rf), "--col_file", str(cf), "--write_to_file", str(csv))
    assert r.returncode == 0 and csv.read_text().split("\n")[0] == "Accession," + ",".join(cols) + ","
    # the reference's own test/row_file.txt / col_file.txt name accessions that are not in the toy set
    rf.write_text("DRR005002\nSRR9992156\n")
    r = run(exe, "--matrix", idx[1], "--db", db, "--row_file", str(rf), "--col_file", str(cf))
    assert r.returncode == 1 and "Empty row or col accessions." in r.stderr and "Aborting..." in r.stderr
    # option errors
    assert run(exe, "--help").returncode == 0
    assert run(exe, "--bogus").returncode == 1
    r = run(exe, "--db", db, "--query_ids", "1")
    assert r.returncode == 1 and "matrix folder is required" in r.stderr
    r = run(exe, "--matrix", idx[1], "--db", db)
    assert r.returncode == 1 and "No query files given." in r.stderr
    r = run(exe, "--matrix", idx[1], "--db", db, "--query_ids", "1", "--write_to_file", str(tmp_path / "x.npy"))
    assert r.returncode == 1 and "Expected: csv, tsv or txt." in r.stderr


def test_empty_and_missing_shards(tmp_path, gold):
    """a shard with no kept rows must not crash the reader (the reference dereferences row 0 there)"""
    sys.path.insert(0, PKG)
    import read_pc_mat_module as rpc
    db = tmp_path / "db"
    db.mkdir()
    (db / "vector_norms.txt").write_text("a 1\nb 2\nc 3\nd 4\n")
    cells = tmp_path / "cells.txt"
    cells.write_text("0 0 255\n0 1 40\n1 0 40\n1 1 255\n")          # rows 2, 3 (shard 1) have nothing
    m = tmp_path / "idx"
    assert run(os.path.join(BIN, "mvs_write_matrix"), str(cells), str(m), "4", "2").returncode == 0
    q = tmp_path / "q.txt"
    q.write_text("a\nc\n")
    res = rpc.query(str(m), str(db), str(q))
    assert res[0]["id"] == "a" and list(res[0]["neighbor_ids"]) == ["a", "b"]
    assert len(res) == 2 and list(res[1].get("neighbor_ids", [])) == []


def test_reader_and_codec_under_address_sanitizer(toy_index, gold, tmp_path):
    """`make asan`: the device-free host tools built with -fsanitize=address,undefined; the codec self-test (codec round
    trips, shard writer on 1 and 7 threads, hash text parser, CSR cache) and a query of each kind must run clean"""
    d, db, idx, by_row = toy_index
    r = run("make", "-s", "-C", os.path.join(PKG, "csrc"), "asan")
    assert r.returncode == 0, r.stderr
    asan = os.path.join(BIN, "asan")
    r = run(os.path.join(asan, "mvs_codec_selftest"))
    assert r.returncode == 0 and "mvs_codec_selftest ok" in r.stdout, r.stderr
    r = run(os.path.join(asan, "mvs_codec_selftest_tsan"))      # the shard writer's threads under ThreadSanitizer
    assert r.returncode == 0 and "mvs_codec_selftest ok" in r.stdout and "ThreadSanitizer" not in r.stderr, r.stderr
    qf = tmp_path / "q.txt"
    qf.write_text("DRR000821\n10\nNOT_THERE\n")
    r = run(os.path.join(asan, "query_pc_mat"), "--matrix", idx[3], "--db", db, "--query_file", str(qf), "--top", "5")
    assert r.returncode == 0 and "ERROR: AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    rf, cf = tmp_path / "r.txt", tmp_path / "c.txt"
    rf.write_text("\n".join(gold.names[i] for i in (6, 20, 60)) + "\n")
    cf.write_text("\n".join(gold.names[i] for i in (6, 10, 22)) + "\n")
    r = run(os.path.join(asan, "query_pc_mat"), "--matrix", idx[1], "--db", db, "--row_file", str(rf), "--col_file", str(cf),
            "--write_to_file", str(tmp_path / "s.npy"), "--batch_size", "2")
    assert r.returncode == 0 and "AddressSanitizer" not in r.stderr and "runtime error" not in r.stderr, r.stderr
    assert np.load(str(tmp_path / "s.npy")).shape == (3, 3)
