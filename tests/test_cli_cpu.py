"""CPU: command-line contracts of the drop-in executables (usage / exit codes as the reference's
clipp parsers give them) and the codec round trips.  No device needed."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin")


def run(*args):
    return subprocess.run(list(args), capture_output=True, text=True)


def test_codec_and_text_parsing_selftest(tmp_path):
    r = run(os.path.join(BIN, "mvs_codec_selftest"), str(tmp_path) + "/")
    assert r.returncode == 0, r.stderr
    assert "ok" in r.stdout


def test_project_everything_usage_errors():
    exe = os.path.join(BIN, "project_everything")
    for args in ([], ["sketch"], ["sketch", "a"], ["frobnicate", "a", "b"], ["sketch", "a", "b", "-d"],
                 ["sketch", "a", "b", "--dimension", "x"], ["sketch", "a", "b", "c"],
                 # the README's stale spelling is rejected by the real parser too
                 ["toy", "toy_db/", "-t", "8", "-d", "2048", "-s", "0"]):
        r = run(exe, *args)
        assert r.returncode == 1, args
        assert r.stderr.startswith("Usage:\n  Convert mode:\n") and "Sketch mode:" in r.stderr
        assert r.stdout == ""


def test_pairwise_usage_errors():
    exe = os.path.join(BIN, "pairwise_comp_optimized")
    r = run(exe, "--help")
    assert r.returncode == 0 and r.stdout.startswith("Usage:")
    full = ["--db", "x/", "--max_memory_gb", "12", "--num_threads", "8", "--output_folder", "o", "--num_shards", "1",
            "--shard_idx", "0"]
    for i in range(0, len(full), 2):          # every one of the six flags is required
        r = run(exe, *(full[:i] + full[i + 2:]))
        assert r.returncode == 1 and r.stdout.startswith("Usage:"), full[i]
    r = run(exe, *(full[:3] + ["abc"] + full[4:]))
    assert r.returncode == 1
    # README spelling (--dimension/--strategy) is rejected
    r = run(exe, *full, "--dimension", "2048")
    assert r.returncode == 1


def test_pairwise_missing_db(tmp_path):
    exe = os.path.join(BIN, "pairwise_comp_optimized")
    r = run(exe, "--db", str(tmp_path) + "/nodb/", "--max_memory_gb", "12", "--num_threads", "8", "--output_folder",
            str(tmp_path) + "/out", "--num_shards", "1", "--shard_idx", "0")
    assert r.returncode == 1
    assert "Required file 'vector_norms.txt' not found" in r.stderr


def test_standalone_projection_usage():
    exe = os.path.join(BIN, "standalone_projection")
    r = run(exe)
    assert r.returncode == 1 and r.stderr.startswith("Usage: ")
    r = run(exe, "/nonexistent/file", "8")
    assert r.returncode == 1 and "Error opening file: /nonexistent/file" in r.stderr


def test_convert_matches_reference_fixture(tmp_path):
    """`project_everything convert` on the reference's toy .sig.zip set: every sample's k=31 hash set equals
    the fixture (tests/golden/toy_hashes.npz, cross-checked against the reference's own convert by
    make_golden.py).  Needs the reference's test data, which exists only in the dev container."""
    import json
    import numpy as np
    import pytest
    toy = "/root/reference/test/toy"
    if not os.path.isdir(toy):
        pytest.skip("reference test data not present on this machine")
    out = tmp_path / "toy_hashes.txt"
    r = run(os.path.join(BIN, "project_everything"), "convert", toy, str(out), "-t", "4")
    assert r.returncode == 0, r.stderr
    assert r.stdout.count("Processed ") == 61 and "Time to convert all signatures: " in r.stdout
    got = {}
    for line in out.read_text().split("\n"):
        if ":" in line:
            name, rest = line.split(":", 1)
            got[name] = np.array([int(t) for t in rest.split()], dtype=np.uint64)
    h = np.load(os.path.join(ROOT, "tests", "golden", "toy_hashes.npz"))
    names = [str(x) for x in h["names"]]
    offs = h["offsets"]
    deltas = h["deltas"].astype(np.uint64)
    assert sorted(got) == names
    for i, n in enumerate(names):
        want = np.cumsum(deltas[offs[i]:offs[i + 1]], dtype=np.uint64)
        assert np.array_equal(got[n], want), n
    with open(os.path.join(ROOT, "tests", "golden", "toy_sketch_digests.json")) as f:
        dg = json.load(f)
    assert all(len(got[n]) == dg[n]["n_hashes"] for n in names)


def test_convert_error_paths(tmp_path):
    exe = os.path.join(BIN, "project_everything")
    (tmp_path / "sigs").mkdir()
    (tmp_path / "sigs" / "broken.sig.zip").write_bytes(b"this is not a zip archive at all")
    r = run(exe, "convert", str(tmp_path / "sigs"), str(tmp_path / "out.txt"))
    assert r.returncode == 0 and "Failed to unzip" in r.stderr
    assert (tmp_path / "out.txt").read_text() == "broken:\n"       # sample kept with an empty set, like the reference


# ---- the text parser against what the REFERENCE binaries did with the same bytes (tests/golden/ref_parser.json,
# ---- written by tests/golden/make_golden_parser.py from oracle/_ref/project_everything and standalone_projection) ----
def _ref_parser_fixture():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "ref_parser.json")) as f:
        return json.load(f)


def _parse_with_tool(path, mode, simd):
    """mvs_codec_selftest --parse: (names, [sorted unique values]) as read_hash_file() returns them"""
    import numpy as np
    env = dict(os.environ)
    if not simd:
        env["MVS_HOST_NO_SIMD"] = "1"
    r = subprocess.run([os.path.join(BIN, "mvs_codec_selftest"), "--parse", str(path), mode], capture_output=True, env=env)
    assert r.returncode == 0, r.stderr
    names, sets = [], []
    for line in r.stdout.decode("ascii").split("\n")[:-1]:
        hexname, count, vals = line.split("\t")
        v = np.array([int(t) for t in vals.split()], dtype=np.uint64)
        assert len(v) == int(count) and (len(v) < 2 or bool(np.all(v[1:] > v[:-1])))
        names.append(bytes.fromhex(hexname).decode("latin-1"))
        sets.append(v)
    return names, sets


def test_text_parser_against_reference_binary_on_malformed_lines(tmp_path):
    """src/project_everything.cpp:264-281 (`while (iss >> hash)` per "name: ..." record): signs, 2^64 and beyond, digits
    glued to letters or to signs, every kind of blank, lines without / with several ':', empty names, CR LF, no final newline,
    an empty file.  Both parsers (AVX2 fast path with its scalar fallback; scalar only) must give the sets whose d = 128
    projection (oracle, itself pinned against the reference) equals the vectors.bin the reference binary wrote, under the
    names it wrote."""
    import numpy as np
    from oracle import pyoracle as orc
    fx = _ref_parser_fixture()
    d = fx["d"]
    for key, case in sorted(fx["sketch"].items()):
        p = tmp_path / ("sk_%s.txt" % key)
        p.write_bytes(case["input"].encode("latin-1"))
        want = np.array(case["vectors"], dtype=np.int32).reshape(len(case["names"]), d)
        for simd in (True, False):
            names, sets = _parse_with_tool(p, "names", simd)
            assert names == case["names"], (key, simd)
            assert case["stdout_first"].startswith("Loaded %d hash sets from " % len(names)), key
            for i, v in enumerate(sets):
                got = orc.project(v, d)
                assert np.array_equal(got, want[i]), (key, simd, names[i], v[:8])


def test_text_parser_against_reference_standalone_projection(tmp_path):
    """src/standalone_projection.cpp:28-36: every line is a set (no names, no ':' rule); the reference's stdout is the
    float rendering of the d = 128 sketch of what it parsed"""
    import numpy as np
    from oracle import pyoracle as orc
    fx = _ref_parser_fixture()
    d = fx["d"]
    for key, case in sorted(fx["standalone_projection"].items()):
        p = tmp_path / ("sp_%s.txt" % key)
        p.write_bytes(case["input"].encode("latin-1"))
        want_lines = case["stdout"].split("\n")[:-1]
        for simd in (True, False):
            _, sets = _parse_with_tool(p, "lines", simd)
            assert len(sets) == len(want_lines), (key, simd)
            for v, line in zip(sets, want_lines):
                got = " ".join("%g" % float(np.float32(x)) for x in orc.project(v, d))
                assert got == line, (key, simd)


def test_cpp_step_partition_equals_the_python_mirror():
    """csrc/host/mvs_step.hpp (the C++ host of the strong-scaled step) and parallel.py cut the work the same way: rank rows
    (src/pairwise_comp_optimized.cpp:938-940), padded blocks, the symmetric block plan, the chunks the coarse plane travels in
    and the peers' rectangles clipped to a chunk -- mvs_step_plan prints the C++ side's arithmetic"""
    import json
    from metagenome_vector_sketches_amd import parallel, _capi
    exe = os.path.join(BIN, "mvs_step_plan")
    cases = [(1, 61, 61, 2, 0.33, 1), (2, 31, 61, 2, 0.33, 1), (3, 21, 61, 3, 0.5, 1), (4, 25000, 100000, 2, 0.33, 1),
             (8, 12500, 100000, 2, 0.33, 1), (8, 12500, 100000, 4, 0.25, 0), (7, 143, 1000, 2, 0.33, 1), (8, 8, 61, 2, 0.33, 1),
             (5, 2560, 12800, 3, 0.0, 1), (6, 1000, 5001, 2, 0.33, 1), (8, 125000, 1000000, 2, 0.33, 1)]
    for world, block_rows, n_total, chunks, first, sym in cases:
        r = run(exe, str(world), str(block_rows), str(n_total), str(chunks), str(first), str(sym))
        assert r.returncode == 0, r.stderr
        got = json.loads(r.stdout)
        if block_rows == (n_total + world - 1) // world:
            assert (block_rows, got["P"]) == _capi.shard_layout(n_total, world)
        P = got["P"]
        assert got["half_split"] == parallel.half_split(P)
        want_chunks = parallel.chunk_bounds(P, chunks, first if first > 0 else None)
        assert [tuple(c) for c in got["chunks"]] == want_chunks, (world, block_rows, chunks, first)
        for rank, rec in enumerate(got["ranks"]):
            if block_rows == (n_total + world - 1) // world:
                assert tuple(rec["rows"]) == parallel.shard_rows(n_total, world, rank)
            plan = parallel.block_plan(world, rank, P, symmetric=bool(sym))
            assert [tuple(b) for b in rec["plan"]] == plan, (world, rank)
            for (c0, c1), cl in zip(want_chunks, rec["clipped"]):
                assert [tuple(b) for b in cl] == parallel.clip_blocks(plan[1:], P, c0, c1), (world, rank, c0, c1)
