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
