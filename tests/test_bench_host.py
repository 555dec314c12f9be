"""CPU: host-side pieces of bench.py.  fast_norm_sq stands in for the text round trip of the norms
(sketch() writes "%g" of sqrt(sumsq/d), the pairwise stage parses it and squares it,
src/pairwise_comp_optimized.cpp:893-901): it must equal the oracle's printf/strtod path bit for bit."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from bench import fast_norm_sq   # noqa: E402
from oracle import pyoracle as orc   # noqa: E402


def test_fast_norm_sq_equals_text_round_trip():
    rng = np.random.default_rng(7)
    for d in (2048, 100, 4096, 1):
        ss = np.concatenate([
            rng.integers(0, 2 ** 40, 20000),
            rng.integers(0, 3000, 500),
            (10 ** rng.uniform(0, 18, 10000)).astype(np.int64),
            np.array([0, 1, d, 100 * d, 10000 * d, d * 10 ** 10, 999999 ** 2 * d, 2 ** 62,
                      d * 999999, d * 9999995 ** 2 // 100, d * 12345650 ** 2 // 10000]),      # near ties / digit bumps
        ]).astype(np.int64)
        got = fast_norm_sq(ss, d)
        want = np.array([orc.norm_sq_from_text(orc.format_norm(float(np.sqrt(v / d)))) for v in ss])
        assert np.array_equal(got, want), d


# ---- `python bench.py --gpus N` without a launcher: the parent starts N fresh workers itself (launch_workers) ----
_STUB = r'''
import json, os, sys, time
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
assert os.environ["LOCAL_RANK"] == os.environ["RANK"] and os.environ["MASTER_ADDR"] == "127.0.0.1"
assert int(os.environ["MASTER_PORT"]) > 0
mode = sys.argv[1]
if mode == "fail" and rank == 1:
    sys.exit(7)
if mode == "fail" and rank == 0:
    time.sleep(60)          # a rank left inside a collective by the one that died
if mode == "hang":
    time.sleep(60)          # every rank inside a collective that never completes
if rank == 0:
    print("[Gloo] Rank 0 is connected to 2 peer ranks")      # a library that announces itself on stdout
print("noise from rank %d" % rank if rank else json.dumps({"world": world, "argv": sys.argv[1:]}))
'''


def test_launcher_relays_rank0_line_and_worst_exit_code(tmp_path, capfd):
    import json
    import time
    from bench import launch_workers
    stub = tmp_path / "stub.py"
    stub.write_text(_STUB)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    rc = launch_workers(3, ["ok", "--steps", "2"], script=str(stub), env=env)
    out, err = capfd.readouterr()
    assert rc == 0
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0]) == {"world": 3, "argv": ["ok", "--steps", "2"]}
    assert "noise from rank 1" in err and "noise from rank 2" in err      # other ranks never write to the bench line's stream
    assert "[Gloo] Rank 0" in err
    t0 = time.monotonic()
    rc = launch_workers(2, ["fail"], script=str(stub), env=env, grace_s=0.5)
    out, err = capfd.readouterr()
    assert rc != 0 and time.monotonic() - t0 < 30                         # rank 0 was not waited for for a minute
    assert "worker exit codes" in err and not out.strip()
    t0 = time.monotonic()
    rc = launch_workers(2, ["hang"], script=str(stub), env=env, grace_s=0.2, deadline_s=1.0)
    out, err = capfd.readouterr()
    assert rc != 0 and time.monotonic() - t0 < 30 and "still running after" in err and not out.strip()


def test_bench_parent_does_not_import_torch_before_launching(tmp_path):
    """the process that becomes the launcher must not initialise HIP: bench.py decides before `import torch`"""
    import ast
    with open(os.path.join(ROOT, "bench.py")) as f:
        tree = ast.parse(f.read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    launch_line = next(n.lineno for n in ast.walk(main) if isinstance(n, ast.Call) and getattr(n.func, "id", "") == "launch_workers")
    torch_line = min(n.lineno for n in ast.walk(main) if isinstance(n, ast.Import) and any(a.name.startswith("torch") for a in n.names))
    assert launch_line < torch_line
    top = [a.name for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) for a in n.names]
    assert not any(x.startswith("torch") or x.startswith("metagenome") for x in top)


def test_search_front_end_takes_the_references_command_lines():
    """src/jaccard.py:334-346: `index <folder> [-t N]`, `search <index_folder> <query_file> [-j J] [-t N]`, `-v` -- the two
    command lines of the reference's README run unchanged against metagenome_vector_sketches_amd.search (parser only: no GPU)"""
    from metagenome_vector_sketches_amd import search
    p = search.build_parser()
    a = p.parse_args("index toy_db -t 8".split())                  # README: python3 ../src/jaccard.py index toy_db -t 8
    assert (a.command, a.output_index, a.threads, a.version) == ("index", "toy_db", 8, False)
    a = p.parse_args("search toy_db queries.txt -j 0.2 -t 4".split())
    assert (a.command, a.index_folder, a.query_file, a.j, a.threads) == ("search", "toy_db", "queries.txt", 0.2, 4)
    assert p.parse_args("search db q".split()).j == 0.1            # the reference's default
    assert p.parse_args("-v index x".split()).version is True
    import pytest
    for bad in ([], ["frobnicate"], ["search", "only_one"], ["index"]):
        with pytest.raises(SystemExit):
            p.parse_args(bad)


def test_search_index_subcommand_checks_the_folder(tmp_path, capsys):
    """`index` builds nothing (the search runs on vectors.bin) but validates the folder and prints the reference's line"""
    import numpy as np
    from metagenome_vector_sketches_amd import search
    d = tmp_path / "db"
    d.mkdir()
    np.arange(3 * 8, dtype="<i4").tofile(d / "vectors.bin")
    (d / "dimension.txt").write_text("8\n")
    (d / "vector_norms.txt").write_text("a 1\nb 2\nc 3\n")
    (d / "dtype.txt").write_text("int32\n")
    assert search.main(["index", str(d), "-t", "8"]) == 0
    out = capsys.readouterr().out
    assert "Indexed 3 vectors of dimension 8 into " in out and out.startswith("Version: ")
    assert sorted(p.name for p in d.iterdir()) == ["dimension.txt", "dtype.txt", "vector_norms.txt", "vectors.bin"]   # nothing deleted
    (d / "vector_norms.txt").write_text("a 1\nb 2\n")
    import pytest
    with pytest.raises(ValueError):
        search.main(["index", str(d)])
    assert search.main(["-v", "index", str(d)]) == 0 and capsys.readouterr().out.startswith("Version: ")
