#!/usr/bin/env python3
"""Regenerate the pairwise reference pins (dev container only).

Runs oracle/_ref/ref_pairwise32 / ref_pairwise16 -- the reference's OWN `load_matrix_block`,
`compute_sparse_dot_products_optimized`, `Matrix16`, `load_matrix_block_int16` and
`compute_sparse_dot_products_optimized_16`, compiled from line ranges of /root/reference/src by
`make -C oracle ref_pairwise` (recipe + driver: oracle/Makefile, oracle/ref_pairwise_driver.inc) -- on small DB
folders and records the kept cells `(i, j, P)` in the order the reference appends them.

Outputs (data only):
  tests/golden/ref_pairwise.json        cases: dtype, d, norm text lines, runs [{max_memory_gb|-, shards, digest of the cells}]
  tests/golden/ref_pairwise_inputs.npz  the sketch matrices of the cases that are not the toy DB or a formula, and
                                        every run's kept cells (int64 [kept, 3] = i, j, P in the reference's order)
                                        and "rows/<case>": the same cells through the bits-free first lines of the
                                        reference's WRITERS (`rows` mode of the drivers): (row, col, q) -- the Jaccard
                                        quantiser of src/pairwise_comp_optimized.cpp:654-672 -- for int32 DBs,
                                        (row, col, round(dot / d)) -- _16bits.cpp:260-280 -- for int16 DBs, in the writer's
                                        emission order
  tests/golden/toy_pairwise_cells*.txt  (row, col, dot) REFERENCE output; the int32 list's q column REFERENCE output too
                                        (writer head); the int16 list's q is this repository's extension (the reference's
                                        int16 writer stores round(dot / d), never a Jaccard byte)
  kat.json: provenance updated

Run:  make -C oracle ref && python tests/golden/make_golden_pairwise.py
"""
import hashlib
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")
REFBIN = os.path.join(ROOT, "oracle", "_ref")
WORK = os.path.join(REFBIN, "work_pairwise")

from oracle import pyoracle as orc  # noqa: E402  (only to assert oracle == reference while generating)

M64 = (1 << 64) - 1


def mix_array(x):
    """splitmix64 finaliser on a uint64 array (the generator of the formula cases; conftest.py has the same)"""
    x = (x + np.uint64(0x9e3779b97f4a7c15))
    x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
    x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
    return x ^ (x >> np.uint64(31))


def formula_sketches(n, d, seed, cluster, amp, shared_amp):
    """deterministic clustered int sketches: entry = noise(row, k) + shared(cluster(row), k), both uniform integers
    from a splitmix64 counter (no library RNG, so the fixture is a formula and not 8 MB of data)"""
    with np.errstate(over="ignore"):
        rows = np.arange(n, dtype=np.uint64)[:, None]
        ks = np.arange(d, dtype=np.uint64)[None, :]
        a = mix_array(np.uint64(seed) * np.uint64(1000003) + rows * np.uint64(65537) + ks)
        b = mix_array(np.uint64(seed) * np.uint64(7919) + (rows // np.uint64(cluster)) * np.uint64(2654435761) + ks
                      + np.uint64(1 << 40))
    noise = (a % np.uint64(2 * amp + 1)).astype(np.int64) - amp
    shared = (b % np.uint64(2 * shared_amp + 1)).astype(np.int64) - shared_amp
    return noise + shared


def write_db(folder, vectors, norm_lines):
    os.makedirs(folder, exist_ok=True)
    vectors.tofile(os.path.join(folder, "vectors.bin"))
    with open(os.path.join(folder, "vector_norms.txt"), "w") as f:
        f.write("".join(l + "\n" for l in norm_lines))


def run_ref(folder, elem, d, max_memory_gb, num_shards, shard_idx, threads=1, rows=False):
    """rows=False: kept cells (i, j, P) in append order; rows=True: the same cells through the bits-free head of the
    reference's writer -- (row, col, q) for int32 DBs (:654-672), (row, col, round(dot / d)) for int16 DBs
    (_16bits.cpp:260-280) -- in the order the writer would emit them (its unordered_map's iteration order)"""
    exe = os.path.join(REFBIN, "ref_pairwise32" if elem == 4 else "ref_pairwise16")
    cmd = [exe, os.path.join(folder, "vectors.bin"), os.path.join(folder, "vector_norms.txt"), str(d)]
    if elem == 4:
        cmd.append(repr(float(max_memory_gb)))
    cmd += [str(num_shards), str(shard_idx), str(threads)]
    if rows:
        cmd.append("rows")
    out = subprocess.run(cmd, check=True, capture_output=True, text=True).stdout
    return [[int(t) for t in line.split()] for line in out.split("\n") if line]


def chunk_of(elem, d, max_memory_gb):
    if elem == 4:
        return orc.chunk_size(max_memory_gb, d)
    return max((16 * 1024 * 1024) // (2 * 2 * d), 64)         # _16bits.cpp:362-369


def norm_sq(line):
    """:893-901: stod(text after the first ' ') squared (x * x: 1e200 squared is inf, not an OverflowError)"""
    x = float(line.split(" ", 1)[1])
    return x * x


def oracle_cells(vectors, norm_lines, elem, d, max_memory_gb, num_shards, shard_idx):
    n2 = np.array([norm_sq(l) for l in norm_lines])
    b, e = orc.shard_rows(len(vectors), num_shards, shard_idx)
    c = orc.pairwise_rows(vectors, n2, row_begin=b, row_end=e, chunk=chunk_of(elem, d, max_memory_gb))
    return [[int(x["row"]), int(x["col"]), int(x["dot"])] for x in c], [int(x["q"]) for x in c]


def default_norm_lines(vectors, d):
    return ["s%d %s" % (i, orc.format_norm(orc.norm(v.astype(np.int32)))) for i, v in enumerate(vectors)]


def main():
    for exe in ("ref_pairwise32", "ref_pairwise16"):
        if not os.path.exists(os.path.join(REFBIN, exe)):
            sys.exit("build the reference functions first: make -C oracle ref_pairwise")
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    db = np.load(os.path.join(GOLD, "toy_db.npz"))
    names = [str(x) for x in db["names"]]
    toy = db["vectors"]
    toy_norm_lines = [l for l in open(os.path.join(GOLD, "toy_vector_norms.txt")).read().split("\n") if l]
    rng = np.random.default_rng(20251004)
    inputs, cases = {}, {}

    def add_case(name, vectors, elem, d, norm_lines, runs, source, note):
        vectors = np.ascontiguousarray(vectors, dtype=np.int32 if elem == 4 else np.int16)
        folder = os.path.join(WORK, name)
        write_db(folder, vectors, norm_lines)
        rec = {"elem": elem, "d": d, "n": int(len(vectors)), "vectors": source, "norm_lines": norm_lines,
               "note": note, "runs": []}
        if source == "inline":
            inputs[name] = vectors
        for gb, shards, k in runs:
            cells = run_ref(folder, elem, d, gb, shards, k, threads=1)
            cells4 = run_ref(folder, elem, d, gb, shards, k, threads=4)
            assert sorted(cells4) == sorted(cells), name           # thread count changes at most the order (int16)
            want, _ = oracle_cells(vectors, norm_lines, elem, d, gb, shards, k)
            if cells != want:
                diff = set(map(tuple, cells)) ^ set(map(tuple, want))
                sys.exit("ORACLE != REFERENCE in case %s run %s: %d differing cells, e.g. %s"
                         % (name, (gb, shards, k), len(diff), sorted(diff)[:5]))
            key = "cells/%s/%d" % (name, len(rec["runs"]))
            inputs[key] = np.array(cells, dtype=np.int64).reshape(-1, 3)
            rec["runs"].append({"max_memory_gb": gb if elem == 4 else None, "chunk": int(chunk_of(elem, d, gb)),
                                "num_shards": shards, "shard_idx": k, "kept": len(cells), "cells": key,
                                "cells_sha256": hashlib.sha256(
                                    "".join("%d %d %d\n" % tuple(c) for c in cells).encode()).hexdigest()})
            print("  %-22s gb=%-5s shards=%d/%d chunk=%-6d kept %6d of %d   == oracle, same order"
                  % (name, gb, k, shards, chunk_of(elem, d, gb), len(cells), len(vectors) ** 2 // shards))
        # the writer's head on the first (whole) run: quantised Jaccard / rounded dot per cell, rows in the writer's order
        gb, shards, k = runs[0]
        assert shards == 1
        wr = run_ref(folder, elem, d, gb, shards, k, threads=1, rows=True)
        cells, qs = oracle_cells(vectors, norm_lines, elem, d, gb, shards, k)
        n2 = [norm_sq(l) for l in norm_lines]
        if elem == 4:
            want = {(r, c): q for (r, c, _), q in zip(cells, qs)}
        else:                                  # std::round: half away from zero
            want = {(r, c): int(np.sign(dot) * np.floor(abs(dot) / d + 0.5)) for (r, c, dot) in cells}
        undefined = set()
        for r, c, q in wr:
            j_nan = elem == 4 and not np.isfinite(n2[r] + n2[c])
            if want.get((r, c)) != q:
                if j_nan:                     # NaN / infinite norms: uint16(round(NaN)) is undefined behaviour in the reference
                    undefined.add((r, c))
                    continue
                sys.exit("ORACLE != REFERENCE WRITER HEAD in case %s: cell (%d, %d) reference %d oracle %s"
                         % (name, r, c, q, want.get((r, c))))
        assert len(wr) == len(cells) and {(r, c) for r, c, _ in wr} == set(want)
        by_row = {}
        for r, c, q in wr:
            by_row.setdefault(r, []).append(c)
        assert all(v == sorted(v) for v in by_row.values())          # columns ascending within a row (:720 asserts it)
        inputs["rows/%s" % name] = np.array(wr, dtype=np.int64).reshape(-1, 3)
        rec["writer_head"] = {"cells": "rows/%s" % name, "what": "(row, col, q)" if elem == 4 else "(row, col, round(dot / d))",
                              "row_order": list(dict.fromkeys(r for r, _, _ in wr)),
                              "undefined_in_reference": sorted(undefined)}
        print("  %-22s writer head: %d cells, %d rows in the writer's order, == oracle%s"
              % (name, len(wr), len(by_row), " (%d cells with non-finite norms excluded)" % len(undefined) if undefined else ""))
        cases[name] = rec

    # 1. the reference's toy set (DB written by the reference's own `sketch`)
    add_case("toy_int32", toy, 4, 2048, toy_norm_lines,
             [(12, 1, 0), (1, 1, 0), (12, 2, 0), (12, 2, 1), (2.5, 3, 2)], "toy_db",
             "reference-built toy DB; 12 GB = the README's setting (chunk 192), 1 GB = chunk 16 (many tiles)")
    toy16 = np.clip(toy, -32768, 32767).astype(np.int16)        # == the reference's --int16 vectors.bin (make_golden.py asserts it)
    add_case("toy_int16", toy16, 2, 2048, toy_norm_lines, [(None, 1, 0), (None, 2, 1)], "toy_db_int16",
             "reference-built toy DB saturated to int16 (floating keep test)")

    # 2. cells sitting on / next to the keep threshold (tests/test_pairwise_gpu.py::test_keep_threshold_edges + more):
    #    d = 64, dots between -3 d and 3 d, norms whose squares make 0.05 (n2_i + n2_j) land on and next to integers
    d = 64
    e = np.zeros((28, d), dtype=np.int64)
    e[0, :] = 1
    e[1, :63] = 1
    e[2, :] = -1
    e[3, 0] = 1
    for r in range(4, 28):
        e[r] = rng.integers(-2, 3, d)
    edge_norms = ["1", "0.707107", "1", "0", "4", "2", "4.47214", "3.16228", "0.5", "1e-3", "6.32456", "2.23607",
                  "1.41421", "3", "5.47723", "4.47213", "4.47215", "1e-30", "7", "2", "4", "4", "2", "0.1",
                  "10", "0", "3.87298", "5"]
    edge_lines = ["e%d %s" % (i, t) for i, t in enumerate(edge_norms)]
    add_case("edges_int32", e, 4, d, edge_lines, [(12, 1, 0), (0.001, 1, 0)], "inline",
             "threshold edges, truncating division (:140-141); 0.001 GB = chunk 16")
    add_case("edges_int16", e, 2, d, edge_lines, [(None, 1, 0)], "inline", "threshold edges, floating division (:218)")

    # 3. products that wrap mod 2^32 (:135 int32 GEMM) and the sign games the truncating division plays on them
    w = rng.integers(-(2 ** 31), 2 ** 31, (20, d)).astype(np.int64)
    w[0] = 2 ** 31 - 1
    w[1] = -(2 ** 31)
    w[2, :] = 65536
    w[3, :] = 46341
    w_lines = ["w%d %s" % (i, t) for i, t in enumerate(["0", "1000", "3", "46341", "1e4"] * 4)]
    add_case("wrap_int32", w, 4, d, w_lines, [(12, 1, 0)], "inline", "four-limb values: products wrap mod 2^32")
    w16 = rng.integers(-32768, 32768, (20, d)).astype(np.int64)
    w16[0] = -32768
    w16[1] = 32767
    w16[2, ::2] = -32768
    w16_lines = ["w%d %s" % (i, t) for i, t in enumerate(["0", "1000", "30000", "4000", "1e4"] * 4)]
    add_case("wrap_int16", w16, 2, d, w16_lines, [(None, 1, 0)], "inline",
             "int16 extremes: madd pair sums of 2^31 and an int32 accumulator that wraps (:144-208)")

    # 4. norms that are not numbers / do not belong to the vectors
    nn = rng.integers(-40, 41, (12, d)).astype(np.int64)
    nn_lines = ["n%d %s" % (i, t) for i, t in
                enumerate(["nan", "inf", "-inf", "-3", "0", "1e200", "1e-200", "5", "-nan", "2", "20", "0.0"])]
    add_case("odd_norms_int32", nn, 4, d, nn_lines, [(12, 1, 0)], "inline", "NaN / infinite / negative / huge norms")
    add_case("odd_norms_int16", nn, 2, d, nn_lines, [(None, 1, 0)], "inline", "NaN / infinite / negative / huge norms")

    # 5. a dimension that is no multiple of 64 / 32 / 16 (scalar tails of :173-208; -ffast-math and x / d)
    d100 = 100
    v100 = np.zeros((60, d100), dtype=np.int64)
    base = rng.integers(-6, 7, (6, d100))
    for r in range(60):
        v100[r] = base[r % 6] * rng.integers(0, 2, d100) + rng.integers(-2, 3, d100)
    l100 = default_norm_lines(v100, d100)
    add_case("d100_int32", v100, 4, d100, l100, [(12, 1, 0), (0.0009, 2, 1)], "inline", "d = 100, chunk 80530 / 6")
    add_case("d100_int16", v100, 2, d100, l100, [(None, 1, 0)], "inline", "d = 100: AVX2 body 96 + scalar tail 4")

    # 6. clustered sketches with the magnitudes of 3000-hash samples, several tiles per side
    from metagenome_vector_sketches_amd import synth
    cl = synth.make_sketches_numpy(300, 256, 3000, seed=4, cluster=8)
    lcl = default_norm_lines(cl, 256)
    add_case("clustered_int32", cl, 4, 256, lcl, [(12, 1, 0), (0.001, 1, 0), (0.002, 4, 3)], "inline",
             "300 x 256 clustered; 0.001 GB = chunk 1 ... the loop order of :949-982 cell by cell")
    add_case("clustered_int16", cl, 2, 256, lcl, [(None, 1, 0)], "inline", "300 x 256 clustered, int16 path")

    # 7. the int16 path's own tiling (chunk = 2048 columns at d = 2048): 2100 rows from a formula
    big = formula_sketches(2100, 2048, seed=77, cluster=12, amp=60, shared_amp=45)
    lbig = default_norm_lines(big, 2048)
    add_case("formula_int16_2100", big, 2, 2048, lbig, [(None, 1, 0), (None, 3, 2)],
             {"formula": "formula_sketches", "n": 2100, "d": 2048, "seed": 77, "cluster": 12, "amp": 60,
              "shared_amp": 45},
             "2100 x 2048: two column tiles of the int16 driver (:390-416)")

    # ---- toy cell lists: (row, col, dot) from the reference run, q from the oracle -------------------------------
    n2 = np.array([norm_sq(l) for l in toy_norm_lines])
    for elem, fn, case, head in ((4, "toy_pairwise_cells.txt", "toy_int32",
                                  "# row_name col_name dot q   (int32 path, chunk 192, 1 shard; row/col/dot = the "
                                  "reference's own compute_sparse_dot_products_optimized via oracle/_ref/ref_pairwise32; "
                                  "q = the reference's own quantiser lines :654-672 via the same binary's rows mode)\n"),
                                 (2, "toy_pairwise_cells_int16.txt", "toy_int16",
                                  "# row_name col_name dot q   (int16 path: floating keep test; row/col/dot = the "
                                  "reference's own compute_sparse_dot_products_optimized_16 via oracle/_ref/ref_pairwise16; "
                                  "q = oracle: the Jaccard byte for an int16 DB is this repository's extension)\n")):
        cells = inputs[cases[case]["runs"][0]["cells"]].tolist()
        vec = toy if elem == 4 else toy16
        _, q = oracle_cells(vec, toy_norm_lines, elem, 2048, 12, 1, 0)
        old = open(os.path.join(GOLD, fn)).read().split("\n")[1:]
        new = ["%s %s %d %d" % (names[r], names[c], dot, qq) for (r, c, dot), qq in zip(cells, q)]
        assert [l for l in old if l] == new, "toy cell list changed"      # the oracle's list WAS right: now it is pinned
        if elem == 4:                                                      # q column == the reference's quantiser lines
            wq = {(r, c): qq for r, c, qq in inputs["rows/toy_int32"].tolist()}
            assert [wq[(r, c)] for r, c, _ in cells] == [int(x) for x in q]
        with open(os.path.join(GOLD, fn), "w") as f:
            f.write(head + "".join(l + "\n" for l in new))
    assert cases["toy_int32"]["runs"][0]["kept"] == 1291 and cases["toy_int16"]["runs"][0]["kept"] == 1293

    with open(os.path.join(GOLD, "ref_pairwise.json"), "w") as f:
        json.dump({"provenance": "oracle/_ref/ref_pairwise32 + ref_pairwise16: the reference's own functions "
                                 "(src/pairwise_comp_optimized.cpp:33-160, src/pairwise_comp_optimized_16bits.cpp:40-244) "
                                 "compiled from line ranges by `make -C oracle ref_pairwise`; driver loop restated from "
                                 ":893-982 / _16bits.cpp:343-416 (oracle/ref_pairwise_driver.inc); writer_head = the same "
                                 "cells through the bits-free first lines of the writers (:654-672 Jaccard quantiser; "
                                 "_16bits.cpp:260-264 + :274-280 grouping and round(dot / d)), included as reference text "
                                 "by the same recipe; generated by "
                                 "tests/golden/make_golden_pairwise.py",
                   "cases": cases}, f, separators=(",", ":"))
    np.savez_compressed(os.path.join(GOLD, "ref_pairwise_inputs.npz"), **inputs)
    with open(os.path.join(GOLD, "kat.json")) as f:
        kat = json.load(f)
    prov = {k: v for k, v in kat["provenance"].items() if not k.startswith("oracle-generated")
            and not k.startswith("reference functions")}      # (both "reference functions ..." keys are rewritten below)
    prov["reference functions compiled from line ranges (oracle/_ref/ref_pairwise32, ref_pairwise16: load_matrix_block, "
         "compute_sparse_dot_products_optimized, Matrix16, load_matrix_block_int16, "
         "compute_sparse_dot_products_optimized_16)"] = [
        "ref_pairwise.json (every case: kept cells (i, j, P) in the reference's order)",
        "toy_pairwise_cells.txt, toy_pairwise_cells_int16.txt: columns row, col, dot"]
    prov["reference functions + the bits-free first lines of the reference's writers (src/pairwise_comp_optimized.cpp:654-672 "
         "Jaccard quantiser; _16bits.cpp:260-264, :274-280 round(dot / d)), same binaries in rows mode"] = [
        "ref_pairwise.json writer_head / ref_pairwise_inputs.npz rows/<case> (every case)",
        "toy_pairwise_cells.txt: column q"]
    prov["oracle-generated (oracle/mvs_oracle.c; NOT reference output -- the reference's int16 writer stores round(dot / d), "
         "a Jaccard byte for an int16 DB is this repository's extension)"] = [
        "toy_pairwise_cells_int16.txt: column q only"]
    kat["provenance"] = prov
    with open(os.path.join(GOLD, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    shutil.rmtree(WORK, ignore_errors=True)
    for fn in ("ref_pairwise.json", "ref_pairwise_inputs.npz"):
        print("  %-28s %8d B" % (fn, os.path.getsize(os.path.join(GOLD, fn))))


if __name__ == "__main__":
    main()
