#!/usr/bin/env python3
"""Regenerate tests/golden/* from the reference (dev container only).

Inputs : /root/reference/test/toy/*.sig.zip (the reference's own test data) and the reference
         binaries built by `make -C oracle ref` into oracle/_ref/ (project_everything,
         standalone_projection -- compiled from /root/reference/src, nothing copied).
Outputs: small data fixtures (inputs + expected outputs) committed under tests/golden/.
         The pairwise half of the reference is unbuildable here (absent `bits` submodule), so
         toy_pairwise_cells.txt is produced by oracle/mvs_oracle.c on the *reference-built* toy DB
         and is pinned by the values SURVEY.md section 4 recorded (checked below and in the tests).

Run:  python tests/golden/make_golden.py
"""
import ctypes
import glob
import gzip
import hashlib
import json
import os
import shutil
import subprocess
import sys
import zipfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
REF = "/root/reference"
REFBIN = os.path.join(ROOT, "oracle", "_ref")
WORK = os.path.join(REFBIN, "work")


def ingest_toy():
    """Python restatement of project_everything.cpp:94-176 (ksize==31 only, name = file name up to
    the first '.'), used to key everything by sample name."""
    out = {}
    for path in sorted(glob.glob(os.path.join(REF, "test", "toy", "*.sig.zip"))):
        name = os.path.basename(path).split(".")[0]
        hashes = set()
        with zipfile.ZipFile(path) as z:
            for member in z.namelist():
                if not (member.startswith("signatures/") and member.endswith(".gz")):
                    continue
                for rec in json.loads(gzip.decompress(z.read(member))):
                    for sig in rec["signatures"]:
                        if sig["ksize"] == 31:
                            hashes.update(int(h) for h in sig["mins"])
        out[name] = np.array(sorted(hashes), dtype=np.uint64)
    return out


def run(cmd, cwd=None):
    return subprocess.run(cmd, cwd=cwd, check=True, capture_output=True, text=True).stdout


def parse_hash_file(path):
    res = {}
    with open(path) as f:
        for line in f:
            if ":" not in line:
                continue
            name, rest = line.split(":", 1)
            res[name] = np.array(sorted(set(int(t) for t in rest.split())), dtype=np.uint64)
    return res


def write_hash_file(path, names, table):
    with open(path, "w") as f:
        for n in names:
            f.write(n + ":" + "".join(" %d" % int(h) for h in table[n]) + "\n")


def read_db(folder, n, d, dtype=np.int32):
    v = np.fromfile(os.path.join(folder, "vectors.bin"), dtype=dtype).reshape(n, d)
    with open(os.path.join(folder, "vector_norms.txt")) as f:
        norms = f.read()
    return v, norms


def main():
    if not os.path.exists(os.path.join(REFBIN, "project_everything")):
        sys.exit("build the reference first: make -C oracle ref")
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    os.makedirs(GOLD, exist_ok=True)

    toy = ingest_toy()
    names = sorted(toy)
    print("toy samples:", len(names), "total hashes:", sum(len(v) for v in toy.values()))

    # 1. cross-check the Python ingest against the reference's `convert`
    run([os.path.join(REFBIN, "project_everything"), "convert", os.path.join(REF, "test", "toy"),
         os.path.join(WORK, "toy_hashes_ref.txt"), "-t", "4"], cwd=WORK)
    ref_tab = parse_hash_file(os.path.join(WORK, "toy_hashes_ref.txt"))
    assert sorted(ref_tab) == names
    for n in names:
        assert np.array_equal(ref_tab[n], toy[n]), n
    print("python ingest == reference convert for all samples")

    # 2. reference sketch on the name-sorted hash file (sample order == sorted names)
    hf = os.path.join(WORK, "toy_hashes.txt")
    write_hash_file(hf, names, toy)
    d = 2048
    out32 = run([os.path.join(REFBIN, "project_everything"), "sketch", hf,
                 os.path.join(WORK, "toy_db"), "-d", str(d)], cwd=WORK)
    run([os.path.join(REFBIN, "project_everything"), "sketch", hf,
         os.path.join(WORK, "toy_db16"), "-d", str(d), "--int16"], cwd=WORK)
    vec, norms_txt = read_db(os.path.join(WORK, "toy_db"), len(names), d)
    vec16, norms16_txt = read_db(os.path.join(WORK, "toy_db16"), len(names), d, np.int16)
    assert norms_txt == norms16_txt
    assert np.array_equal(vec16.astype(np.int32), np.clip(vec, -32768, 32767))

    # all toy hashes, delta-coded so the file stays small
    flat = np.concatenate([toy[n] for n in names])
    offs = np.zeros(len(names) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(toy[n]) for n in names])
    deltas = flat.copy()
    for i in range(len(names)):
        seg = flat[offs[i]:offs[i + 1]]
        if len(seg):
            deltas[offs[i] + 1:offs[i + 1]] = seg[1:] - seg[:-1]
    np.savez_compressed(os.path.join(GOLD, "toy_hashes.npz"), names=np.array(names),
                        offsets=offs, deltas=deltas)
    np.savez_compressed(os.path.join(GOLD, "toy_db.npz"), names=np.array(names), vectors=vec)
    with open(os.path.join(GOLD, "toy_vector_norms.txt"), "w") as f:
        f.write(norms_txt)
    digests = {}
    for i, n in enumerate(names):
        row = vec[i]
        digests[n] = {"n_hashes": int(len(toy[n])), "sha256": hashlib.sha256(row.tobytes()).hexdigest(),
                      "sum": int(row.sum()), "sumsq": int((row.astype(np.int64) ** 2).sum()),
                      "first8": [int(x) for x in row[:8]]}
    # SURVEY.md section 4 known answers
    assert digests["DRR000824"]["first8"] == [-3, -1, 1, -5, 1, 1, 5, -5]
    assert (digests["DRR000824"]["sum"], digests["DRR000824"]["sumsq"]) == (-288, 10200)
    assert digests["DRR000980"]["first8"] == [164, 204, 306, 138, 466, -340, -30, -138]
    assert (digests["DRR000980"]["sum"], digests["DRR000980"]["sumsq"]) == (-3318, 166140884)
    sorted_sha = hashlib.sha256(vec.tobytes()).hexdigest()
    assert sorted_sha == "93fb4358e2774ad31cd50fe8f00a1eb6a97aa0ad029f166654bbf3385df74cb8", sorted_sha
    print("SURVEY section-4 projection goldens reproduced")

    # 3. stdout protocol of `sketch` (first / last lines; the per-sample lines are unordered under OpenMP)
    lines = out32.strip().split("\n")
    kat = {"splitmix64_of_1": "0x910a2dec89025cc1",
           "toy_vectors_sha256_sorted_by_name": sorted_sha,
           "sketch_stdout_first": lines[0],
           "sketch_stdout_projected_example": sorted(l for l in lines if l.startswith("Projected"))[0],
           "sketch_stdout_last_prefix": lines[-1].split(":")[0]}

    # 4. standalone_projection known answers / edge cases (text in, text out)
    cases = {
        "kat_123_d8": ("1 2 3\n", 8),
        "empty_line_d64": ("\n", 64),
        "one_hash_d100": ("42\n", 100),
        "dups_d64": ("7 7 7 9\n", 64),
        "wrap_u64_d128": ("18446744073709551615 18446744073709551552 5\n", 128),
        "two_lines_d70": ("1 2 3\n4 5 6 7\n", 70),
        "big_values_d4096": (" ".join(str(int(h)) for h in toy["DRR000824"]) + "\n", 4096),
    }
    sp = {}
    for key, (text, dim) in cases.items():
        p = os.path.join(WORK, key + ".txt")
        with open(p, "w") as f:
            f.write(text)
        sp[key] = {"input": text, "d": dim,
                   "stdout": run([os.path.join(REFBIN, "standalone_projection"), p, str(dim)])}
    assert sp["kat_123_d8"]["stdout"] == "-1 1 -1 -1 3 1 -3 -3\n"
    kat["standalone_projection"] = sp

    # 3b. BASELINE config 4 dimension (d = 4096) and a non-multiple-of-64 dimension on the toy set: digests only
    for dim in (4096, 100):
        run([os.path.join(REFBIN, "project_everything"), "sketch", hf, os.path.join(WORK, "toy_db_d%d" % dim), "-d",
             str(dim)], cwd=WORK)
        vd, nd = read_db(os.path.join(WORK, "toy_db_d%d" % dim), len(names), dim)
        kat["toy_vectors_sha256_d%d" % dim] = hashlib.sha256(vd.tobytes()).hexdigest()
        kat["toy_row_sha256_d%d" % dim] = {n: hashlib.sha256(vd[i].tobytes()).hexdigest() for i, n in enumerate(names)}
        kat["toy_norms_d%d" % dim] = nd

    # a large-magnitude case for the float text formatting (|v| >= 1e6 prints in %g exponent form):
    # 1.2M hashes in ONE line is too slow/large for a fixture; covered by unit tests of the formatter.

    with open(os.path.join(GOLD, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    with open(os.path.join(GOLD, "toy_sketch_digests.json"), "w") as f:
        json.dump(digests, f, indent=1, sort_keys=True)

    # 5. pairwise cells: OUR oracle on the REFERENCE-built DB, pinned by SURVEY section 4
    lib = ctypes.CDLL(os.path.join(ROOT, "oracle", "libmvs_oracle.so"))

    class Cell(ctypes.Structure):
        _fields_ = [("row", ctypes.c_int32), ("col", ctypes.c_int32), ("dot", ctypes.c_int32),
                    ("q", ctypes.c_int32)]
    lib.mvs_oracle_pairwise_rows.restype = ctypes.c_int64
    lib.mvs_oracle_pairwise_rows.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int64, ctypes.c_int64,
                                             ctypes.c_void_p, ctypes.c_int64, ctypes.c_int]
    lib.mvs_oracle_norm_sq_from_text.restype = ctypes.c_double
    lib.mvs_oracle_norm_sq_from_text.argtypes = [ctypes.c_char_p]
    n2 = np.array([lib.mvs_oracle_norm_sq_from_text(l.split(" ", 1)[1].encode())
                   for l in norms_txt.strip().split("\n")], dtype=np.float64)
    N = len(names)

    def cells_for(arr, elem):
        buf = (Cell * (N * N))()
        a = np.ascontiguousarray(arr)
        cnt = lib.mvs_oracle_pairwise_rows(a.ctypes.data, elem, N, d, n2.ctypes.data, 0, N, 192,
                                           ctypes.addressof(buf), N * N, 4)
        return [(c.row, c.col, c.dot, c.q) for c in buf[:cnt]]

    c32 = cells_for(vec, 4)
    c16 = cells_for(vec16, 2)
    print("toy kept cells int32 path:", len(c32), " int16 path:", len(c16))
    assert len(c32) == 1291 and len(c16) == 1293, "SURVEY section-4 kept-cell counts not reproduced"
    # SURVEY section 4 also records rows 20 and 6 (indices in the survey container's readdir order ==
    # the order `convert` wrote); re-key them by sample name and pin them.
    ref_order = [l.split(":")[0] for l in open(os.path.join(WORK, "toy_hashes_ref.txt")) if ":" in l]
    pos = {n: i for i, n in enumerate(ref_order)}
    by_row = {}
    for r, c, dot, q in c32:
        by_row.setdefault(pos[names[r]], []).append((pos[names[c]], q))
    survey = {20: ([20, 22], [255, 20]),
              6: ([6, 10, 16, 18, 21, 29, 31, 34, 38, 44, 46, 54],
                  [255, 103, 64, 144, 32, 93, 116, 128, 75, 162, 153, 111])}
    pins = {}
    for r, (cols, qs) in survey.items():
        got = sorted(by_row[r])
        assert [c for c, _ in got][:len(cols)] == cols and [q for _, q in got][:len(qs)] == qs, r
        pins[ref_order[r]] = {"cols": [ref_order[c] for c in cols], "q": qs}
    print("SURVEY section-4 pairwise rows reproduced")
    kat["survey_pairwise_pins"] = pins
    kat["survey_kept_cells"] = {"int32": 1291, "int16": 1293, "total_cells": N * N}
    with open(os.path.join(GOLD, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    with open(os.path.join(GOLD, "toy_pairwise_cells.txt"), "w") as f:
        f.write("# row_name col_name dot q   (int32 path, chunk 192, 1 shard; oracle on reference-built toy DB)\n")
        for r, c, dot, q in c32:
            f.write("%s %s %d %d\n" % (names[r], names[c], dot, q))
    with open(os.path.join(GOLD, "toy_pairwise_cells_int16.txt"), "w") as f:
        f.write("# row_name col_name dot q   (int16 path: floating keep test)\n")
        for r, c, dot, q in c16:
            f.write("%s %s %d %d\n" % (names[r], names[c], dot, q))
    print("wrote fixtures to", GOLD)
    for fn in sorted(os.listdir(GOLD)):
        print("  %-36s %8d B" % (fn, os.path.getsize(os.path.join(GOLD, fn))))


if __name__ == "__main__":
    main()
