#!/usr/bin/env python3
"""Regenerate tests/golden/ref_parser.json (dev container only): what the REFERENCE binaries make of malformed hash text.

The text parser of this repository (csrc/host/mvs_host.hpp: parse_u64_raw_scalar and its AVX2 fast path) restates
`while (iss >> hash)` of src/project_everything.cpp:264-281 and src/standalone_projection.cpp:28-36.  The reference
never prints the sets it parsed, so each case records what it DOES print / write: `sketch` -> "Loaded <N> hash sets",
the names and norms of vector_norms.txt and vectors.bin at d = 128; `standalone_projection` -> its stdout.  A sketch at
d = 128 is a 128-dimensional +-1 projection of the set: two different small sets giving the same vector is not a thing
that happens, so equal vectors pin the parsed sets.

Inputs : oracle/_ref/project_everything, oracle/_ref/standalone_projection (make -C oracle ref: compiled from
         /root/reference/src where it lies, nothing copied).
Output : tests/golden/ref_parser.json -- inputs (latin-1 text of the raw bytes) and expected outputs only.

Run:  python tests/golden/make_golden_parser.py
"""
import json
import os
import shutil
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
GOLD = os.path.join(ROOT, "tests", "golden")
REFBIN = os.path.join(ROOT, "oracle", "_ref")
WORK = os.path.join(REFBIN, "work_parser")
D = 128

# name -> raw bytes of a hash file for `sketch` (records "name: h h h")
SKETCH_CASES = {
    "plain": b"a: 1 2 3\nb: 4 5\n",
    "negative_token": b"neg: 5 -3 7\n",
    "minus_zero_and_minus_max": b"m0: -0 9\nm1: -18446744073709551615 9\nm2: 3 -18446744073709551616 9\n",
    "plus_sign": b"plus: +5 6\n",
    "double_sign": b"ds: 1 +-5 6\nds2: 1 -+5 6\nds3: 1 - 5 6\n",
    "u64_max": b"big: 18446744073709551615 1\n",
    "two_pow_64": b"ovf: 1 18446744073709551616 2\n",
    "two_pow_64_plus_1": b"ovf1: 1 18446744073709551617 2\n",
    "twenty_one_digits": b"d21: 1 184467440737095516150 2\n",
    "leading_zeros_30_digits": b"lz: 000000000000000000000000000005 6\n",
    "digits_then_letters": b"mix: 12abc 7\n",
    "glued_signs": b"g1: 12-3 4\ng2: 1+2+3\ng3: 1- 2\n",
    "hex_dot_exp": b"hex: 0x10 5\ndot: 1.5 2\nexp: 1e5 2\n",
    "tab_cr_vt_ff": b"ws:\t1\r2\v3\f4 \n",
    "no_colon_line": b"nocolon 1 2 3\nyes: 1 2 3\n",
    "two_colons": b"x:y: 1 2\ntwo: 1 2 : 3\n",
    "empty_name": b": 4 5\n",
    "name_with_blanks": b"my name : 1 2\n",
    "trailing_blanks": b"t: 1 2   \n",
    "crlf": b"crlf: 1 2\r\nnext: 3\r\n",
    "no_final_newline": b"a: 1 2\nlast: 3 4",
    "empty_file": b"",
    "only_newlines": b"\n\n\n",
    "blank_and_colon_only": b"\n:\nz:\n",
    "duplicates": b"dups: 7 7 7 9 9\n",
    "high_bytes": b"hb: 1 \xff 2\nhb2: 1 2\xe9\n",
    "nul_byte": b"nul: 1 \x002 3\n",
    "long_line_marks": b"L: " + b" ".join(str(1000003 * (i + 1)).encode() for i in range(40)) + b"\nM: " +
                       b" ".join(str(18446744073709551615 - i).encode() for i in range(13)) + b" x 5\n",
    "unsorted": b"u: 9 3 7 1 18446744073709551615 0\n",
}

# name -> raw bytes for `standalone_projection` (every line is a set)
STANDALONE_CASES = {
    "plain": b"1 2 3\n4 5\n",
    "negative_and_plus": b"5 -3 +7\n",
    "glued_signs": b"12-3 4\n1+2+3\n",
    "overflow": b"1 18446744073709551616 2\n",
    "letters": b"12abc 7\n",
    "whitespace_kinds": b"\t1\r2\v3\f4 \n",
    "colon_in_line": b"a: 1 2\n",
    "crlf_and_no_final_newline": b"1 2\r\n3 4",
    "empty_file": b"",
    "blank_lines": b"\n\n1\n\n",
}


def run(cmd, cwd=None):
    return subprocess.run(cmd, cwd=cwd, check=True, capture_output=True)


def main():
    pe, sp = os.path.join(REFBIN, "project_everything"), os.path.join(REFBIN, "standalone_projection")
    if not (os.path.exists(pe) and os.path.exists(sp)):
        sys.exit("build the reference first: make -C oracle ref")
    shutil.rmtree(WORK, ignore_errors=True)
    os.makedirs(WORK)
    out = {"d": D, "provenance": "oracle/_ref/project_everything sketch <case> <db> -d %d and oracle/_ref/standalone_projection "
                                 "<case> %d, compiled from /root/reference/src (tests/golden/make_golden_parser.py)" % (D, D),
           "sketch": {}, "standalone_projection": {}}
    for key, raw in SKETCH_CASES.items():
        p = os.path.join(WORK, "sk_" + key + ".txt")
        with open(p, "wb") as f:
            f.write(raw)
        db = os.path.join(WORK, "db_" + key)
        r = run([pe, "sketch", p, db, "-d", str(D)], cwd=WORK)
        first = r.stdout.decode("latin-1").split("\n")[0]
        with open(os.path.join(db, "vector_norms.txt"), "rb") as f:
            norm_lines = f.read().decode("latin-1").split("\n")[:-1]
        # "<name> <norm>": the name may hold blanks, the norm never does
        names = [l.rsplit(" ", 1)[0] for l in norm_lines]
        norms = [l.rsplit(" ", 1)[1] for l in norm_lines]
        vec = np.fromfile(os.path.join(db, "vectors.bin"), dtype=np.int32)
        assert vec.size == len(names) * D, (key, vec.size, len(names))
        # the number of hashes the reference parsed per record: sum of squares / d ~ n is not exact, but the entry parity is:
        # every entry of a +-1 projection of n values has the parity of n, and |entry| <= n
        out["sketch"][key] = {"input": raw.decode("latin-1"), "stdout_first": first, "names": names, "norms": norms,
                              "vectors": vec.reshape(len(names), D).tolist()}
        print("sketch %-28s %s -> %s" % (key, first.split(" from ")[0], names))
    for key, raw in STANDALONE_CASES.items():
        p = os.path.join(WORK, "sp_" + key + ".txt")
        with open(p, "wb") as f:
            f.write(raw)
        r = run([sp, p, str(D)], cwd=WORK)
        out["standalone_projection"][key] = {"input": raw.decode("latin-1"), "stdout": r.stdout.decode("latin-1")}
        print("standalone %-24s %d lines" % (key, r.stdout.count(b"\n")))
    with open(os.path.join(GOLD, "ref_parser.json"), "w") as f:
        json.dump(out, f, indent=0, sort_keys=True, separators=(",", ":"))
    print("wrote", os.path.join(GOLD, "ref_parser.json"), os.path.getsize(os.path.join(GOLD, "ref_parser.json")), "B")
    shutil.rmtree(WORK, ignore_errors=True)


if __name__ == "__main__":
    main()
