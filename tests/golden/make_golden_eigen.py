#!/usr/bin/env python3
"""Adds to tests/golden/kat.json (dev container only, needs /root/reference for `make -C oracle ref`):
  * eigen_gemm_cases: int32 products `block_i.transpose() * block_j` (src/pairwise_comp_optimized.cpp:135) computed by
    the reference's own vendored Eigen (oracle/_ref/eigen_gemm_check, built from oracle/eigen_gemm_check.cpp + the
    reference's include/Eigen) on generated blocks -- the one part of the pairwise path that can be pinned against
    reference code here (the translation unit itself needs the absent `bits` submodule);
  * provenance: which fixtures come from the reference binaries and which from this repository's oracle.
Inputs are regenerated from (seed, sample, k) by the tests: value = mix(seed*1000003 + sample*65537 + k) % (2*mag+1) - mag
with mix = splitmix64 (samples 0.. for block_i, 1000.. for block_j)."""
import json
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
EXE = os.path.join(ROOT, "oracle", "_ref", "eigen_gemm_check")

CASES = [  # d, c_i, c_j, seed, magnitude
    (2048, 5, 7, 11, 30000),          # two base-256 limbs, products wrap many times
    (2048, 4, 4, 12, 900),            # sketch-like magnitudes, no wrap
    (4096, 3, 5, 13, 127),            # one limb
    (100, 6, 3, 14, 2000000000),      # four limbs (vector-ALU kernel), every product wraps
    (2112, 2, 9, 15, 8000),           # d not a multiple of 128
]


def main():
    subprocess.run(["make", "-s", "-C", os.path.join(ROOT, "oracle"), "ref"], check=True)
    with open(os.path.join(HERE, "kat.json")) as f:
        kat = json.load(f)
    out = []
    for d, ci, cj, seed, mag in CASES:
        r = subprocess.run([EXE, str(d), str(ci), str(cj), str(seed), str(mag)], capture_output=True, text=True, check=True)
        dots = [int(x) for x in r.stdout.split()]
        assert len(dots) == ci * cj
        out.append({"d": d, "c_i": ci, "c_j": cj, "seed": seed, "magnitude": mag, "dots": dots})
    kat["eigen_gemm_cases"] = out
    kat["provenance"] = {
        "reference binaries (oracle/_ref, compiled from /root/reference/src by oracle/Makefile)":
            ["toy_db.npz", "toy_vector_norms.txt", "toy_sketch_digests.json", "toy_hashes.npz",
             "kat.json: splitmix64_of_1, kat_123_d8, standalone_projection cases, toy_* digests and norms"],
        "reference's vendored Eigen (oracle/_ref/eigen_gemm_check)": ["kat.json: eigen_gemm_cases"],
        "SURVEY.md section 4 (values the survey session recorded from a reference build with a stand-in codec header; "
        "kept as cross-checks only)": ["kat.json: survey_kept_cells, survey_pairwise_pins"],
        "oracle-generated (this repository's oracle/mvs_oracle.c on the reference-built toy DB; NOT reference output)":
            ["toy_pairwise_cells.txt", "toy_pairwise_cells_int16.txt"],
    }
    with open(os.path.join(HERE, "kat.json"), "w") as f:
        json.dump(kat, f, indent=1, sort_keys=True)
    print("eigen_gemm_cases:", [(c["d"], c["c_i"], c["c_j"]) for c in out])


if __name__ == "__main__":
    main()
