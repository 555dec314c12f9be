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
