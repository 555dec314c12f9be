"""GPU: randomised cross-check of every way the comparison result leaves the library.

    python tests/fuzz_stream.py [--seconds 300] [--seed 1] [--max-n 2500]

One case = one random sketch set (size, dimension, value range -> limb code, cluster size -> density, norms that belong to
the vectors or not) and a random way to ask for it (row range, keep test, filter on / off / forced, device budget, row
blocks, dense byte matrix or packed list, LDS stage of the encoder, tile-granular two-stage comparison with its density
threshold and its list / matrix switch).  For each case

    mvs_pairwise_rows (cell list)  ==  oracle (int32 keep test, whole rows x all columns)
    mvs_pairwise_stream (CSR pieces)  ==  the cell list
    mvs_pairwise_stream_encoded, decoded by tests/test_encode_gpu.py's independent decoder  ==  the cell list

tests/test_stream_gpu.py::test_stream_fuzz_seeded runs a fixed handful of cases under -m gpu; the script form is for
longer runs on the GPU box (prints one line per case and the seed to reproduce a failure)."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from metagenome_vector_sketches_amd import _capi, synth  # noqa: E402
from oracle import pyoracle as orc  # noqa: E402

OPTIONS = ("pairwise_filter", "stream_dense", "stream_block_rows", "encode_stage_words", "pairwise_symmetric", "filter_variant",
           "tile_dense_thr", "stream_list_cells", "stream_pipeline")


def _n2(sk):
    return np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(row))) for row in sk.astype(np.int32)])


def _triples(row_ptr, col, q, row_begin=0):
    rows = np.repeat(np.arange(len(row_ptr) - 1, dtype=np.int64) + row_begin, np.diff(row_ptr))
    return np.stack([rows, col.astype(np.int64), q.astype(np.int64)], axis=1)


def _cells_triples(cells):
    return np.stack([cells["row"].astype(np.int64), cells["col"].astype(np.int64), cells["q"].astype(np.int64)], axis=1)


def make_case(rng, max_n):
    n = int(rng.integers(2, max_n + 1)) if rng.random() < 0.8 else int(rng.integers(2, 200))
    d = int(rng.choice([64, 96, 256, 256, 512, 1000, 2048]))
    cluster = int(rng.choice([1, 2, 8, 16, 64, 100, 500, max(2, n // 3), n]))
    hashes = int(rng.choice([60, 300, 3000, 3000, 50_000, 2_000_000]))
    shared = float(rng.choice([0.4, 0.6, 0.8]))
    sk = synth.make_sketches_numpy(n, d, hashes, seed=int(rng.integers(1, 1 << 30)), cluster=min(cluster, n), shared=shared)
    kind = "as-is"
    r = rng.random()
    if r < 0.15:
        sk = np.clip(sk, -127, 127).astype(np.int32)            # one limb
        kind = "one-limb"
    elif r < 0.25:
        sk = (sk.astype(np.int64) * int(rng.choice([40, 300]))).clip(-2**30, 2**30).astype(np.int32)   # three limbs or more
        kind = "scaled"
    if rng.random() < 0.3:
        sk[rng.integers(0, n, size=max(1, n // 50))] = 0        # empty samples
    n2 = _n2(sk)
    norms = "own"
    r = rng.random()
    if r < 0.2:                                                 # boundary behaviour: norms slightly off
        n2[rng.integers(0, n, size=max(1, n // 10))] *= rng.choice([0.5, 0.97, 1.03, 2.0])
        norms = "off"
    elif r < 0.3:                                               # q beyond a byte / rows that keep nothing
        n2[rng.integers(0, n, size=max(1, n // 40))] = 1e-3
        n2[rng.integers(0, n, size=max(1, n // 20))] = 1e12
        norms = "wild"
    return sk, n2, dict(n=n, d=d, cluster=cluster, hashes=hashes, kind=kind, norms=norms)


def run_case(ctx, rng, max_n, decode_limit=150_000, log=None):
    from test_encode_gpu import _decode
    sk, n2, info = make_case(rng, max_n)
    n = info["n"]
    keep = _capi.KEEP_INT32 if rng.random() < 0.75 else _capi.KEEP_INT16
    filt = int(rng.choice([0, 1, 2, 2]))
    rb = 0 if rng.random() < 0.4 else int(rng.integers(0, n))
    re = n if rng.random() < 0.4 else int(rng.integers(rb, n + 1))
    budget = 0 if rng.random() < 0.4 else int(rng.choice([1 << 16, 1 << 20, 4 << 20, 64 << 20]))
    opts = {"pairwise_filter": filt,
            "stream_dense": int(rng.choice([0, 1, 1, 2, 3])),        # 0 packed list, 1 dense (stream by flow), 2 one stream, 3 side stream
            "stream_block_rows": int(rng.choice([0, 0, 64, 128, 200, 512])),
            "encode_stage_words": int(rng.choice([64, 64, 64, 8, 1])),
            "pairwise_symmetric": int(rng.random() < 0.85),
            # the tile-granular comparison: the ping-pong filter (large blocks' default) forced on small inputs, a random
            # density threshold per wave, and a small list bound so that the dense byte matrix takes the flagged tiles
            "filter_variant": int(rng.choice([-1, -1, 8, 8])),
            "tile_dense_thr": int(rng.choice([64, 64, 1, 8, 500, 0])),
            "stream_list_cells": int(rng.choice([1 << 26, 1 << 26, 0, 5000])),
            "stream_pipeline": int(rng.random() < 0.7)}
    info.update(keep=keep, rows=(rb, re), budget=budget, **opts)
    old = {k: ctx.get_option(k) for k in OPTIONS}
    ss = None
    try:
        for k, v in opts.items():
            ctx.set_option(k, v)
        ss = ctx.sketch_set(sk)
        info["limbs"] = ss.limbs
        cells, cnt = ctx.pairwise_rows(ss, n2, row_begin=rb, row_end=re, keep_mode=keep)
        want = _cells_triples(cells)
        if keep == _capi.KEEP_INT32 and n * n <= 4_000_000:
            ref = orc.pairwise_rows(sk, n2, row_begin=rb, row_end=re, chunk=192, threads=8)
            ref = ref[np.lexsort((ref["col"], ref["row"]))]
            assert np.array_equal(want, _cells_triples(ref)), "cell list differs from the oracle"
            assert np.array_equal(cells["dot"], ref["dot"]), "dots differ from the oracle"
        pieces = []
        n_s = ctx.pairwise_stream(ss, n2, on_block=lambda b, e, rp, c, qq: pieces.append((b, e, rp, c, qq)) and None,
                                  row_begin=rb, row_end=re, keep_mode=keep, device_budget_bytes=budget)
        st = ctx.stream_stats()
        assert n_s == cnt, "stream count %d, list %d" % (n_s, cnt)
        if re > rb:
            assert pieces and pieces[0][0] == rb and pieces[-1][1] == re
            assert all(a[1] == b[0] for a, b in zip(pieces, pieces[1:])), "pieces do not tile the row range"
            got = np.concatenate([_triples(rp, c, qq, b) for (b, e, rp, c, qq) in pieces])
            assert np.array_equal(got, want), "CSR pieces differ from the cell list"
        else:
            assert not pieces or all(int(p[2][-1]) == 0 for p in pieces)
        info.update(cells=int(cnt), blocks=st["row_blocks"], two_stage=st["two_stage"])
        if cnt <= decode_limit:
            enc = ctx.pairwise_stream_encoded(ss, n2, row_begin=rb, row_end=re, keep_mode=keep, device_budget_bytes=budget)
            assert enc["n_cells"] == cnt
            dec = _decode(enc)
            assert dec == [tuple(int(x) for x in t) for t in want], "decoded rows differ from the cell list"
            info["encoded_bytes"] = int(len(enc["bytes"]))
    finally:
        if ss is not None:
            ss.close()
        for k, v in old.items():
            ctx.set_option(k, v)
    if log:
        log(info)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=2500)
    ap.add_argument("--cases", type=int, default=0, help="stop after this many cases (0: by time)")
    args = ap.parse_args()
    from metagenome_vector_sketches_amd import Context
    ctx = Context(0)
    t0 = time.time()
    k = 0
    while time.time() - t0 < args.seconds and (args.cases == 0 or k < args.cases):
        seed = args.seed * 1_000_003 + k
        rng = np.random.default_rng(seed)
        try:
            info = run_case(ctx, rng, args.max_n)
        except BaseException:
            print("FAILED case %d (rng seed %d): rerun with --seed %d --cases %d" % (k, seed, args.seed, k + 1), flush=True)
            raise
        print("%4d %6.1fs %s" % (k, time.time() - t0, info), flush=True)
        k += 1
    ctx.close()
    print("fuzz ok: %d cases in %.0f s" % (k, time.time() - t0))


if __name__ == "__main__":
    main()
