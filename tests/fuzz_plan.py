"""GPU: randomised cross-check of the block plans (include/mvs_hip.h "block plans") against mvs_pairwise_rows.

    python tests/fuzz_plan.py [--seconds 300] [--seed 1] [--max-n 6000]

One case = one random sketch set (size, dimension, cluster size -> density) split over a random number of ranks, all of them
played by this process on the one card (tests/test_plan_gpu.py: Split / _union), with a random way to run the plans: the peers'
columns in 1 .. 4 chunks, symmetric schedule or rows x all columns, the other ranks' limb planes present or poisoned and rebuilt
from their low limbs (mvs_plan_wire), the density threshold of the flagged tiles, the plan waiting for its own counts or running
ahead of them (option plan_speculate: the second run of a shape takes its sizes from the first), odd ranks announcing every row
(mvs_plan_rows_ready), even ranks not.  The union of the ranks' shards must equal the cell list of mvs_pairwise_rows on the whole
set -- which tests/test_pairwise_gpu.py and the golden fixtures pin against the oracle and the reference's own functions
(src/pairwise_comp_optimized.cpp:135-147, :654-665) -- cell for cell, in (row, col) order.  Prints one line per case; a failure
prints the seed and the case to reproduce it."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

from metagenome_vector_sketches_amd import Context, _capi, synth  # noqa: E402
import test_plan_gpu as tp  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--max-n", type=int, default=6000)
    ap.add_argument("--cases", type=int, default=0)
    args = ap.parse_args()
    rng = np.random.default_rng(args.seed)
    ctx = Context(0)
    ctx.set_stream(torch.cuda.current_stream())
    t_end = time.time() + args.seconds
    case = 0
    while time.time() < t_end and (args.cases == 0 or case < args.cases):
        case += 1
        n = int(rng.integers(300, args.max_n))
        d = int(rng.choice([64, 128, 256, 384, 512, 1024, 2048]))
        if n * d > 6_000_000:
            d = 256
        world = int(rng.integers(1, 9))
        cluster = int(rng.choice([4, 8, 16, 64, 300, 700] if world >= 4 else [4, 8, 16, 64, 300]))   # (a rank's buffers hold 400 n cells)
        chunks = int(rng.integers(1, 5))
        symmetric = bool(rng.integers(0, 4) != 0)
        wire = bool(rng.integers(0, 2)) and world > 1
        thr = int(rng.choice([0, 16, 64, 256]))
        speculate = int(rng.integers(0, 2))
        nh = int(rng.choice([500, 3000, 20000]))
        desc = "n %d d %d world %d cluster %d chunks %d symmetric %d wire %d tile_dense_thr %d plan_speculate %d hashes %d" % (
            n, d, world, cluster, chunks, symmetric, wire, thr, speculate, nh)
        sk = synth.make_sketches_numpy(n, d, nh, seed=int(rng.integers(1 << 30)), cluster=cluster)
        n2 = (sk.astype(np.float64) ** 2).sum(axis=1) / d
        n2 = np.array([float("%g" % v) for v in np.sqrt(n2)]) ** 2          # norms as vector_norms.txt carries them
        ctx.set_option("pairwise_filter", 2)
        ctx.set_option("tile_dense_thr", thr)
        plain = ctx.sketch_set(sk)
        ref, cnt = ctx.pairwise_rows(plain, n2)
        plain.close()
        ref = ref[np.lexsort((ref["col"], ref["row"]))]
        want = np.stack([ref[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
        if len(want) > 350 * n:               # the harness gives a rank room for 400 n cells (Split.rank_cells): another case
            print("case %d: %s -> %d cells: more than the harness's buffers hold, skipped" % (case, desc, len(want)), flush=True)
            continue
        print("case %d: %s ..." % (case, desc), flush=True)
        split = tp.Split(ctx, sk, n2, world)
        ok = True
        with ctx.options(plan_speculate=speculate):
            for rep in range(2 if speculate else 1):                         # the second run of a shape runs ahead of its counts
                got, per_rank = tp._union(split, symmetric=symmetric, chunks=chunks, wire=wire)
                if not np.array_equal(got, want):
                    ok = False
                    break
        torch.cuda.synchronize()
        split.sset.close()
        cand = sum(x[2]["candidates"] for x in per_rank)
        flagged = sum(x[2]["flagged_tiles"] for x in per_rank)
        print("case %d: %s -> %d cells, %d candidates, %d flagged tiles: %s" % (case, desc, len(want), cand, flagged,
                                                                              "equal" if ok else "DIFFERENT"), flush=True)
        if not ok:
            print("FAILED with --seed %d at case %d" % (args.seed, case), flush=True)
            sys.exit(1)
    print("%d cases, all equal (seed %d)" % (case, args.seed))
    ctx.close()


if __name__ == "__main__":
    main()
