"""GPU: randomised cross-check of the projection (mvs_project_csr / _stats) against the oracle.

    python tests/fuzz_project.py [--seconds 300] [--seed 1]

One case = random sample sizes (empty, single, around the batch / unit borders, a few very long ones), a random dimension
(incl. d % 64 != 0 and d < 64), hash values drawn from adversarial families (uniform, FracMinHash-like below 2^64 / 1000,
values next to 0 and 2^64 so that h + 64 * block wraps, values whose bits 8..29 are all ones -- the carry hazard of the
shared first round --, runs of consecutive integers), a random kernel variant, host or device input.  Checked: the
sketches, the fused sums of squares and the largest |v| against oracle/ (src/random_projection.cpp:9-26 restated).

tests/test_random_gpu.py::test_projection_fuzz_seeded runs a fixed handful under -m gpu."""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from oracle import pyoracle as orc  # noqa: E402

U64 = np.uint64
SIZES = [0, 0, 1, 2, 5, 63, 64, 65, 511, 512, 513, 960, 961, 2047, 4097, 8128, 8129, 30_000, 65_536, 65_537, 140_000]
VARIANTS = [0, 0, 0, 1, 2, 12, 14]


def draw_hashes(rng, n, family):
    if n == 0:
        return np.zeros(0, dtype=U64)
    if family == "uniform":
        h = rng.integers(0, 2**64 - 1, size=n, dtype=U64, endpoint=True)
    elif family == "fracminhash":
        h = rng.integers(0, 18446744073709552, size=n, dtype=U64)
    elif family == "edges":             # around 0 and 2^64: h + 64 b + golden wraps in different places
        h = (rng.integers(0, 1 << 14, size=n, dtype=U64) * U64(rng.choice([1, 64, 1 << 30]))
             + U64(rng.choice([0, 2**64 - (1 << 20), 2**63 - 300, 2**64 - 0x9e3779b97f4a7c15 - 500])))
    elif family == "hazard":            # bits 8..29 of (h + 64 * first block + golden) all ones for many of them
        golden = 0x9e3779b97f4a7c15
        base = rng.integers(0, 2**64 - 1, size=n, dtype=U64, endpoint=True)
        x = (base & U64(0xFFFFFFFFC00000FF)) | U64(0x3FFFFF00)           # the sum with the pattern set
        blk = U64(64 * int(rng.choice([0, 4, 8, 28])) + golden & (2**64 - 1))
        h = x - blk                                                       # wraps like the kernel's add does
        plain = rng.random(n) < 0.5
        h[plain] = base[plain]
    else:                               # runs of consecutive integers
        start = rng.integers(0, 2**64 - 1, dtype=U64, endpoint=True)
        h = start + np.arange(n, dtype=U64) * U64(rng.choice([1, 2, 64, 65]))
    return np.unique(h)                 # the precondition of the C ABI: unique hashes per sample


def run_case(ctx, rng, max_total=600_000, log=None):
    import torch
    d = int(rng.choice([1, 7, 63, 64, 65, 128, 200, 256, 777, 1024, 2048, 2048, 2048, 3000, 4096]))
    n_samples = int(rng.integers(1, 48))
    sizes = rng.choice(SIZES, size=n_samples)
    while sizes.sum() > max_total:
        sizes[int(np.argmax(sizes))] = int(rng.choice(SIZES[:12]))
    fam = str(rng.choice(["uniform", "fracminhash", "edges", "hazard", "runs"]))
    lists = [draw_hashes(rng, int(s), fam) for s in sizes]
    offs = np.zeros(n_samples + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate(lists) if offs[-1] else np.zeros(0, dtype=U64)
    variant = int(rng.choice(VARIANTS))
    where = str(rng.choice(["host", "device", "stats"]))
    old = ctx.get_option("project_variant")
    info = dict(d=d, samples=n_samples, hashes=int(offs[-1]), family=fam, variant=variant, where=where)
    try:
        ctx.set_option("project_variant", variant)
        want = orc.project_csr(flat, offs, d, threads=8, fast=True)
        if where == "host":
            got = ctx.project_csr(flat, offs, d)
        else:
            h_t = torch.from_numpy(flat.view(np.int64)).to("cuda") if len(flat) else torch.zeros(0, dtype=torch.int64, device="cuda")
            out = torch.empty((n_samples, d), dtype=torch.int32, device="cuda")
            if where == "device":
                ctx.project_csr(h_t, offs, d, out=out)
                ctx.synchronize()
            else:
                ss = torch.empty(n_samples, dtype=torch.int64, device="cuda")
                m = ctx.project_csr_stats(h_t, offs, d, out, ss)
                ctx.synchronize()
                w64 = want.astype(np.int64)
                assert np.array_equal(ss.cpu().numpy(), (w64 * w64).sum(axis=1)), "sums of squares"
                assert m == (int(np.abs(w64).max()) if w64.size else 0), "largest |v|"
            got = out.cpu().numpy()
        assert np.array_equal(got, want), "sketches differ from the oracle"
    finally:
        ctx.set_option("project_variant", old)
    if log:
        log(info)
    return info


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=300.0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--cases", type=int, default=0)
    args = ap.parse_args()
    from metagenome_vector_sketches_amd import Context
    ctx = Context(0)
    t0 = time.time()
    k = 0
    while time.time() - t0 < args.seconds and (args.cases == 0 or k < args.cases):
        seed = args.seed * 1_000_003 + k
        try:
            info = run_case(ctx, np.random.default_rng(seed))
        except BaseException:
            print("FAILED case %d (rng seed %d): rerun with --seed %d --cases %d" % (k, seed, args.seed, k + 1), flush=True)
            raise
        print("%4d %6.1fs %s" % (k, time.time() - t0, info), flush=True)
        k += 1
    ctx.close()
    print("fuzz ok: %d cases in %.0f s" % (k, time.time() - t0))


if __name__ == "__main__":
    main()
