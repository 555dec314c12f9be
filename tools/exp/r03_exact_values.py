#!/usr/bin/env python3
"""What the operand VALUES cost the exact two-limb kernel (k_pairwise_pp, MODE 0/1) -- a what-if like the filter's
(tools/exp/r03_filter.py): the same 100k x 2048 sketches, their entries re-written so that the base-256 limb planes the
kernel reads look like another limb code would make them.  Norms are inflated so that nothing is kept (the epilogue's
rare path stays out of the picture); results are garbage by design, only the kernel time matters.
   python tools/exp/r03_exact_values.py [N] [d] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
cells = torch.empty((1 << 22, 4), dtype=torch.int32, device="cuda")
n2 = torch.full((n,), 1e30, dtype=torch.float64, device="cuda")


def run(label, sk):
    sset = ctx.sketch_set(sk.contiguous())
    with ctx.options(pairwise_filter=0):
        ts = []
        for r in range(reps + 2):
            _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(ctx.kernel_ms(1))
    print("%-64s limbs %d  exact kernel %.2f ms (min %.2f)  kept %d" % (label, sset.limbs, np.mean(ts), np.min(ts), cnt), flush=True)
    sset.close()


real = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
for rnd in range(2):
    run("sketches as they are (lo uniform signed byte, hi in -5..4)", real)
    # base-128 digits of v + 8192, both non-negative: lo' = (v + 8192) & 127 in 0..127, hi' = (v + 8192) >> 7 in 64 +- 9
    u = real + 8192
    run("offset-coded base-128 digits (lo 0..127, hi 64 +- 9)", (u >> 7) * 256 + (u & 127))
    # signed base-128 digits (what MVS_LIMBS_K3 uses for its first two planes): lo in -64..63, hi = (v - lo) / 128
    lo = ((real + 64) & 127) - 64
    run("signed base-128 digits (lo -64..63, hi in -9..9)", ((real - lo) >> 7) * 256 + lo)
    run("|v|", real.abs())
    # lo as it is, hi made non-negative (hi + 8)
    run("lo as it is, hi + 8 (non-negative high limb)", real + 8 * 256)
    z = torch.zeros_like(real)
    z[0, 0] = 300
    run("all zero", z)
