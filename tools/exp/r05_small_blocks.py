#!/usr/bin/env python3
"""filter kernel time of small / mid-size blocks: 128 x 128 ring tiles on the row-major coarse plane (filter_variant 0) against
the ping-pong kernel on 256 x 256 tiles of the fragment-major plane (8) -- where does the launcher's switch belong?"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth, _capi

ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
d = 2048
N = 26000
sk = synth.make_sketches_torch(N, d, 50_000, seed=7, device="cuda")
ss = torch.empty(N, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = (ss.double() / d)
sset = ctx.sketch_set(sk)
cells = torch.empty((1 << 22, 4), dtype=torch.int32, device="cuda")
ctx.set_option("pairwise_filter", 2)


def run(fn, reps=6):
    ts = []
    for r in range(reps):
        fn()
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(ctx.kernel_ms(2))
    return float(np.mean(ts))


print("symmetric square n x n (mvs_pairwise_rows on rows [0,n) of a set of n): tiles(256^2), ring ms, ping-pong ms")
for n in (2048, 3072, 4096, 6144, 8192, 10000, 12544, 16384, 20000, 26000):
    sub = ctx.sketch_set(sk[:n])
    out = []
    for v in (0, 8):
        ctx.set_option("filter_variant", v)
        out.append(run(lambda: ctx.pairwise_rows(sub, n2[:n].contiguous(), cells_out=cells)))
    t = (n + 255) // 256
    print("  n=%6d  tiles %5d  ring %.4f  pp %.4f   %s" % (n, t * (t + 1) // 2, out[0], out[1], "pp" if out[1] < out[0] else "ring"))
    sub.close()
print("rectangular block r x c, no symmetry (mvs_pairwise_block): tiles, ring ms, ping-pong ms")
for (r, c) in ((1024, 8192), (2048, 8192), (4096, 8192), (4096, 12544), (8192, 12544), (12544, 12544), (12544, 25088)):
    out = []
    for v in (0, 8):
        ctx.set_option("filter_variant", v)
        out.append(run(lambda: ctx.pairwise_block(sset, n2, 0, r, 26000 - c - (26000 - c) % 256 if False else 256, 256 + c, 0, cells, 0)))
    print("  %6d x %6d  tiles %5d  ring %.4f  pp %.4f   %s" % (r, c, ((r + 255) // 256) * ((c + 255) // 256), out[0], out[1],
                                                             "pp" if out[1] < out[0] else "ring"))
