#!/usr/bin/env python3
"""Where the configs[1] step's time outside the two big kernels goes (host view, one GPU)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import bench
import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import parallel, synth

S, NH, D = 10_000, 50_000, 2048
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
hashes, offsets = synth.make_csr_torch(S, NH, seed=1234, device=dev, cluster=16, shared=0.4)
sketches = torch.empty((S, D), dtype=torch.int32, device=dev)
sumsq = torch.empty(S, dtype=torch.int64, device=dev)
cells = torch.empty((1 << 20, 4), dtype=torch.int32, device=dev)
sc = parallel.ShardedComparison(parallel.GpuOps(ctx, dev), 0, 1)
laps = {}


def lap(name, t0):
    t = time.perf_counter()
    laps.setdefault(name, []).append((t - t0) * 1e3)
    return t


for it in range(30):
    torch.cuda.synchronize()
    t = time.perf_counter()
    max_abs = ctx.project_csr_stats(hashes, offsets, D, sketches, sumsq)
    t = lap("project_csr_stats (K1 + max_abs readback)", t)
    h = sumsq.cpu().numpy()
    t = lap("sumsq D2H", t)
    n2 = bench.fast_norm_sq(h, D)
    t = lap("norm text round trip (numpy)", t)
    _, cnt, info = sc.run(sketches, n2, S, cells_out=cells, max_abs_local=max_abs)
    t = lap("sc.run (H2D norms, limb split, coarse, filter, re-check, count)", t)
    torch.cuda.synchronize()
    t = lap("final sync", t)
for k, v in laps.items():
    print("%-70s %.3f ms (min %.3f)" % (k, np.mean(v[5:]), np.min(v[5:])))
print("K1 %.3f ms  K2 %.3f ms" % (ctx.kernel_ms(0), ctx.kernel_ms(1)))
