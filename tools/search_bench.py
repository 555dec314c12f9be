#!/usr/bin/env python3
"""mvs_search_block on resident limb planes: nq query sketches against N database sketches (the device part of
search.search_index, SURVEY 8f row 4 -- the reference runs FAISS IndexFlatIP on float32 copies on the CPU).
   python tools/search_bench.py [N] [d] [nq] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 64
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
sset = ctx.sketch_set_alloc(n + nq, d, 2)
step = 100_000
ss_all = torch.empty(n + nq, dtype=torch.int64, device="cuda")
for r0 in range(0, n + nq, step):                      # the queries are the last nq rows: members of the last clusters
    rows = min(step, n + nq - r0)
    sk = synth.make_sketches_torch(rows, d, 50_000, seed=4567 + r0, device="cuda")
    ctx.sumsq(sk, out=ss_all[r0:r0 + rows])
    sset.fill(sk, r0)
    del sk
n2 = (ss_all.double() / d)
cells = torch.empty((1 << 24, 4), dtype=torch.int32, device="cuda")
import time
for j in (0.1, 0.05):
    ts, ks = [], []
    for r in range(reps + 2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        cnt = ctx.search_block(sset, n2, j, n, n + nq, 0, n, cells)
        torch.cuda.synchronize()
        if r >= 2:
            ts.append((time.perf_counter() - t0) * 1e3)
            ks.append(ctx.kernel_ms(1))
    cand = ctx.pairwise_candidates()
    try:                                    # the filter's interval exists only when the two-stage comparison ran
        stages = "two-stage: filter %.3f ms + re-check %.3f ms, %d candidates" % (ctx.kernel_ms(2), ctx.kernel_ms(3), cand)
    except Exception:                       # noqa: BLE001
        stages = "exact kernel"
    planes_gb = (n + nq) * sset.d_pad * 2 / 1e9
    print("N %d d %d queries %d j > %.2f: %d hits, wall %.3f ms (min %.3f), kernels %.3f ms, %.2f GB of limb planes => %.2f TB/s, %.3g pairs/s (%s)"
          % (n, d, nq, j, cnt, np.mean(ts), np.min(ts), np.mean(ks), planes_gb, planes_gb / (np.mean(ks) * 1e-3) / 1e3,
             n * nq / (np.mean(ts) * 1e-3), stages), flush=True)
