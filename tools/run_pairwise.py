#!/usr/bin/env python3
"""Run the pairwise comparison alone on synthesised sketches (profiling helper).
   python tools/run_pairwise.py [N] [d] [reps] [hashes per sample]
The MVS_* environment (read once by the context) selects variants, e.g. MVS_PAIRWISE_FILTER=0 for the exact kernel."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
nh = int(sys.argv[4]) if len(sys.argv) > 4 else 50_000
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
sk = synth.make_sketches_torch(n, d, nh, seed=2345, device="cuda")
if os.environ.get("ZERO_DATA") == "1":       # power experiment: same instruction stream on all-zero operands
    sk.zero_()
    sk[:, 0] = 200                            # keeps limbs == 2 (max |v| > 127)
ss = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2).to("cuda")
sset = ctx.sketch_set(sk)
cells = torch.empty((int(os.environ.get("CELLS_CAP", max(1 << 22, 64 * n))), 4), dtype=torch.int32, device="cuda")
for r in range(reps):
    t0 = time.perf_counter()
    _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = ctx.kernel_ms(1)
    if ctx.pairwise_candidates():
        print("filter %.3f ms  re-check %.3f ms" % (ctx.kernel_ms(2), ctx.kernel_ms(3)), end="  ")
    print("candidates %d" % ctx.pairwise_candidates(), end="  ")
    print("N=%d d=%d limbs=%d kept=%d wall %.3f ms kernel %.3f ms -> %.3g cells/s, %.1f algorithmic TFLOP/s, MFMA issue %.1f%%"
          % (n, d, sset.limbs, cnt, dt * 1e3, ms, n * n / (ms * 1e-3), 2.0 * d * n * n / (ms * 1e-3) / 1e12,
             2.0 * d * n * n * (1 if ctx.pairwise_candidates() else {1: 1, 0x103: 3, 2: 4}.get(sset.limbs, 0)) *
             (0.5 + 64.0 / n if ctx.get_option("pairwise_symmetric") else 1.0) / (ms * 1e-3) / 5e15 * 100))
