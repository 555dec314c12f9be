#!/usr/bin/env python3
"""Host-side wall time of each phase of one bench.py step (synchronising after every phase, so the sum is
larger than the real step): where the time outside the two dominant kernels goes.
   python tools/step_breakdown.py [samples] [hashes] [d]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi, synth
from bench import fast_norm_sq

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
hashes, offsets = synth.make_csr_torch(S, NH, seed=1234, device=dev, cluster=16, shared=0.4)
sketches = torch.empty((S, D), dtype=torch.int32, device=dev)
sumsq = torch.empty(S, dtype=torch.int64, device=dev)
cells = torch.empty((max(1 << 20, 64 * S), 4), dtype=torch.int32, device=dev)
limbs = 2
n_alloc, d_pad, nbytes = ctx.limb_geometry(S, D, limbs)
planes = torch.zeros(nbytes, dtype=torch.int8, device=dev)


def phase(acc, name, fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = fn()
    torch.cuda.synchronize()
    acc.setdefault(name, []).append((time.perf_counter() - t0) * 1e3)
    return out


acc = {}
for it in range(8):
    a = {} if it < 3 else acc
    max_abs = phase(a, "project_csr_stats (K1 + max|v| readback)", lambda: ctx.project_csr_stats(hashes, offsets, D, sketches, sumsq))
    ss_host = phase(a, "sumsq -> host", lambda: sumsq.cpu().numpy())
    n2 = phase(a, "norms text round trip (host)", lambda: fast_norm_sq(ss_host, D))
    phase(a, "limb split", lambda: ctx.limb_split(sketches, limbs, planes, d_pad, 0))
    n2_dev = phase(a, "norms -> device", lambda: torch.from_numpy(n2).to(dev))
    sset = phase(a, "sketch_set_from_planes", lambda: ctx.sketch_set_from_planes(planes, S, n_alloc, D, d_pad, limbs))
    res = phase(a, "pairwise_rows (coarse, filter, exact, sort)", lambda: ctx.pairwise_rows(sset, n2_dev, cells_out=cells))
    a.setdefault("  of which filter + exact kernels (HIP events)", []).append(ctx.kernel_ms(1))
    phase(a, "set close", sset.close)
tot = 0.0
for k, v in acc.items():
    m = sum(v) / len(v)
    if not k.startswith("  "):
        tot += m
    print("%-50s %8.3f ms" % (k, m))
print("%-50s %8.3f ms   (kept %d, candidates %d)" % ("sum", tot, res[1], ctx.pairwise_candidates()))
