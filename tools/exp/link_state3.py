#!/usr/bin/env python3
"""Which earlier call makes the 10 %-dense encoded stream's D2H copies slow?  Variations of bench.density_leg's sequence."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
import bench
import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi, synth
dev = torch.device("cuda", 0)
n, d = 100000, 2048
variant = sys.argv[1]
ctx = pkg.Context(0); ctx.set_stream(torch.cuda.current_stream()); ctx.set_timing(True)
seen = {"cells": 0}
def count(_u, bp):
    seen["cells"] += bp.contents.n_cells
    return 0
ecb = _capi.ENCODED_ROWS_CB(count)
def make(c):
    sk = synth.make_sketches_torch(n, d, 50000, seed=2345, device=dev, cluster=c)
    ss = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(bench.fast_norm_sq(ss.cpu().numpy(), d)).to(dev)
    sset = ctx.sketch_set(sk)
    return sset, n2
def stream(sset, n2, reps=3):
    ws = []
    for r in range(reps):
        cnt = ctypes.c_int64()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        rc = ctx.lib.mvs_pairwise_stream_encoded(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, ecb, None, ctypes.byref(cnt))
        ws.append((time.perf_counter() - t0) * 1e3)
        assert rc == 0
    return min(ws[1:])
if variant == "A":      # 10000 only, two-stage only
    s, n2 = make(10000); print(variant, "10000:", round(stream(s, n2), 1))
elif variant == "B":    # 16 two-stage, then 10000
    s, n2 = make(16); print(variant, "16:", round(stream(s, n2), 1)); s.close()
    s, n2 = make(10000); print(variant, "10000:", round(stream(s, n2), 1))
elif variant == "C":    # 1024 two-stage, then 10000
    s, n2 = make(1024); print(variant, "1024:", round(stream(s, n2), 1)); s.close()
    s, n2 = make(10000); print(variant, "10000:", round(stream(s, n2), 1))
elif variant == "D":    # 16 exact (filter 0), then 10000
    s, n2 = make(16)
    with ctx.options(pairwise_filter=0):
        print(variant, "16 exact:", round(stream(s, n2), 1))
    s.close()
    s, n2 = make(10000); print(variant, "10000:", round(stream(s, n2), 1))
elif variant == "E":    # 10000 twice with a fresh set in between (allocation churn only)
    s, n2 = make(10000); print(variant, "10000:", round(stream(s, n2), 1)); s.close(); del s; torch.cuda.empty_cache()
    s, n2 = make(10000); print(variant, "10000 again:", round(stream(s, n2), 1))
