#!/usr/bin/env python3
"""Does a device-to-host copy share the compute units with the kernels?  Pinned D2H of 32 MiB pieces alone, then beside the exact
pairwise kernel on another stream (and the kernel's time alone / beside the copies).   python tools/exp/d2h_contention.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n, d = 60000, 2048
ctx = pkg.Context(0)
ctx.set_option("pairwise_filter", 0)                 # the exact kernel on every tile: ~10 ms of matrix-core work per call
ctx.set_stream(torch.cuda.current_stream())
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
ss = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2).to("cuda")
sset = ctx.sketch_set(sk)
cells = torch.empty((1 << 22, 4), dtype=torch.int32, device="cuda")
piece = 32 << 20
src = torch.empty(piece, dtype=torch.uint8, device="cuda")
dst = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2)]
side = torch.cuda.Stream()


def copies(k):
    with torch.cuda.stream(side):
        for i in range(k):
            dst[i & 1].copy_(src, non_blocking=True)


def compare():
    return ctx.pairwise_rows(sset, n2, cells_out=cells) if "cells_out" in ctx.pairwise_rows.__code__.co_varnames else ctx.pairwise_rows(sset, n2)


for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter(); copies(16); torch.cuda.synchronize(); t_copy = time.perf_counter() - t0
    t0 = time.perf_counter(); compare(); torch.cuda.synchronize(); t_cmp = time.perf_counter() - t0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        e0.record(side)
    copies(16)
    with torch.cuda.stream(side):
        e1.record(side)
    compare()
    torch.cuda.synchronize()
    t_both = time.perf_counter() - t0
    print("copies alone %.2f ms (%.1f GB/s) | comparison alone %.2f ms | together %.2f ms, the copies' own span %.2f ms (%.1f GB/s)" %
          (t_copy * 1e3, 16 * piece / t_copy / 1e9, t_cmp * 1e3, t_both * 1e3, e0.elapsed_time(e1), 16 * piece / (e0.elapsed_time(e1) * 1e-3) / 1e9), flush=True)
