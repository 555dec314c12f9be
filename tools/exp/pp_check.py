#!/usr/bin/env python3
"""A/B check of comparison kernel variants on synthesised sketches: identical cells + timings.
   python tools/exp/pp_check.py N d filter_variants(comma) exact_variants(comma) [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n, d = int(sys.argv[1]), int(sys.argv[2])
fvs = [int(x) for x in sys.argv[3].split(",") if x]
xvs = [int(x) for x in sys.argv[4].split(",") if x]
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 5
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
ss = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2).to("cuda")
sset = ctx.sketch_set(sk)
cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device="cuda")
ref = None


def run(label, **opts):
    global ref
    with ctx.options(**opts):
        ts, fs, cs = [], [], []
        for r in range(reps + 2):
            _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(ctx.kernel_ms(1))
                if ctx.pairwise_candidates():
                    fs.append(ctx.kernel_ms(2))
                    cs.append(ctx.kernel_ms(3))
        got = cells[:cnt].clone()
    same = "first"
    if ref is None:
        ref = got
    else:
        same = "SAME" if (got.shape == ref.shape and bool((got == ref).all())) else "DIFFERENT (%d vs %d cells)" % (len(got), len(ref))
    extra = " filter %.3f (min %.3f) recheck %.3f" % (np.mean(fs), np.min(fs), np.mean(cs)) if fs else ""
    print("%-28s kernels %.3f ms (min %.3f)%s kept %d cand %d  %s" % (label, np.mean(ts), np.min(ts), extra, cnt,
                                                                    ctx.pairwise_candidates(), same), flush=True)


for v in fvs:
    run("filter_variant=%d" % v, pairwise_filter=2, filter_variant=v)
for v in xvs:
    run("exact pairwise_variant=%d" % v, pairwise_filter=0, pairwise_variant=v)
