#!/usr/bin/env python3
"""A/B of the projection kernel variants (option project_variant): identical sketches + kernel times.
   python tools/exp/k1_check.py [samples] [hashes] [d] [variants]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

S = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
NH = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
D = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
variants = [int(x) for x in (sys.argv[4] if len(sys.argv) > 4 else "0,12,14,0,12,14").split(",")]
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
hashes, offsets = synth.make_csr_torch(S, NH, seed=1234, device="cuda")
# hazard cases for the shared-round variants: hashes whose low word + golden + 64*b0 has bits 8..29 all ones
golden = 0x9e3779b97f4a7c15
hz = []
for b0 in range(0, (D + 63) // 64, 2):
    for low8 in (0, 63, 64, 200, 255):
        target = 0x3fffff00 | low8 | (np.random.randint(0, 4) << 30)
        lo = (target - ((golden + 64 * b0) & 0xffffffff)) & 0xffffffff
        hz.append((np.random.randint(0, 2**31) << 32) | lo)
hz = torch.tensor(np.array(hz, dtype=np.uint64).view(np.int64), device="cuda")
hashes[:len(hz)] = hz                                     # all inside sample 0
out = torch.empty((S, D), dtype=torch.int32, device="cuda")
ss = torch.empty(S, dtype=torch.int64, device="cuda")
ref = None
for v in variants:
    ctx.set_option("project_variant", v)
    ts = []
    for r in range(6):
        ctx.project_csr_stats(hashes, offsets, D, out, ss)
        torch.cuda.synchronize()
        if r >= 2:
            ts.append(ctx.kernel_ms(0))
    got = out.clone()
    if ref is None:
        ref = got
        same = "first"
        from oracle import pyoracle as orc
        hh = hashes[:NH].cpu().numpy().view(np.uint64)
        assert np.array_equal(got[0].cpu().numpy(), orc.project(hh, D)), "sample 0 differs from the oracle"
    else:
        same = "SAME" if bool((got == ref).all()) else "DIFFERENT"
    print("project_variant=%-3d kernel %.3f ms (min %.3f) -> %.0f samples/s  %s" % (v, np.mean(ts), np.min(ts), S / (np.mean(ts) * 1e-3), same), flush=True)
