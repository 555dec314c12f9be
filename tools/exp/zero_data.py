#!/usr/bin/env python3
"""Filter / exact kernel on real sketches against all-zero and constant sketches (same instruction stream):
what the data costs in clock.   python tools/exp/zero_data.py N d"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n, d = int(sys.argv[1]), int(sys.argv[2])
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device="cuda")


def run(label, sk):
    ss = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2 * (1.0 if label.startswith("synth") else 1e6) + 1.0).to("cuda")
    sset = ctx.sketch_set(sk)
    for mode, name in ((2, "two-stage"), (0, "exact")):
        with ctx.options(pairwise_filter=mode):
            ts, fs = [], []
            for r in range(7):
                _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
                torch.cuda.synchronize()
                if r >= 2:
                    ts.append(ctx.kernel_ms(1))
                    if mode == 2:
                        fs.append(ctx.kernel_ms(2))
        print("%-28s %-9s kernels %.3f ms%s kept %d" % (label, name, np.mean(ts), (" filter %.3f" % np.mean(fs)) if fs else "", cnt), flush=True)
    sset.close()


real = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
run("synthesised sketches", real)
z = torch.zeros_like(real)
z[0, 0] = 300                      # one entry beyond one int8 limb: the set still gets the two-limb kernels
run("all zero (but one entry)", z)
alt = torch.zeros_like(real)
alt[:, ::2] = 85                   # bytes 0x55 / 0x00 alternating along k, identical rows; thresholds keep nothing
alt[0, 0] = 300
run("0x55 / 0 alternating", alt)
run("synthesised sketches again", real)
