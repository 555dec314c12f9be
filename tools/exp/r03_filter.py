#!/usr/bin/env python3
"""Round-3 filter experiments on synthesised sketches (100k x 2048 unless told otherwise), interleaved in one process:
  * LDS-DMA cache policy (filter_variant 8 / 40 / 41 / 42) and sub-patch shape (pairwise_map 0 / 1 / 2)
  * radix rule of the coarse plane (coarse_radix 0 / 1): candidates and re-check time
  * with the ablation build (MVS_HIP_LIBRARY=.../libmvs_hip_abl.so): constant operands with and without injected
    candidates (pairwise_debug 4) -- what the epilogue's rare path costs apart from what the data costs
   python tools/exp/r03_filter.py [N] [d] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
abl = "abl" in os.environ.get("MVS_HIP_LIBRARY", "")
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device="cuda")
ref = {}


def prep(sk, scale=1.0):
    ss = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2 * scale + (0.0 if scale == 1.0 else 1.0)).to("cuda")
    return ctx.sketch_set(sk), n2


def run(tag, label, sset, n2, **opts):
    with ctx.options(pairwise_filter=2, **opts):
        ts, fs, cs = [], [], []
        for r in range(reps + 2):
            _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            if r >= 2:
                ts.append(ctx.kernel_ms(1))
                fs.append(ctx.kernel_ms(2))
                cs.append(ctx.kernel_ms(3))
        got = cells[:cnt].clone()
        cand = ctx.pairwise_candidates()
    same = "first"
    if tag in ref:
        same = "SAME" if (got.shape == ref[tag].shape and bool((got == ref[tag]).all())) else "DIFFERENT (%d vs %d)" % (len(got), len(ref[tag]))
    else:
        ref[tag] = got
    print("%-44s filter %.3f (min %.3f) recheck %.3f total %.3f  kept %d cand %d  %s" %
          (label, np.mean(fs), np.min(fs), np.mean(cs), np.mean(ts), cnt, cand, same), flush=True)


real = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
sset, n2 = prep(real)
quick = os.environ.get("R03_QUICK") == "1"
for rnd in range(2):                       # twice: the order effect of a warming chip is visible
    run("real", "sketches cand_regions=0 (atomic per wave)", sset, n2, cand_regions=0)
    run("real", "sketches cand_regions=1 (default)", sset, n2, cand_regions=1)
    if os.environ.get("R03_RECHECK") == "1":
        for reg in (1, 0):
            for mode in (0, 1, 2, 3):
                for blocks in (8, 16, 24, 32):
                    run("real", "regions=%d recheck_mode=%d blocks=%d" % (reg, mode, blocks), sset, n2, cand_regions=reg,
                        recheck_mode=mode, recheck_blocks=blocks)
    if quick:
        continue
    for fv in (8, 40, 41, 42):
        run("real", "sketches filter_variant=%d" % fv, sset, n2, filter_variant=fv)
    for mp in (1, 2):
        run("real", "sketches pairwise_map=%d" % mp, sset, n2, pairwise_map=mp)
        run("real", "sketches pairwise_map=%d filter_variant=40" % mp, sset, n2, pairwise_map=mp, filter_variant=40)
    run("real", "sketches coarse_radix=0 (max|v| / 127)", sset, n2, coarse_radix=0)
    run("real", "sketches coarse_radix=1 (least residual)", sset, n2, coarse_radix=1)
if abl:
    run("real", "sketches, injected candidates only", sset, n2, pairwise_debug=4)
    run("real", "sketches, injected, cand_regions=0", sset, n2, pairwise_debug=4, cand_regions=0)
sset.close()

z = torch.zeros_like(real)
z[0, 0] = 300                              # one entry beyond one limb: the set still gets the two-limb kernels
zset, zn2 = prep(z, 1e6)
run("zero", "all zero", zset, zn2)
if abl:
    run("zero-inj", "all zero + injected candidates", zset, zn2, pairwise_debug=4)
    run("zero-inj", "all zero + injected, cand_regions=0", zset, zn2, pairwise_debug=4, cand_regions=0)
    run("zero", "all zero again", zset, zn2)
    run("zero-inj", "all zero + injected candidates again", zset, zn2, pairwise_debug=4)
    run("zero-inj", "all zero + injected, cand_regions=0 again", zset, zn2, pairwise_debug=4, cand_regions=0)
zset.close()

# what the operand VALUES cost: the same sketches with every entry made non-negative (|v|), and scaled down
# offset-coded operands: every entry shifted so that the coarse values centre on +32 / +64 instead of 0 (what an offset
# coding of the coarse plane would feed the matrix cores; the radix grows with max|v|, so the spread shrinks a little)
for label, sk in (("|v| (no sign changes)", real.abs()), ("v >> 2 (two fewer significant bits)", real >> 2),
                  ("v + 32 m (coarse values centred on ~ +25)", real + 32 * 9), ("v + 64 m (centred on ~ +40)", real + 64 * 9),
                  ("v + 300 m (all positive, narrow)", real + 300 * 9), ("-|v|", -real.abs())):
    s2, m2 = prep(sk.contiguous(), 1e6)
    run(label, "sketches " + label, s2, m2)
    s2.close()
