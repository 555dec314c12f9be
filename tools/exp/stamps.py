#!/usr/bin/env python3
"""Per-workgroup timeline of the filter pass (ablation library, pairwise_debug bit 8 -> /tmp/mvs_stamps.bin).
   MVS_HIP_LIBRARY=.../libmvs_hip_abl.so python tools/exp/stamps.py N d"""
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
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
ss = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2).to("cuda")
sset = ctx.sketch_set(sk)
cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device="cuda")
with ctx.options(pairwise_filter=2):
    for _ in range(3):
        ctx.pairwise_rows(sset, n2, cells_out=cells)
        torch.cuda.synchronize()
with ctx.options(pairwise_filter=2, pairwise_debug=8):
    ctx.pairwise_rows(sset, n2, cells_out=cells)
    torch.cuda.synchronize()
    print("filter %.3f ms" % ctx.kernel_ms(2))
st = np.fromfile("/tmp/mvs_stamps.bin", dtype=np.uint64).reshape(-1, 8)
used = st[:, 1] != 0
st = st[used]
xcc = (st[:, 0] >> np.uint64(32)).astype(np.int64) & 0xf
hw = (st[:, 0] & np.uint64(0xffffffff)).astype(np.int64)
cu = (hw >> 8) & 0xf
sh = (hw >> 12) & 0x1
se = (hw >> 13) & 0x7
t0 = st[:, 1].astype(np.int64)
t1 = st[:, 2].astype(np.int64)
tk = st[:, 3].astype(np.int64)
base = t0.min()
real = t1 > t0
tick = 1e-2   # realtime clock: 100 MHz -> 10 ns = 1e-2 us
print("workgroups %d, with a tile %d; span %.1f us" % (len(st), real.sum(), (t1.max() - base) * tick))
for x in range(8):
    m = xcc == x
    mr = m & real
    print("XCC %d: wgs %6d tiles %6d  first start %8.1f us  last end %8.1f us  mean tile %6.2f us (k-loop %6.2f, epilogue %5.2f)" % (
        x, m.sum(), mr.sum(), (t0[m].min() - base) * tick, (t1[mr].max() - base) * tick,
        ((t1 - t0)[mr]).mean() * tick, ((tk - t0)[mr]).mean() * tick, ((t1 - tk)[mr]).mean() * tick))
ta, tb, tc_ = st[:, 4].astype(np.int64), st[:, 5].astype(np.int64), st[:, 6].astype(np.int64)
mr = real & (ta > 0)
print("epilogue parts (us): wait for the other group %.2f, staging + barrier %.2f, sweep %.2f, rest %.2f" % (
    ((ta - tk)[mr]).mean() * tick, ((tb - ta)[mr]).mean() * tick, ((tc_ - tb)[mr]).mean() * tick, ((t1 - tc_)[mr]).mean() * tick))
# per CU: busy fraction and gaps between consecutive tiles
key = xcc * 4096 + se * 512 + sh * 256 + cu
gaps = []
busy = []
for k in np.unique(key[real]):
    m = (key == k) & real
    o = np.argsort(t0[m])
    a, b = t0[m][o], t1[m][o]
    gaps.append(a[1:] - b[:-1])
    busy.append((b - a).sum() / float(b.max() - base))
gaps = np.concatenate(gaps) * tick
print("CUs seen %d; busy fraction of [kernel start, CU's last end]: mean %.3f min %.3f" % (len(busy), np.mean(busy), np.min(busy)))
print("gap between consecutive tiles on a CU: mean %.2f us, median %.2f, p90 %.2f, p99 %.2f, max %.1f; negative (overlap) %d" % (
    gaps.mean(), np.median(gaps), np.percentile(gaps, 90), np.percentile(gaps, 99), gaps.max(), (gaps < 0).sum()))
ends = []
for k in np.unique(key[real]):
    m = (key == k) & real
    ends.append((t1[m].max() - base) * tick)
ends = np.array(ends)
print("CU last-end: min %.1f us, mean %.1f, max %.1f" % (ends.min(), ends.mean(), ends.max()))
