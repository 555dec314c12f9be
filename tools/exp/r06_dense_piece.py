#!/usr/bin/env python3
"""The 10 %-dense 100k leg (encoded rows streamed to the host) against the size of a device-to-host piece: a copy is a blit
kernel that fills the card while the link drains it (profiles/r05_exp_dense_copy.log), so a second-half kernel of a row block
that starts beside one ends with it -- 0.6 ms with 32 MiB pieces.  Smaller pieces = shorter copies to queue behind.
   python tools/exp/r06_dense_piece.py [N] [d] [cluster] [options: name=value ...]"""
import ctypes
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
cluster = int(sys.argv[3]) if len(sys.argv) > 3 else 10_000
extra = dict(kv.split("=") for kv in sys.argv[4:])
dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
for k, v in extra.items():
    ctx.set_option(k, int(v))
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device=dev, cluster=cluster)
ss = torch.empty(n, dtype=torch.int64, device=dev)
ctx.sumsq(sk, out=ss)
x = np.sqrt(ss.cpu().numpy().astype(np.float64) / d)
n2 = torch.from_numpy(np.array([float("%g" % v) for v in x]) ** 2).to(dev)
sset = ctx.sketch_set(sk)
del sk
seen = {"cells": 0, "pieces": 0}


def count(_user, bp):
    seen["cells"] += bp.contents.n_cells
    seen["pieces"] += 1
    return 0


ecb = _capi.ENCODED_ROWS_CB(count)


def stream():
    seen.update(cells=0, pieces=0)
    cnt = ctypes.c_int64()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = ctx.lib.mvs_pairwise_stream_encoded(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, ecb, None,
                                             ctypes.byref(cnt))
    wall = (time.perf_counter() - t0) * 1e3
    assert rc == 0 and seen["cells"] == cnt.value, (rc, seen, cnt.value)
    st = ctx.stream_stats()
    return wall, st["kernel_ms"], st["bytes"], st["row_blocks"], seen["pieces"], int(st["two_stage"]), int(cnt.value)


sweep = [(32, 1, 0, 0), (32, 5, 0, 0), (32, 5, 1, 0), (32, 5, 0, 1), (32, 5, 1, 1), (64, 5, 1, 1), (32, 2, 1, 1), (32, 1, 0, 0), (32, 5, 1, 1)]
for mib, dense_mode, spec, copy in sweep:
    ctx.set_option("stream_piece_mib", mib)
    ctx.set_option("stream_dense", dense_mode)
    ctx.set_option("stream_spec", spec)
    ctx.set_option("stream_copy", copy)
    print("stream_copy %d stream_spec %d stream_dense %d " % (copy, spec, dense_mode), end="")
    runs = [stream() for _ in range(4)][1:]
    w = sorted(r[0] for r in runs)
    print("piece %3d MiB: wall %s ms (median %.2f)  kernels %.2f ms  %d bytes = %.2f ms of a 55 GB/s link  %d row blocks  %d pieces  path %d  %d cells"
          % (mib, " ".join("%.2f" % r[0] for r in runs), w[len(w) // 2], runs[-1][1], runs[-1][2], runs[-1][2] / 55e6, runs[-1][3], runs[-1][4],
             runs[-1][5], runs[-1][6]), flush=True)
