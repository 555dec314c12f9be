#!/usr/bin/env python3
"""mvs_pairwise_stream on synthesised sketches: compare + download wall against the comparison kernels' own time and the
bare link time of the same bytes (pinned D2H).   python tools/stream_bench.py N d cluster [reps]
cluster = related samples per cluster: 16 is the sparse default; N/10 keeps 10 % of all cells, N/3 a third (the density of
the reference's toy set).  The callback only counts (a shard writer's work is measured by tools/cli_pairwise_e2e.py)."""
import ctypes
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi, synth

n, d, cluster = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
encoded = len(sys.argv) > 5 and sys.argv[5] == "encoded"     # rows encoded in the shard codec on the device
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(os.environ.get("MVS_BENCH_TIMING", "1") != "0")     # 0: no events in the library (kernels_ms reads 0)
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda", cluster=cluster)
ss = torch.empty(n, dtype=torch.int64, device="cuda")
ctx.sumsq(sk, out=ss)
n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2).to("cuda")
sset = ctx.sketch_set(sk)
del sk
seen = {"cells": 0, "pieces": 0, "rows": 0}


def count(_user, bp):
    b = bp.contents
    seen["cells"] += b.n_cells
    seen["pieces"] += 1
    seen["rows"] += b.row_end - b.row_begin
    return 0


cb = (_capi.ENCODED_ROWS_CB if encoded else _capi.ROW_BLOCK_CB)(count)
entry = ctx.lib.mvs_pairwise_stream_encoded if encoded else ctx.lib.mvs_pairwise_stream
res = []
for r in range(reps + 1):
    seen.update(cells=0, pieces=0, rows=0)
    cnt = ctypes.c_int64()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rc = entry(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, cb, None, ctypes.byref(cnt))
    wall = time.perf_counter() - t0
    assert rc == 0, ctx.lib.mvs_last_error()
    st = ctx.stream_stats()
    assert seen["cells"] == cnt.value and seen["rows"] == n
    if r:
        res.append((wall * 1e3, st))
    print("run %d: wall %.2f ms, kernels %.2f ms, %d cells, %.3f GB in %d pieces, %d row block(s), %s" %
          (r, wall * 1e3, st["kernel_ms"], cnt.value, st["bytes"] / 1e9, st["pieces"], st["row_blocks"],
           "two-stage" if st["two_stage"] else "exact kernel"), flush=True)
# the bare link: the same number of bytes, device -> pinned host, in pieces of 32 MiB
nbytes = max(res[-1][1]["bytes"], 1)
piece = min(nbytes, 32 << 20)
src = torch.empty(piece, dtype=torch.uint8, device="cuda")
dst = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2)]
link = []
for _ in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range((nbytes + piece - 1) // piece):
        dst[i & 1].copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    link.append((time.perf_counter() - t0) * 1e3)
wall = float(np.mean([w for w, _ in res]))
kern = float(np.mean([s["kernel_ms"] for _, s in res]))
out = {"n": n, "d": d, "cluster": cluster, "form": "encoded rows" if encoded else "CSR (col int32, q uint8)", "kept_cells": int(cnt.value), "density": cnt.value / float(n) / n,
       "wall_ms": wall, "kernels_ms": kern, "bytes": int(nbytes), "link_ms": min(link),
       "link_GBps": nbytes / (min(link) * 1e-3) / 1e9, "wall_over_max_kernel_link": wall / max(kern, min(link)),
       "cells_per_s": float(n) * n / (wall * 1e-3), "kept_cells_per_s": cnt.value / (wall * 1e-3),
       "link_share_of_wall": min(link) / wall, "row_blocks": res[-1][1]["row_blocks"], "two_stage": res[-1][1]["two_stage"]}
print(json.dumps(out))
