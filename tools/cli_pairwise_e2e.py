#!/usr/bin/env python3
"""End-to-end timing of `pairwise_comp_optimized` on a synthesised DB (one shard = all rows), with the per-stage
wall times the executable prints under MVS_STAGE_TIMING=1.   python tools/cli_pairwise_e2e.py [N] [d] [runs] [cluster]
cluster: related samples per cluster (default 16; N/10 keeps 10 % of all cells)"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

BIN = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
runs = int(sys.argv[3]) if len(sys.argv) > 3 else 2
cluster = int(sys.argv[4]) if len(sys.argv) > 4 else 16
w = tempfile.mkdtemp(prefix="mvs_pairwise_e2e_")
try:
    t0 = time.perf_counter()
    db = w + "/db/"
    os.makedirs(db)
    ctx = pkg.Context(0)
    sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda", cluster=cluster)
    ss = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    ctx.sumsq(sk, out=ss)
    torch.cuda.synchronize()
    norms = np.sqrt(ss.cpu().numpy().astype(np.float64) / d)
    with open(db + "vectors.bin", "wb") as f:
        for s0 in range(0, n, 100_000):
            sk[s0:s0 + 100_000].cpu().numpy().tofile(f)
    with open(db + "vector_norms.txt", "w") as f:
        f.write("".join("s%d %s\n" % (i, "%g" % x) for i, x in enumerate(norms)))
    open(db + "dimension.txt", "w").write("%d\n" % d)
    open(db + "dtype.txt", "w").write("int32\n")
    del sk, ss, ctx
    torch.cuda.empty_cache()
    print("DB written: %.1f s, vectors.bin %.2f GB" % (time.perf_counter() - t0, os.path.getsize(db + "vectors.bin") / 1e9), flush=True)
    for run in range(runs):
        out = "%s/idx%d" % (w, run)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(BIN, "pairwise_comp_optimized"), "--db", db, "--max_memory_gb", "12", "--num_threads",
                            "16", "--output_folder", out, "--num_shards", "1", "--shard_idx", "0"],
                           capture_output=True, text=True, env=dict(os.environ, MVS_STAGE_TIMING="1"))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr[-2000:]
        own = [l for l in r.stdout.split("\n") if l.startswith("Total computation time") or l.startswith("Jac space")]
        size = sum(os.path.getsize(os.path.join(out, "shard_0", f)) for f in os.listdir(os.path.join(out, "shard_0")))
        print("run %d: %.3f s wall; %s; shard files %.1f MB" % (run, dt, "; ".join(own), size / 1e6), flush=True)
        print("   " + " | ".join(l[8:] for l in r.stderr.split("\n") if l.startswith("[stage]")), flush=True)
        for l in r.stderr.split("\n"):
            if l.startswith("[stream]"):
                print("   " + l, flush=True)
        shutil.rmtree(out)
finally:
    shutil.rmtree(w, ignore_errors=True)
