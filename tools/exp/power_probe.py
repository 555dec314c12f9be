#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi) while the filter pass runs back to back on real and on constant data.
   python tools/exp/power_probe.py N d"""
import os
import subprocess
import sys
import threading
import time

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


def smi():
    try:
        r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                           text=True, timeout=20)
        return r.stdout.strip()[:1500] or r.stderr.strip()[:300]
    except Exception as e:   # noqa: BLE001
        return "rocm-smi failed: %s" % e


def run(label, sk, scale, mode):
    ss = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2 * scale + 1.0).to("cuda")
    sset = ctx.sketch_set(sk)
    stop = [False]
    out = []

    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            out.append(smi())
            time.sleep(0.7)

    th = threading.Thread(target=sampler)
    with ctx.options(pairwise_filter=mode):
        ctx.pairwise_rows(sset, n2, cells_out=cells)
        torch.cuda.synchronize()
        th.start()
        t0 = time.time()
        ts = []
        while time.time() - t0 < 4.0:
            ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            ts.append(ctx.kernel_ms(1))
        stop[0] = True
        th.join()
    print("== %s: %d launches, kernels %.3f ms" % (label, len(ts), np.mean(ts)))
    for o in out[:3]:
        print(o)
    sset.close()


print("idle:", smi())
real = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
run("two-stage, synthesised sketches", real, 1.0, 2)
z = torch.zeros_like(real)
z[0, 0] = 300
run("two-stage, all zero", z, 1e6, 2)
run("exact kernel, synthesised sketches", real, 1.0, 0)
