#!/usr/bin/env python3
"""Socket power and shader clock (rocm-smi) while the projection kernel runs back to back: on the synthetic hash
lists and on a list of identical hashes (same instruction stream, operands that do not toggle)."""
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

S, NH, D = 10000, 50000, 2048
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
sk = torch.empty((S, D), dtype=torch.int32, device="cuda")
ss = torch.empty(S, dtype=torch.int64, device="cuda")


def smi():
    r = subprocess.run(["rocm-smi", "--showpower", "--showclocks", "--json"], capture_output=True, text=True, timeout=20)
    try:
        import json
        d = json.loads(r.stdout)["card0"]
        return "sclk %s  socket power %s W" % (d["sclk clock speed:"], d["Current Socket Graphics Package Power (W)"])
    except Exception:   # noqa: BLE001
        return r.stdout[:200] + r.stderr[:200]


def run(label, hashes, offsets):
    stop = [False]
    out = []

    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            out.append(smi())
            time.sleep(0.7)

    ctx.project_csr_stats(hashes, offsets, D, sk, ss)
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.time()
    ts = []
    while time.time() - t0 < 4.0:
        ctx.project_csr_stats(hashes, offsets, D, sk, ss)
        ts.append(ctx.kernel_ms(0))
    stop[0] = True
    th.join()
    print("== %s: %d launches, kernel %.3f ms" % (label, len(ts), np.mean(ts)))
    for o in out[:3]:
        print(o)


h, o = synth.make_csr_torch(S, NH, seed=1234, device="cuda")
run("synthetic hash lists", h, o)
run("all hashes equal (results are not sketches of sets; same instruction stream)", torch.full_like(h, 0x0123456789abcdef), o)
