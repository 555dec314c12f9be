#!/usr/bin/env python3
"""What limits the clock during the filter, the exact kernel and an all-zero run: everything the SMI tools of the box expose
(socket power, sclk / mclk, voltage, throttle / limit status, power cap) sampled while each kernel runs back to back.
   python tools/exp/r06_limiter_probe.py N d OUTDIR"""
import json
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

n, d, outdir = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3]
os.makedirs(outdir, exist_ok=True)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
ctx.set_timing(True)
cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device="cuda")

CMDS = {
    "rocm-smi": ["rocm-smi", "--showpower", "--showclocks", "--showtemp", "--showperflevel", "--showvoltage", "--showmaxpower",
                 "--showuse", "--json"],
    "amd-smi-metric": ["amd-smi", "metric", "-g", "0", "--power", "--clock", "--temperature", "--json"],
    "amd-smi-throttle": ["amd-smi", "metric", "-g", "0", "--throttle", "--json"],
}


def run_cmd(cmd):
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=30)
        return (r.stdout.strip() or r.stderr.strip())[:60000]
    except Exception as e:   # noqa: BLE001
        return "%s failed: %s" % (cmd[0], e)


def sample():
    return {k: run_cmd(v) for k, v in CMDS.items()}


def static_info():
    out = {}
    for name, cmd in (("amd-smi-static-limit", ["amd-smi", "static", "-g", "0", "--limit", "--json"]),
                      ("amd-smi-static-board", ["amd-smi", "static", "-g", "0", "--asic", "--vbios", "--json"]),
                      ("rocm-smi-powercap", ["rocm-smi", "--showmaxpower", "--showpowerplaytable" if False else "--showmaxpower", "--json"]),
                      ("rocm-smi-clk-levels", ["rocm-smi", "-s"])):
        out[name] = run_cmd(cmd)
    return out


def run(label, sk, scale, mode, seconds=6.0):
    ss = torch.empty(n, dtype=torch.int64, device="cuda")
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(np.sqrt(ss.cpu().numpy() / d) ** 2 * scale + 1.0).to("cuda")
    sset = ctx.sketch_set(sk)
    stop = [False]
    out = []

    def sampler():
        time.sleep(1.0)
        while not stop[0]:
            out.append(sample())
            time.sleep(0.5)

    th = threading.Thread(target=sampler)
    with ctx.options(pairwise_filter=mode):
        ctx.pairwise_rows(sset, n2, cells_out=cells)
        torch.cuda.synchronize()
        th.start()
        t0 = time.time()
        ts, tf = [], []
        while time.time() - t0 < seconds:
            ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            ts.append(ctx.kernel_ms(1))
            if mode == 2:
                tf.append(ctx.kernel_ms(2))
        stop[0] = True
        th.join()
    rec = {"label": label, "launches": len(ts), "kernels_ms": float(np.mean(ts)), "filter_ms": float(np.mean(tf)) if tf else None,
           "samples": out}
    with open(os.path.join(outdir, "%s.json" % label.replace(" ", "_").replace(",", "")), "w") as f:
        json.dump(rec, f, indent=1)
    print("== %s: %d launches, kernels %.3f ms%s, %d samples" % (label, len(ts), np.mean(ts),
          (", filter %.3f ms" % np.mean(tf)) if tf else "", len(out)), flush=True)
    sset.close()


with open(os.path.join(outdir, "static.json"), "w") as f:
    json.dump({"static": static_info(), "idle": sample()}, f, indent=1)
print("idle sampled", flush=True)
real = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
run("filter two-stage synthesised sketches", real, 1.0, 2)
z = torch.zeros_like(real)
z[0, 0] = 300
run("filter two-stage all zero", z, 1e6, 2)
run("exact kernel synthesised sketches", real, 1.0, 0)
run("exact kernel all zero", z, 1e6, 0)
