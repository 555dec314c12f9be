#!/usr/bin/env python3
"""Why is the 10 %-dense encoded stream slower after bench.py's streamed_dense leg?  Same process, one context or two."""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import metagenome_vector_sketches_amd as pkg
mode = sys.argv[1] if len(sys.argv) > 1 else "same"
dev = torch.device("cuda", 0)
ctx = pkg.Context(0); ctx.set_stream(torch.cuda.current_stream()); ctx.set_timing(True)
if mode != "none":
    s = bench.stream_leg(ctx, dev, 30000, 2048, 50000)
    print("stream leg", round(s["wall_ms"], 1), round(s["encoded_rows"]["wall_ms"], 1))
ctx2 = ctx
if mode == "fresh":
    ctx2 = pkg.Context(0); ctx2.set_stream(torch.cuda.current_stream()); ctx2.set_timing(True)
d = bench.density_leg(ctx2, dev, 100000, 2048, 50000, clusters=(10000,))
print(mode, [(p["cluster"], round(p["wall_ms"], 1), round(p["kernels_ms"], 1)) for p in d["points"]])
