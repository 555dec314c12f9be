"""round 5: k_recode_rows, 8 against 16 rows per workgroup (option recode_rows_wg), 100k x 2048 and 100k x 4096 / 12544 rows"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

dev = torch.device("cuda", 0)
ctx = pkg.Context(0)
ctx.set_stream(torch.cuda.current_stream())
for n, d in ((100096, 2048), (12544, 2048), (100096, 1024), (100096, 4096)):
    sk = synth.make_sketches_torch(n, d, 50_000, seed=7, device=dev)
    n_alloc, d_pad, nbytes = ctx.limb_geometry(n, d, 2)
    planes = torch.zeros(nbytes, dtype=torch.int8, device=dev)
    coarse = torch.zeros(n_alloc * d_pad, dtype=torch.uint8, device=dev)
    stats = torch.zeros(n_alloc * 16, dtype=torch.uint8, device=dev)
    sset = ctx.sketch_set_from_planes(planes, n, n_alloc, d, d_pad, 2)
    ctx.attach_derived(sset, coarse, stats)
    line = "n=%d d=%d:" % (n, d)
    for rep in range(2):
        for rw in (8, 16):
            ctx.set_option("recode_rows_wg", rw)
            for _ in range(3):
                ctx.recode_rows(sset, sk, 0, n)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ctx.recode_rows(sset, sk, 0, n)
            e1.record()
            torch.cuda.synchronize()
            line += "  rw%d %.4f ms" % (rw, e0.elapsed_time(e1) / 10)
    print(line, flush=True)
    sset.close()
    del sk, planes, coarse, stats
ctx.close()
