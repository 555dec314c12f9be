#!/usr/bin/env python3
"""A DB folder (vectors.bin, vector_norms.txt, dimension.txt, dtype.txt -- what `project_everything sketch` writes,
src/project_everything.cpp:306-361) of the synthetic sketches bench.py's pairwise configs use: the executables and the C++
tools then work on exactly the sketches `bench.py --config 3|4|5` times (seed 2345 / 3456 / 4567).
    python tools/make_synth_db.py N d seed OUTDIR/ [hashes=50000] [cluster=16]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth


def main():
    n, d, seed, out = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    hashes = int(sys.argv[5]) if len(sys.argv) > 5 else 50_000
    cluster = int(sys.argv[6]) if len(sys.argv) > 6 else 16
    if not out.endswith("/"):
        out += "/"
    os.makedirs(out, exist_ok=True)
    dev = torch.device("cuda", 0)
    # slab by slab, as bench.py's strong_run synthesises a rank's rows: the same matrix whichever rank count asks
    sk = synth.make_sketches_torch_rows(n, d, hashes, seed=seed, device=dev, row_begin=0, row_end=n, cluster=cluster)
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream())
    ss = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.sumsq(sk, out=ss)
    norms = np.sqrt(ss.cpu().numpy().astype(np.float64) / d)
    sk.cpu().numpy().astype("<i4").tofile(out + "vectors.bin")
    with open(out + "vector_norms.txt", "w") as f:
        f.write("".join("s%d %s\n" % (i, "%g" % v) for i, v in enumerate(norms)))
    open(out + "dimension.txt", "w").write("%d\n" % d)
    open(out + "dtype.txt", "w").write("int32\n")
    ctx.close()
    print("wrote %d x %d sketches (seed %d) to %s" % (n, d, seed, out))


if __name__ == "__main__":
    main()
