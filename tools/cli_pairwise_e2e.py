#!/usr/bin/env python3
"""Synthesise a DB folder (vectors.bin, vector_norms.txt, dimension.txt) and run the pairwise executable on it
end to end, with its per-stage wall times.   python tools/cli_pairwise_e2e.py [N] [d] [workdir]"""
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 2048
work = sys.argv[3] if len(sys.argv) > 3 else "/tmp/mvs_e2e"
db = os.path.join(work, "db")
os.makedirs(db, exist_ok=True)
ctx = pkg.Context(0)
sk = synth.make_sketches_torch(n, d, 50_000, seed=2345, device="cuda")
ss = ctx.sumsq(sk.cpu().numpy())
sk.cpu().numpy().tofile(os.path.join(db, "vectors.bin"))
with open(os.path.join(db, "vector_norms.txt"), "w") as f:
    for i, v in enumerate(np.sqrt(ss / d)):
        f.write("s%d %g\n" % (i, v))
open(os.path.join(db, "dimension.txt"), "w").write("%d\n" % d)
ctx.close()
del sk
torch.cuda.empty_cache()
exe = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin", "pairwise_comp_optimized")
env = dict(os.environ, MVS_STAGE_TIMING="1")
for rep in range(2):
    t0 = time.time()
    r = subprocess.run([exe, "--db", db + "/", "--max_memory_gb", "12", "--num_threads", "8", "--output_folder",
                        os.path.join(work, "idx"), "--num_shards", "1", "--shard_idx", "0"],
                       capture_output=True, text=True, env=env)
    print("run %d: rc=%d wall %.2f s" % (rep, r.returncode, time.time() - t0))
    print(r.stdout.strip())
    print(r.stderr.strip())
