#!/usr/bin/env python3
"""A/B of two project_everything binaries on the same box: first-run `sketch` (parse) wall and stage times."""
import os, subprocess, sys, tempfile, shutil, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
BIN = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin")
order = sys.argv[1:] or []
w = tempfile.mkdtemp(prefix="mvs_ab_")
subprocess.run([os.path.join(BIN, "mvs_make_hashes"), w + "/h.txt", "10000", "50000", "1234"] + order, check=True)
for rep in range(2):
    for exe in ("project_everything_prearena", "project_everything"):
        if os.path.exists(w + "/h.txt.csr"):
            os.remove(w + "/h.txt.csr")
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(BIN, exe), "sketch", w + "/h.txt", w + "/db", "-d", "2048"], capture_output=True, text=True,
                           env=dict(os.environ, MVS_STAGE_TIMING="1"))
        dt = time.perf_counter() - t0
        st = [l[8:] for l in r.stderr.split("\n") if l.startswith("[stage]")]
        print("%-30s %.3f s wall | %s" % (exe, dt, st[0] if st else r.stderr[-200:]), flush=True)
        shutil.rmtree(w + "/db", ignore_errors=True)
shutil.rmtree(w, ignore_errors=True)
