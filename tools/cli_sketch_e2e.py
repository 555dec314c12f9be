#!/usr/bin/env python3
"""End-to-end timing of `project_everything sketch`: the first run parses the hash text (and leaves <file>.csr next to
it), the following runs map the binary cache.  python tools/cli_sketch_e2e.py [samples] [hashes] [dimension] [sorted]
"sorted": the text has every sample's hashes ascending (this repository's `convert`); default: generation order (the
reference's `convert` dumps an unordered_set)."""
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
h = int(sys.argv[2]) if len(sys.argv) > 2 else 50000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 2048
order = ["sorted"] if len(sys.argv) > 4 and sys.argv[4] == "sorted" else []
w = tempfile.mkdtemp(prefix="mvs_sketch_e2e_")
try:
    t0 = time.perf_counter()
    subprocess.run([os.path.join(BIN, "mvs_make_hashes"), w + "/h.txt", str(n), str(h), "1234"] + order, check=True)
    print("generate text: %.2f s, %d bytes" % (time.perf_counter() - t0, os.path.getsize(w + "/h.txt")), flush=True)
    digests = []
    for run in ("parse", "cache", "cache"):
        db = "%s/db_%s" % (w, run)
        t0 = time.perf_counter()
        r = subprocess.run([os.path.join(BIN, "project_everything"), "sketch", w + "/h.txt", db, "-d", str(d)],
                           capture_output=True, text=True, env=dict(os.environ, MVS_STAGE_TIMING="1"))
        dt = time.perf_counter() - t0
        assert r.returncode == 0, r.stderr
        own = [l for l in r.stdout.split("\n") if l.startswith("Time to compute")]
        dg = hashlib.sha256(open(db + "/vectors.bin", "rb").read()).hexdigest()[:16]
        digests.append(dg)
        print("sketch (%s): %.3f s wall; %s; vectors.bin %s" % (run, dt, own[0] if own else "?", dg), flush=True)
        print("   " + " | ".join(l[8:] for l in r.stderr.split("\n") if l.startswith("[stage]")), flush=True)
        shutil.rmtree(db)
    assert len(set(digests)) == 1, digests
    print("cache bytes", os.path.getsize(w + "/h.txt.csr"))
finally:
    shutil.rmtree(w, ignore_errors=True)
