#!/usr/bin/env python3
"""Reproduce bench.py's slow 10 %-dense stream (stream leg, then clusters 16 / 1024 / 10000) and show where the pinned
buffers live (numa_maps) and which NUMA node the GPU hangs on."""
import os, sys, re, glob
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import bench
import metagenome_vector_sketches_amd as pkg
dev = torch.device("cuda", 0)
ctx = pkg.Context(0); ctx.set_stream(torch.cuda.current_stream()); ctx.set_timing(True)
order = sys.argv[1] if len(sys.argv) > 1 else "stream-first"
if order == "stream-first":
    bench.stream_leg(ctx, dev, 30000, 2048, 50000)
d = bench.density_leg(ctx, dev, 100000, 2048, 50000)
print(order, [(p["cluster"], round(p["wall_ms"], 1)) for p in d["points"]])
bus = torch.cuda.get_device_properties(0).pci_bus_id if hasattr(torch.cuda.get_device_properties(0), "pci_bus_id") else None
for f in glob.glob("/sys/class/drm/card*/device/numa_node"):
    print(f, open(f).read().strip())
big = []
for line in open("/proc/self/numa_maps"):
    m = re.findall(r"N(\d)=(\d+)", line)
    pages = sum(int(c) for _, c in m)
    if pages >= 4000 and ("anon" in line or "kfd" in line or "dev" in line):
        big.append((pages, line.split()[0], " ".join("N%s=%s" % x for x in m), line.split()[1] if len(line.split()) > 1 else ""))
for b in sorted(big, reverse=True)[:14]:
    print(b)
