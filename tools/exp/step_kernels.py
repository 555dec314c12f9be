#!/usr/bin/env python3
"""Every kernel of ONE steady step from a rocprofv3 --kernel-trace CSV, in start order: offset from the step's first
kernel, duration, the idle gap in front of it.   python tools/exp/step_kernels.py trace.csv [marker] [nth-from-last]
A step starts at a kernel whose name contains `marker` (default k_recode_rows) and ends in front of the next one."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
marker = sys.argv[2] if len(sys.argv) > 2 else "k_recode_rows"
back = int(sys.argv[3]) if len(sys.argv) > 3 else 2
ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
starts = [i for i, e in enumerate(ev) if marker in e[2]]
if len(starts) < back + 1 or back < 0:
    sys.exit("fewer than %d steps in the trace" % (back + 1))
lo = starts[-back - 1]
hi = starts[-back] if back > 0 else len(ev) - 1          # back = 0: from the last marker to the end of the trace
sel = ev[lo:hi]
t0 = sel[0][0]
end = t0
busy = 0
print("step of %d kernels, %.3f ms from its first kernel to the next step's first" % (len(sel), (ev[hi][0] - t0) / 1e6))
for s, e, n in sel:
    name = n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mvs::", "").split("(")[0][:44]
    gap = (s - end) / 1e3
    print("  %9.1f us  %8.1f us  gap %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, name))
    busy += max(0, e - max(s, end))
    end = max(end, e)
print("busy (union) %.3f ms, last kernel ends at %.3f ms, tail to the next step %.3f ms" %
      (busy / 1e6, (end - t0) / 1e6, (ev[hi][0] - end) / 1e6))
