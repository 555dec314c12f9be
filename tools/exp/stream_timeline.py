#!/usr/bin/env python3
"""Merged timeline of kernels and device-to-host copies of the LAST mvs_pairwise_stream run in a rocprofv3 trace
(--kernel-trace --memory-copy-trace, CSV): per row block (a block starts at a k_dense_count launch) when its comparison
kernels ran, when its second half (count / fill / encode) ran, when its bytes crossed the link -- and how long the link
and the device idled.   python tools/exp/stream_timeline.py kernel_trace.csv memory_copy_trace.csv"""
import csv
import sys

kr = list(csv.DictReader(open(sys.argv[1])))
cr = list(csv.DictReader(open(sys.argv[2])))
def short(n):
    return n.replace("(anonymous namespace)::", "").replace("void ", "").replace("mvs::", "").split("(")[0].split("<")[0][:28]
K = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"])) for r in kr)
C = []
for r in cr:
    d = r.get("Direction", "")
    b = int(r.get("Bytes", r.get("Size", "0")) or 0)
    if "DEVICE_TO_HOST" in d.upper() or "D2H" in d.upper():
        C.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), b))
C.sort()
# the last run: from the last k_filter_meta (or the first kernel after a gap > 50 ms) to the end
starts = [i for i, k in enumerate(K) if "k_clear_tiles" in k[2] or "k_filter_meta" in k[2]]
run0 = 0
for i in range(1, len(K)):
    if K[i][0] - K[i - 1][1] > 30_000_000:
        run0 = i
K = K[run0:]
t0 = K[0][0]
tend = max(max(e for _, e, _ in K), max([e for s, e, _ in C if s >= t0] or [0]))
C = [c for c in C if c[0] >= t0]
big = [c for c in C if c[2] >= 1 << 16]
print("last run: %d kernels, %d D2H copies (%d of >= 64 KiB, %.1f MB), span %.2f ms" %
      (len(K), len(C), len(big), sum(c[2] for c in big) / 1e6, (tend - t0) / 1e6))
def union(iv):
    iv = sorted(iv)
    tot, cs, ce = 0, None, None
    for s, e in iv:
        if cs is None:
            cs, ce = s, e
        elif s > ce:
            tot += ce - cs
            cs, ce = s, e
        else:
            ce = max(ce, e)
    if cs is not None:
        tot += ce - cs
    return tot
kb = union([(s, e) for s, e, _ in K])
cb = union([(s, e) for s, e, _ in big])
print("device busy (union of kernels) %.2f ms; link busy (union of big copies) %.2f ms -> %.1f GB/s while busy; both idle %.2f ms" %
      (kb / 1e6, cb / 1e6, sum(c[2] for c in big) / max(cb, 1), (tend - t0 - union([(s, e) for s, e, _ in K] + [(s, e) for s, e, _ in big])) / 1e6))
# per block
blocks = [i for i, k in enumerate(K) if k[2].startswith("k_dense_count")]
print("%d row blocks; per block: [first comparison kernel .. k_dense_count) | second half [k_dense_count .. last k_enc_fill] | copies" % len(blocks))
prev_end_idx = 0
for bi, i in enumerate(blocks):
    nxt = blocks[bi + 1] if bi + 1 < len(blocks) else len(K)
    # second half: from k_dense_count to the last k_enc_fill / k_dense_fill before the next block's comparison
    j = i
    while j + 1 < nxt and K[j + 1][2].startswith(("k_dense", "k_enc", "k_active", "k_scan", "rocprim", "hipcub", "k_row")):
        j += 1
    cmp_k = K[prev_end_idx:i]
    sec = K[i:j + 1]
    cmp_s = (cmp_k[0][0] - t0) / 1e6 if cmp_k else float("nan")
    cmp_busy = union([(s, e) for s, e, _ in cmp_k]) / 1e6
    sec_busy = union([(s, e) for s, e, _ in sec]) / 1e6
    print("  block %2d: compare from %7.2f, busy %5.2f ms (%d k) | second half %7.2f .. %7.2f, busy %5.2f ms (%d k)" %
          (bi, cmp_s, cmp_busy, len(cmp_k), (sec[0][0] - t0) / 1e6, (sec[-1][1] - t0) / 1e6, sec_busy, len(sec)))
    prev_end_idx = j + 1
print("big copies (start ms, ms, MB, GB/s):")
for s, e, b in big[:80]:
    print("  %8.2f  %6.2f  %7.2f  %5.1f" % ((s - t0) / 1e6, (e - s) / 1e6, b / 1e6, b / max(e - s, 1)))
