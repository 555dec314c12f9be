#!/usr/bin/env python3
"""Summarise rocprofv3 output directories.

  python tools/pmc_summary.py <dir> [<dir> ...]              text: mean per (kernel, counter) of every
                                                              *_counter_collection.csv below the directories
  python tools/pmc_summary.py --stats <dir>                   text: per-kernel calls / total / average duration from
                                                              the *_kernel_trace.csv below <dir>
  python tools/pmc_summary.py --traffic out.json key=<dir>... JSON: HBM bytes per launch per kernel for bench.py
                                                              (FETCH_SIZE / WRITE_SIZE passes; key = workload name)

Kernel names are shortened to the part that tells the variants apart: the filter is `k_pairwise_mfma<1, false, 2, ...>`
(MODE 2), the exact two-limb kernel `k_pairwise_mfma16<0, ...>`.
"""
import collections
import csv
import glob
import hashlib
import json
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def short(name):
    name = re.sub(r"^void\s+", "", name)
    name = name.replace("(anonymous namespace)::", "").replace("mvs::", "")
    name = re.sub(r"\(.*$", "", name).strip()
    m = re.match(r"k_pairwise_pp<\s*(\d+)", name)
    if m:
        return {"2": "k_pairwise_pp_filter", "1": "k_pairwise_pp_dots", "0": "k_pairwise_pp_exact"}[m.group(1)]
    m = re.match(r"k_pairwise_mfma<\s*(\d+),\s*(?:false|true|0|1),\s*(\d+)", name)
    if m:
        return {"2": "k_pairwise_mfma_filter", "1": "k_pairwise_mfma_dots"}.get(m.group(2), "k_pairwise_mfma_exact_L" + m.group(1))
    m = re.match(r"k_pairwise_mfma16<\s*(\d+)", name)
    if m:
        return "k_pairwise_mfma16_exact" if m.group(1) == "0" else "k_pairwise_mfma16_dots"
    m = re.match(r"k_filter_pp<", name)
    if m:
        return "k_filter_pp"
    return re.sub(r"<.*$", "", name)


def counters(dirs):
    agg = collections.defaultdict(list)
    for d in dirs:
        for f in sorted(glob.glob(d + "/**/*_counter_collection.csv", recursive=True)):
            for r in csv.DictReader(open(f)):
                agg[(short(r["Kernel_Name"]), r["Counter_Name"])].append(float(r["Counter_Value"]))
    return agg


def kernel_stats(d):
    agg = collections.defaultdict(list)
    for f in sorted(glob.glob(d + "/**/*_kernel_trace.csv", recursive=True)):
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6)
    return agg


def source_sha():
    h = hashlib.sha256()
    base = os.path.join(ROOT, "metagenome_vector_sketches_amd", "csrc")
    for fn in ("mvs_project.hip", "mvs_pairwise.hip", "mvs_pairwise_dev.h", "mvs_recode.hip", "mvs_cells.hip", "mvs_internal.h"):
        with open(os.path.join(base, fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def main():
    args = sys.argv[1:]
    if args and args[0] == "--stats":
        for k, v in sorted(kernel_stats(args[1]).items(), key=lambda kv: -sum(kv[1])):
            print("%-34s calls=%-5d total_ms=%-10.3f avg_ms=%-10.4f min_ms=%.4f" % (k, len(v), sum(v), sum(v) / len(v), min(v)))
        return
    if args and args[0] == "--traffic":
        out = {"_provenance": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, --kernel-trace, csv), "
                              "means over the launches of each pass, 1 x MI355X.  FETCH_SIZE is in KiB and on gfx950 "
                              "counts 128-B requests at 64 B (MI355X_MICROARCH.md, HBM section): doubled.  "
                              "hbm_bytes_per_launch_corrected = 2 * 1024 * FETCH_SIZE + 1024 * WRITE_SIZE.",
               "kernel_source_sha": source_sha(), "collected": time.strftime("%Y-%m-%d"), "workloads": {}}
        for spec in args[2:]:
            key, d = spec.split("=", 1)
            cmd = ""
            if os.path.exists(os.path.join(d, "command.txt")):
                cmd = open(os.path.join(d, "command.txt")).read().strip()
            agg = counters([d])
            kernels = {}
            for (k, c), v in agg.items():
                if c in ("FETCH_SIZE", "WRITE_SIZE") and k.startswith("k_"):
                    kernels.setdefault(k, {})[c + "_KiB_raw"] = sum(v) / len(v)
                    kernels[k]["launches_" + c] = len(v)
            for (k, c), v in agg.items():        # instruction / conflict counts bench.py and the docs quote, same passes
                if c in ("SQ_INSTS_VALU", "SQ_INSTS_VALU_MFMA_I8", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_LDS") and k in kernels:
                    kernels[k][c + "_per_launch"] = sum(v) / len(v)
            for k, e in kernels.items():
                e["hbm_bytes_per_launch_corrected"] = 2048.0 * e.get("FETCH_SIZE_KiB_raw", 0.0) + 1024.0 * e.get("WRITE_SIZE_KiB_raw", 0.0)
            out["workloads"][key] = {"command": cmd, "kernels": kernels}
        with open(args[1], "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)
        return
    if args and args[0] == "--derived":
        # --derived <pmc dir> <stats dir>: per library kernel the quantities the roofline discussion uses
        agg = {kc: sum(v) / len(v) for kc, v in counters([args[1]]).items()}
        dur = {k: sum(v) / len(v) for k, v in kernel_stats(args[2]).items()}
        for k in sorted({k for k, _ in agg if k.startswith("k_")}):
            g = agg.get((k, "GRBM_GUI_ACTIVE"))
            if not g or k not in dur:
                continue
            cyc = g / 8.0                                     # the counter sums the 8 XCDs
            line = "%-26s avg %.4f ms (kernel-trace pass)  clock %.2f GHz" % (k, dur[k], cyc / (dur[k] * 1e-3) / 1e9)
            busy = agg.get((k, "SQ_VALU_MFMA_BUSY_CYCLES"), 0.0)
            if busy:
                line += "  MFMA busy %.3f of SIMD cycles (%.4g / (%.4g x 1024))" % (busy / (cyc * 1024.0), busy, cyc)
            hit, miss = agg.get((k, "TCC_HIT_sum")), agg.get((k, "TCC_MISS_sum"))
            if hit is not None and miss is not None and hit + miss > 0:
                line += "  L2 hit %.3f" % (hit / (hit + miss))
            f, w = agg.get((k, "FETCH_SIZE")), agg.get((k, "WRITE_SIZE"))
            if f is not None and w is not None:
                b = 2048.0 * f + 1024.0 * w
                line += "  HBM-side bytes %.4g (%.2f TB/s)" % (b, b / (dur[k] * 1e-3) / 1e12)
            wc, wa = agg.get((k, "SQ_WAVE_CYCLES")), agg.get((k, "SQ_WAIT_ANY"))
            if wc and wa is not None:
                line += "  waiting %.2f of wave cycles" % (wa / wc)
            print(line)
        return
    for (k, c), v in sorted(counters(args).items()):
        if k.startswith("k_"):
            print("%s %s n=%d mean=%.6g" % (k, c, len(v), sum(v) / len(v)))


if __name__ == "__main__":
    main()
