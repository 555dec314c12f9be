#!/usr/bin/env python3
"""profiles/rNN_rehearsal_timelines.txt from the rehearsal records of tools/exp/r05_final_records.sh:
    python tools/exp/rehearsal_timelines.py gpurun_out/r05_rehearsal_c3_r2.json gpurun_out/r05_rehearsal_c3_r4.json gpurun_out/r05_rehearsal_default_r2.json"""
import json
import sys


def last_line(path):
    return json.loads([x for x in open(path) if x.startswith("{")][-1])


def show(title, rec):
    cfg = rec["config"]
    print("== %s   (all ranks on ONE card, file transport)" % title)
    print("   kept cells %s  checksum %s  ms/step %.1f (ranks share the GPU: not a scaling number)  collectives %s; on the wire: %s" %
          (cfg.get("kept_cells"), cfg.get("cells_checksum"), rec["ms_per_step"], cfg.get("collectives"), cfg.get("wire")))
    print("   rank 0's last instrumented step, ms since 'step begin' (events on the compute stream / on the exchange's):")
    for what, at in rec.get("timeline", []):
        print("      %9.3f  %s" % (at, what))


for path in sys.argv[1:]:
    rec = last_line(path)
    g = rec["n_gpus"]
    if rec.get("scaling") == "strong":
        show("MVS_BENCH_REHEARSAL=1 python bench.py --gpus %d --config 3 --steps 3 --warmup 1" % g, rec)
    else:
        print("== MVS_BENCH_REHEARSAL=1 python bench.py --gpus %d --steps 5 --warmup 2 --no-cpu-baseline   (default line; its `strong` record)" % g)
        print("   configs[1] value %.0f samples/s over %d ranks sharing one card (not a scaling number); collectives %s" %
              (rec["value"], g, rec["config"].get("collectives")))
        for k, v in rec.get("strong", {}).items():
            if isinstance(v, dict):
                print("   %s: kept cells %s  checksum %s  ms/step %.1f  rccl_ranks %s" %
                      (k, v.get("kept_cells"), v.get("cells_checksum"), v.get("ms_per_step", float("nan")), v.get("rccl_ranks")))
            else:
                print("   strong.%s: %s" % (k, v))
