#!/usr/bin/env python3
"""Stage times of ONE rank of a G-way strong-scaled pairwise step, measured on one GPU, + a link model -> predicted scaling.

    python tools/strong_model.py [N] [d] [--ranks 1,2,4,8] [--link-GBps 61,122] [--chunks 2]

No multi-GPU node is needed: the storage buffers of all G ranks (per-rank blocks padded to 256 rows: _capi.shard_layout) are
built on the one device, then rank 0's step runs exactly as parallel.ShardedComparison runs it -- re-code + derive its OWN
rows (k_recode_rows), its block plan of the symmetric schedule (diagonal block, then the peers' blocks in `chunks` launches:
mvs_plan_begin / _filter / _finish), kept cells routed and sorted -- with every byte the exchange would have delivered
already in place.  What is measured: the device time of every stage for the per-rank problem size (events on the stream),
and the host-visible step time without any exchange.  What is modelled: the exchange.  Per rank and peer the all-gathers move
P * 24 bytes (statistics + norms), P * d_pad bytes (coarse plane, in `chunks` pieces) and P * d_pad bytes (low limbs; the
receiver rebuilds the high limbs from them and the coarse plane: mvs_sketch_set_planes_from_wire, timed here; --no-wire: both limb planes),
P = padded rows per rank; on the fully connected xGMI mesh every peer's block arrives over its own link, so a gather takes
block bytes / link rate (+ a latency per collective).  The filter of the diagonal block starts at once; the peers' chunk c
can start when chunk c of the coarse plane has landed; the re-check needs the limb planes.  The model walks that timeline.
Link rates: MI355X xGMI is 153.6 GB/s per link bidirectional = 76.8 GB/s per direction (the task statement quotes ~153 GB/s
per link); 61 GB/s = 80 % of one direction (conservative), 122 GB/s = 80 % of 153 (optimistic).
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi, parallel, synth


def fast_norm_sq(sumsq, d):
    x = np.sqrt(sumsq.astype(np.float64) / d)
    return np.array([float("%g" % v) for v in x]) ** 2


STAGES = ("prepare_ms", "diag_filter_ms", "peer_filters_ms", "rebuild_ms", "finish_ms", "route_report_ms", "sort_ms")


def model_step(m, G, P, d_pad, plan_blocks, rate_GBps, latency_us, chunks, first, no_wire=False, clock=1.0):
    """One rank's step on the timeline of the exchange.  m: measured stage spans (STAGES), wall_ms (no events on the stream),
    foreign_cells.  rate_GBps per link and direction, latency_us per collective, clock: the shader clock relative to the
    measurement (0.95 = every compute span 1 / 0.95 longer: eight cards under load in one chassis).  Returns (step_ms, exposed)."""
    lat = latency_us * 1e-3
    small = lat + P * 24 / (rate_GBps * 1e6) if G > 1 else 0.0                       # ms
    bounds = parallel.chunk_bounds(P, chunks, first)
    scale = min(1.0, m["wall_ms"] / sum(m[k] for k in STAGES)) / clock
    sp = {k: m[k] * scale for k in STAGES}
    # the exchange starts when the own rows are ready
    t_comm, arrive = sp["prepare_ms"] + small, []
    for (c0, c1) in bounds:
        t_comm += (lat + (c1 - c0) * d_pad / (rate_GBps * 1e6)) if G > 1 else 0.0
        arrive.append(t_comm)
    # what the re-check waits for: the limb planes, or the low limbs (their rebuild is a stage of the compute stream)
    planes_at = t_comm + ((lat + P * (2 if no_wire else 1) * d_pad / (rate_GBps * 1e6)) if G > 1 else 0.0)
    t = sp["prepare_ms"] + sp["diag_filter_ms"]
    waited = 0.0
    for a, (c0, c1) in zip(arrive, bounds):
        if G > 1 and a > t:
            waited += a - t
            t = a
        t += sp["peer_filters_ms"] * (c1 - c0) / P if plan_blocks > 1 else 0.0      # a piece's share of the peers' filter time
    if G > 1 and planes_at > t:
        waited += planes_at - t
        t = planes_at
    t += sp["rebuild_ms"] + sp["finish_ms"] + sp["route_report_ms"]
    exch = (lat + (64 + 16 * m["foreign_cells"] * 1.25) / (rate_GBps * 1e6)) if G > 1 else 0.0   # the mirrored cells
    t += exch + sp["sort_ms"]
    t += max(0.0, m["wall_ms"] / clock - sum(sp.values()))                            # host gaps
    return t, waited + exch


def median(xs):
    xs = sorted(xs)
    return xs[len(xs) // 2]


def from_cpp(paths, args):
    """mvs_step_bench JSONs (csrc/host/mvs_step_bench.cpp: every rank of a G-way split timed alone, exchange bytes in place; several
    runs per G) -> per G: every rank's median over the runs, the slowest rank, the modelled step and speed-up"""
    runs = {}
    for pth in paths:
        with open(pth) as f:
            rec = json.load(f)
        runs.setdefault(rec["ranks"], []).append(rec)
    out = {"host": "csrc/host/mvs_step.hpp (C++), mvs_step_bench", "runs_per_G": {str(g): len(v) for g, v in runs.items()}, "ranks": {}}
    n, d = next(iter(runs.values()))[0]["n"], next(iter(runs.values()))[0]["d"]
    d_pad = (d + 127) // 128 * 128
    out["n"], out["d"] = n, d
    keys = ("wall_ms_median", "prepare_own_rows_ms", "diag_filter_ms", "peer_filters_ms", "finish_ms", "cells_route_exchange_sort_ms",
            "filter_ms", "recheck_ms", "flagged_tiles_ms", "foreign_cells", "filter_tiles", "candidates", "own_cells", "plan_blocks")
    base = None
    for G in sorted(runs):
        per_rank = {}
        for rec in runs[G]:
            for pr in rec["per_rank"]:
                per_rank.setdefault(pr["rank"], []).append(pr)
        ranks = {}
        for r, lst in sorted(per_rank.items()):
            ranks[r] = {k: median([x[k] for x in lst]) for k in keys if k in lst[0]}
            ranks[r]["wall_ms_runs"] = [x["wall_ms_median"] for x in lst]
            ranks[r]["P"] = lst[0]["rows_per_rank_padded"]
        slow = max(ranks, key=lambda r: ranks[r]["wall_ms_median"])
        if base is None:
            base = ranks[slow]["wall_ms_median"]
        rec_out = {"per_rank_wall_ms_median": {str(r): v["wall_ms_median"] for r, v in ranks.items()}, "slowest_rank": slow,
                   "slowest_rank_wall_ms_median": ranks[slow]["wall_ms_median"], "slowest_rank_wall_ms_runs": ranks[slow]["wall_ms_runs"],
                   "slowest_rank_stages": {k: v for k, v in ranks[slow].items() if k != "wall_ms_runs"}, "model": {}, "sensitivity": []}
        def m_of(v):
            return {"wall_ms": v["wall_ms_median"], "prepare_ms": v["prepare_own_rows_ms"], "diag_filter_ms": v.get("diag_filter_ms", 0.0),
                    "peer_filters_ms": v.get("peer_filters_ms", 0.0), "rebuild_ms": 0.0, "finish_ms": v.get("finish_ms", 0.0),
                    "route_report_ms": v["cells_route_exchange_sort_ms"], "sort_ms": 0.0, "foreign_cells": v["foreign_cells"]}
        for rate in [float(x) for x in args.link_GBps.split(",")]:
            # every rank walks its own timeline; the step ends when the slowest one does
            worst = max(model_step(m_of(v), G, v["P"], d_pad, v["plan_blocks"], rate, args.latency_us, args.chunks, args.first)[0]
                        for v in ranks.values())
            rec_out["model"]["%g GB/s per link and direction" % rate] = {"step_ms": worst, "speedup_vs_1gpu_measured": base / worst}
        for lat in (20.0, 50.0, 100.0):
            for rate in (45.0, 61.0):
                for clock in (1.0, 0.95):
                    worst = max(model_step(m_of(v), G, v["P"], d_pad, v["plan_blocks"], rate, lat, args.chunks, args.first, clock=clock)[0]
                                for v in ranks.values())
                    rec_out["sensitivity"].append({"latency_us": lat, "link_GBps": rate, "clock": clock, "step_ms": worst,
                                                   "speedup_vs_1gpu_measured": base / worst})
        out["ranks"][str(G)] = rec_out
        print("G=%d  slowest rank %d: wall %.3f ms (median of %d runs: %s); per rank %s" %
              (G, slow, ranks[slow]["wall_ms_median"], len(ranks[slow]["wall_ms_runs"]), " ".join("%.3f" % x for x in ranks[slow]["wall_ms_runs"]),
               " ".join("%.3f" % v["wall_ms_median"] for v in ranks.values())))
        for k, v in rec_out["model"].items():
            print("      %s: step %.3f ms -> %.2f x the measured 1-GPU step" % (k, v["step_ms"], v["speedup_vs_1gpu_measured"]))
        if G > 1:
            print("      sensitivity (speed-up): " + "  ".join("%gus/%gGB/s/clk%.2f=%.2f" % (x["latency_us"], x["link_GBps"], x["clock"],
                                                                                           x["speedup_vs_1gpu_measured"]) for x in rec_out["sensitivity"]))
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("n", nargs="?", type=int, default=100_000)
    ap.add_argument("d", nargs="?", type=int, default=2048)
    ap.add_argument("--ranks", default="1,2,4,8")
    ap.add_argument("--link-GBps", default="61,122")
    ap.add_argument("--latency-us", type=float, default=20.0, help="per collective")
    ap.add_argument("--chunks", type=int, default=2)
    ap.add_argument("--first", type=float, default=0.33, help="share of the first piece of the coarse plane (parallel.py: MVS_GATHER_FIRST)")
    ap.add_argument("--reps", type=int, default=8)
    ap.add_argument("--seed", type=int, default=2345)
    ap.add_argument("--no-speculate", action="store_true", help="every plan waits for its own counts (two host syncs per step)")
    ap.add_argument("--no-wire", action="store_true", help="limb planes on the wire (3 bytes per entry)")
    ap.add_argument("--wire", action="store_true", help="(default) low limbs on the wire, 2 bytes per entry; the plan rebuilds the rows it reads")
    ap.add_argument("--overlap", action="store_true", help="filter launches of a plan alternate between two streams (plan_overlap)")
    ap.add_argument("--report-spin", type=int, default=-1, help="option report_spin: microseconds mvs_cells_report polls before it blocks")
    ap.add_argument("--out", default="")
    ap.add_argument("--from-cpp", nargs="+", default=None,
                    help="mvs_step_bench JSON files (the C++ step, every rank, several runs): model only, nothing is measured here")
    ap.add_argument("--rank", type=int, default=0, help="the rank whose step is measured (clamped to G - 1)")
    args = ap.parse_args()
    if args.from_cpp:
        return from_cpp(args.from_cpp, args)
    dev = torch.device("cuda", 0)
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream())
    ctx.set_timing(True)
    if args.overlap:
        ctx.set_option("plan_overlap", 1)
    if args.report_spin >= 0:
        ctx.set_option("report_spin", args.report_spin)
    n, d = args.n, args.d
    sk = synth.make_sketches_torch(n, d, 50_000, seed=args.seed, device=dev)
    ss = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(fast_norm_sq(ss.cpu().numpy(), d)).to(dev)
    out = {"n": n, "d": d, "chunks": args.chunks, "ranks": {}}
    base_ms = None
    for G in [int(x) for x in args.ranks.split(",")]:
        rps, P = _capi.shard_layout(n, G)
        no_wire = args.no_wire                                     # (default: low limbs on the wire, as parallel.ShardedComparison)
        n_st = P * G
        n_alloc, d_pad, nbytes = ctx.limb_geometry(n_st, d, 2)
        planes = torch.zeros(nbytes, dtype=torch.int8, device=dev)
        coarse = torch.zeros(n_alloc * d_pad, dtype=torch.uint8, device=dev)
        stats = torch.zeros(n_alloc * 16, dtype=torch.uint8, device=dev)
        n2_st = torch.zeros(n_alloc, dtype=torch.float64, device=dev)
        sset = ctx.sketch_set_from_planes(planes, n_st, n_alloc, d, d_pad, 2)
        ctx.attach_derived(sset, coarse, stats)
        for r in range(G):                                      # what the exchange would have delivered
            b, e = parallel.shard_rows(n, G, r)
            ctx.recode_rows(sset, sk[b:e] if e > b else None, r * P, P)
            n2_st[r * P:r * P + (e - b)] = n2[b:e]
        rank = min(args.rank, G - 1)                               # whose step is measured (the C++ tool measures all of them)
        b0, e0 = parallel.shard_rows(n, G, rank)
        f0 = rank * P                                              # the rank's frame in storage rows
        cap = max(1 << 21, 40 * (e0 - b0) + (1 << 20))
        raw = torch.empty((2 * cap, 4), dtype=torch.int32, device=dev)
        own = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        outc = torch.empty((cap, 4), dtype=torch.int32, device=dev)
        d_own = torch.zeros(2 + (e0 - b0 + 2) // 2, dtype=torch.int64, device=dev)      # the shard's state block
        send = torch.zeros(_capi.CELLS_HEADER_BYTES + 16 * cap, dtype=torch.uint8, device=dev)
        plan = parallel.block_plan(G, rank, P)
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(10)]
        acc = []
        lo = torch.zeros(n_alloc * d_pad, dtype=torch.int8, device=dev)
        gops = parallel.GpuOps(ctx, dev)
        gops.wire_rows(planes, lo, d_pad, 0, n_st)                 # what the exchange of low limbs would have delivered
        class _NoEvent:
            def record(self):
                pass
        state = {"last_max_row": 1 << 30, "rep": 0}

        def one_step(ev):
            """rank 0's step as parallel.ShardedComparison runs it; ev: ten events recorded between the stages, or stand-ins that
            record nothing (an event between two kernels costs the stream ~6 us: the wall is measured without them)"""
            ev[0].record()
            ctx.recode_rows(sset, sk[b0:e0] if e0 > b0 else None, f0, P)      # own rows only
            ev[1].record()
            with ctx.options(plan_speculate=0 if args.no_speculate else 1):     # as parallel.GpuOps runs it
                ctx.plan_begin(sset, n2_st, f0, f0 + P, G > 1, raw)
            ctx.plan_filter(plan[:1])
            ev[2].record()
            if len(plan) > 1:
                ctx.plan_rows_ready(0, n_st)                    # as parallel.ShardedComparison: statistics + norms have arrived
            for (c0, c1) in parallel.chunk_bounds(P, args.chunks, args.first):
                blocks = parallel.clip_blocks(plan[1:], P, c0, c1)
                if blocks:
                    ctx.plan_filter(blocks)
            ev[3].record()
            if G > 1 and not no_wire:                           # the plan rebuilds the foreign rows its second half reads (inside finish)
                ctx.plan_wire(lo)
            ev[9].record()
            d_cnt = ctx.plan_finish()
            ev[4].record()
            ctx.cells_route(raw, d_cnt, P, rps, n, b0, e0, own, d_own, send, cap)
            ev[5].record()
            ahead = state["rep"] > 0 and state["last_max_row"] <= 64    # as parallel.ShardedComparison: the sort in front of the step's host sync
            if ahead:
                ctx.cells_sort_rows_ahead(own, b0, e0, d_own, outc)
            n_own, heads, max_row = ctx.cells_report(send, 1, cap, e0 - b0, d_own)
            if n_own and not (ahead and max_row <= 64):
                if max_row <= 64:
                    ctx.cells_sort_rows(own, n_own, b0, e0, d_own, outc)
                else:
                    ctx.cells_sort(own, n_own, outc)
            state["last_max_row"] = max_row
            state["rep"] += 1
            ev[6].record()
            return n_own, heads

        for rep in range(args.reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            n_own, heads = one_step(ev)
            torch.cuda.synchronize()
            wall = (time.perf_counter() - t0) * 1e3
            if rep >= 2:
                ps = ctx.plan_stats()
                acc.append({"wall_instrumented_ms": wall, "prepare_ms": ev[0].elapsed_time(ev[1]), "diag_filter_ms": ev[1].elapsed_time(ev[2]),
                            "peer_filters_ms": ev[2].elapsed_time(ev[3]), "finish_ms": ev[9].elapsed_time(ev[4]),
                            "route_report_ms": ev[4].elapsed_time(ev[5]), "sort_ms": ev[5].elapsed_time(ev[6]),      # route | sort + report
                            "filter_kernels_ms": ps["filter_ms"], "recheck_ms": ps["recheck_ms"], "tiles_ms": ps["tiles_ms"],
                            "filter_tiles": ps["filter_tiles"], "filter_launches": ps["filter_launches"], "candidates": ps["candidates"],
                            "flagged_tiles": ps["flagged_tiles"], "own_cells": int(n_own), "foreign_cells": int(heads[0][0]),
                            "rebuild_ms": ev[3].elapsed_time(ev[9]),
                            "speculated": float(ps["speculated"]), "stale": float(ps["stale"])})
        # the wall of the step as production runs it: no timing events in the library, none between the stages
        ctx.set_timing(False)
        none = [_NoEvent() for _ in range(10)]
        walls = []
        for rep in range(args.reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            one_step(none)
            torch.cuda.synchronize()
            if rep >= 2:
                walls.append((time.perf_counter() - t0) * 1e3)
        ctx.set_timing(True)
        for a in acc:
            a["wall_ms"] = float(np.mean(walls))
        m = {k: float(np.mean([a[k] for a in acc])) for k in acc[0]}
        m["rows_per_rank_padded"] = P
        m["wire"] = "coarse plane + limb planes" if no_wire else "coarse plane + low limbs (peers' limb planes rebuilt in front of the re-check)"
        m["plan_blocks"] = len(plan)
        if base_ms is None:
            base_ms = m["wall_ms"]
        # ---- the exchange, modelled ----
        models = {}
        for rate in [float(x) for x in args.link_GBps.split(",")]:
            lat = args.latency_us * 1e-3
            small = lat + P * 24 / (rate * 1e6) if G > 1 else 0.0                       # ms
            chunks = parallel.chunk_bounds(P, args.chunks, args.first)
            t_comm, arrive = m["prepare_ms"] + small, []
            for (c0, c1) in chunks:
                t_comm += (lat + (c1 - c0) * d_pad / (rate * 1e6)) if G > 1 else 0.0
                arrive.append(t_comm)
            # what the re-check waits for: the limb planes, or the low limbs (their rebuild is a stage of the compute stream)
            if no_wire:
                planes_at = t_comm + ((lat + P * 2 * d_pad / (rate * 1e6)) if G > 1 else 0.0)
            else:
                planes_at = t_comm + ((lat + P * d_pad / (rate * 1e6)) if G > 1 else 0.0)      # the low limbs
            # compute stream: prepare, diagonal filter, then the chunk launches (each waits for its chunk), finish waits for the planes.
            # The stage spans were measured with events between the stages (each costs the stream ~6 us); the step's wall without
            # them.  The walk uses the spans scaled to the uninstrumented wall -- compute reaches its wait points EARLIER than the
            # instrumented spans say, so waits come out longer, not shorter.
            stages = ("prepare_ms", "diag_filter_ms", "peer_filters_ms", "rebuild_ms", "finish_ms", "route_report_ms", "sort_ms")
            scale = min(1.0, m["wall_ms"] / sum(m[k] for k in stages))
            sp = {k: m[k] * scale for k in stages}
            t_comm_scaled_shift = m["prepare_ms"] - sp["prepare_ms"]         # the exchange starts when the own rows are ready
            t = sp["prepare_ms"] + sp["diag_filter_ms"]
            waited = 0.0
            for a, (c0, c1) in zip(arrive, chunks):
                a -= t_comm_scaled_shift
                if G > 1 and a > t:
                    waited += a - t
                    t = a
                t += sp["peer_filters_ms"] * (c1 - c0) / P if len(plan) > 1 else 0.0      # a piece's share of the peers' filter time
            if G > 1 and planes_at - t_comm_scaled_shift > t:
                waited += planes_at - t_comm_scaled_shift - t
                t = planes_at - t_comm_scaled_shift
            t += sp["rebuild_ms"] + sp["finish_ms"] + sp["route_report_ms"]
            exch = (lat + (64 + 16 * m["foreign_cells"] * 1.25) / (rate * 1e6)) if G > 1 else 0.0   # the mirrored cells
            t += exch + sp["sort_ms"]
            host_gap = max(0.0, m["wall_ms"] - sum(sp.values()))
            t += host_gap
            models["%g GB/s per link and direction" % rate] = {"step_ms": t, "exposed_exchange_ms": waited + exch,
                                                               "speedup_vs_1gpu_measured": base_ms / t}
        m["model"] = models
        out["ranks"][str(G)] = m
        print("G=%d  [%s]  P=%d  wall %.3f ms (no exchange, no events; %.3f with the events the stage spans come from)  prepare %.3f  diag %.3f  peers %.3f  finish %.3f  route %.3f  sort %.3f | "
              "filter kernels %.3f (%d tiles, %d launches)  re-check %.3f  tiles %.3f | (rebuild of the rows the plan reads: inside finish) %.3f" %
              (G, "3 B/entry on the wire" if no_wire else "2 B/entry on the wire", P, m["wall_ms"], m["wall_instrumented_ms"], m["prepare_ms"], m["diag_filter_ms"], m["peer_filters_ms"], m["finish_ms"], m["route_report_ms"],
               m["sort_ms"], m["filter_kernels_ms"], m["filter_tiles"], m["filter_launches"], m["recheck_ms"], m["tiles_ms"], m["rebuild_ms"]))
        for k, v in models.items():
            print("      %s: step %.3f ms (exchange exposed %.3f) -> %.2f x the measured 1-GPU step" %
                  (k, v["step_ms"], v["exposed_exchange_ms"], v["speedup_vs_1gpu_measured"]))
        sset.close()
        del planes, coarse, stats, n2_st, raw, own, outc, send, lo
        torch.cuda.empty_cache()
    if args.out:
        with open(args.out, "w") as f:
            json.dump(out, f, indent=1)
    ctx.close()


if __name__ == "__main__":
    main()
