#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sketch + pairwise hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

Workload (BASELINE.json configs[1]): per GPU 10 000 synthetic FracMinHash-like samples x 50 000 hashes,
d = 2048.  One "step" = one pass of the whole hot path over that batch with the hash lists already
resident in HBM:
    project (K1)  ->  sum of squares  ->  norms text round trip (host, 10k values)  ->
    limb split    ->  [N > 1: RCCL all-gather of limb-plane row blocks + norms]      ->
    all-vs-all comparison of this rank's rows against ALL columns (K2) -> kept cells sorted by (row, col)
Weak scaling: every rank brings its own 10k samples, so N ranks compare (N*10k)^2 cells in total.

Rank 0 prints ONE JSON line.  `value` = samples sketched-and-compared per second over the whole job.
The pairwise rate of the same step is reported beside it as cells_per_s (cells = ordered (i, j) pairs, full
N x N matrix, as the reference computes them).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
INT8_MFMA_PEAK_TOPS = 5000.0   # dense int8 MFMA = 2x bf16 = ~5 POP/s
VALU_INT_PEAK_TOPS = 39.3      # 256 CU x 64 lanes x 2.4 GHz int32 ops/s


def fast_norm_sq(sumsq, d):
    """(parsed '%g' text of sqrt(sumsq/d))^2 -- what pairwise_comp_optimized.cpp:893-901 builds from the
    vector_norms.txt that sketch() writes.  Vectorised 6-significant-digit decimal rounding; entries
    that sit within 1e-6 of a rounding tie go through the exact printf/strtod path."""
    x = np.sqrt(sumsq.astype(np.float64) / d)
    out = np.zeros_like(x)
    nz = x > 0
    e = np.floor(np.log10(x[nz])).astype(np.int64)
    scale = np.power(10.0, 5 - e)
    m = x[nz] * scale
    r = np.rint(m)
    bump = r >= 1e6          # 999999.6 -> 1000000: one more digit
    r = np.where(bump, r / 10.0, r)
    scale = np.where(bump, scale / 10.0, scale)
    y = r / scale
    tie = np.abs(np.abs(m - np.floor(m)) - 0.5) < 1e-6
    if tie.any() or (5 - e < 0).any() or (5 - e > 22).any():
        idx = np.nonzero(nz)[0]
        bad = tie | (5 - e < 0) | (5 - e > 22)
        for i in np.nonzero(bad)[0]:
            y[i] = float("%g" % x[idx[i]])
    out[nz] = y
    return out * out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10,
                    help="untimed steps; the clocks need ~10 steps (0.1 s) to settle after the set-up phase")
    ap.add_argument("--samples", type=int, default=10_000, help="samples per GPU")
    ap.add_argument("--hashes", type=int, default=50_000, help="hashes per sample")
    ap.add_argument("--dim", type=int, default=2048)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pairwise-extra", type=int, default=0,
                    help="also time a pairwise-only run on this many synthesised sketches (rank 0, N=1)")
    ap.add_argument("--cluster", type=int, default=16, help="related samples per cluster (256: the dense variant)")
    ap.add_argument("--lognormal-sigma", type=float, default=0.0,
                    help="> 0: ragged samples, sizes ~ lognormal(ln hashes, sigma) clipped to [100, 2e6] (SURVEY 8d)")
    ap.add_argument("--host-input", action="store_true",
                    help="also time the step with the hash lists handed over as host buffers (PCIe inclusive; "
                         "reported as pcie_inclusive, never as value)")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    import metagenome_vector_sketches_amd as pkg
    from metagenome_vector_sketches_amd import synth

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if rank == 0:
            print("WORLD_SIZE=%d but --gpus %d" % (world, args.gpus), file=sys.stderr)
        sys.exit(2)
    # rehearsal on a one-GPU box: MVS_BENCH_REHEARSAL=1 puts every rank on device 0 and uses gloo for the
    # collectives (RCCL refuses two ranks on one device); the driver's multi-GPU runs use nccl (= RCCL)
    rehearsal = os.environ.get("MVS_BENCH_REHEARSAL") == "1"
    dev_index = 0 if rehearsal else local_rank
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    S, NH, D = args.samples, args.hashes, args.dim
    ctx = pkg.Context(dev_index)
    stream = torch.cuda.current_stream()
    ctx.set_stream(stream)
    ctx.set_timing(True)

    # ---- synthetic input, resident in HBM ----
    if args.lognormal_sigma > 0:
        hashes, offsets = synth.make_csr_torch_ragged(S, NH, args.lognormal_sigma, seed=1234 + rank, device=dev,
                                                      cluster=args.cluster, shared=0.4)
    else:
        hashes, offsets = synth.make_csr_torch(S, NH, seed=1234 + rank, device=dev, cluster=args.cluster, shared=0.4)
    sketches = torch.empty((S, D), dtype=torch.int32, device=dev)
    sumsq = torch.empty(S, dtype=torch.int64, device=dev)
    N_total = S * world
    cap = max(1 << 20, max(64, 4 * args.cluster) * S)
    cells = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    from metagenome_vector_sketches_amd import parallel
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, dev), rank, world, dist if world > 1 else None)
    state = {}

    def step():
        # K1; the sums of squares and max |v| come out of the same kernel
        max_abs = ctx.project_csr_stats(hashes, offsets, D, sketches, sumsq)
        n2_local = fast_norm_sq(sumsq.cpu().numpy(), D)                    # text round trip of the norms
        # limb split, [all-gather of plane row blocks + norms], K2 on this rank's rows x all columns
        _, cnt, info = sc.run(sketches, n2_local, N_total, cells_out=cells, max_abs_local=max_abs)
        state["cnt"] = cnt
        state["candidates"] = ctx.pairwise_candidates()
        state["limbs"] = info["limbs"]
        state["schedule"] = info.get("schedule", "rows x all columns")

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    k1_ms, k2_ms = [], []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sc.ops.k2_ms = 0.0
        step()
        k1_ms.append(ctx.kernel_ms(0))
        # symmetric multi-rank schedule: several comparison launches per step (summed by GpuOps)
        k2_ms.append(sc.ops.k2_ms if sc.ops.k2_ms > 0 else ctx.kernel_ms(1))
    sync_all()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([state["cnt"]], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        kept_total = int(c.item())
    else:
        kept_total = state["cnt"]

    if rank != 0:
        if world > 1:
            dist.destroy_process_group()
        return

    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = N_total / (elapsed / args.steps)
    cells_per_step = float(N_total) * float(N_total)
    k1 = float(np.mean(k1_ms))
    k2 = float(np.mean(k2_ms))
    limbs = state["limbs"]

    # roofline of the dominant kernel (K1, projection): algorithmic bytes = 8*n_i + 4*d per sample
    total_hashes = float(offsets[S])                       # == S * NH unless --lognormal-sigma
    k1_bytes = 8.0 * total_hashes + 4.0 * D * S
    k1_gbs = k1_bytes / (k1 * 1e-3) / 1e9
    k1_intops = total_hashes * D             # sign accumulations
    k2_flops = 2.0 * D * S * N_total         # this rank's rows x all columns
    # HBM bytes per launch from the PMC passes committed under profiles/ (rocprofv3 cannot run inside
    # this process); only valid for the default workload on one GPU
    traffic = {"k_project": None, "k_pairwise_mfma": None}
    try:
        if (S, NH, D, world, args.cluster, args.lognormal_sigma) == (10_000, 50_000, 2048, 1, 16, 0.0):   # the profiled workload
            with open(os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")) as f:
                pmc = json.load(f)
            traffic = {k: pmc[k]["hbm_bytes_per_launch_corrected"] for k in traffic}
    except Exception:
        pass
    res = {
        "metric": "samples projected/sec + pairwise Jaccard cells/sec, d=2048, 1/2/4/8 GPUs",
        "value": samples_per_s,
        "unit": "samples/s (projected and compared all-vs-all, whole job)",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "u64 hash -> int32 sketch; int8 limbs x int32 MFMA accumulate; fp64 keep test",
        "data": "synthetic",
        "config": {"workload": "configs[1]: %d synthetic samples x %d hashes per GPU, d=%d, projection + "
                               "pairwise" % (S, NH, D),
                   "samples_per_gpu": S, "hashes_per_sample": NH, "d": D, "total_samples": N_total,
                   "cluster": args.cluster, "lognormal_sigma": args.lognormal_sigma,
                   "hashes_per_gpu": int(total_hashes),
                   "limbs": limbs, "kept_cells": kept_total, "parallelism": "row shards x%d" % world,
                   "schedule": state["schedule"]},
        "cells_per_s": cells_per_step / (elapsed / args.steps),
        "stages": {"projection_kernel_ms": k1, "projection_samples_per_s_per_gpu": S / (k1 * 1e-3),
                   "pairwise_kernel_ms": k2, "pairwise_cells_per_s_per_gpu": S * float(N_total) / (k2 * 1e-3),
                   "other_ms": ms_per_step - k1 - k2},
        "roofline": {"kernel": "k_project", "bound": "hbm", "achieved": k1_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS, "traffic": traffic["k_project"],
                     "algorithmic_bytes": k1_bytes,
                     "note": "integer-VALU bound by construction (implicit hash-generated matrix): see valu"},
        "roofline_valu": {"kernel": "k_project", "achieved": k1_intops / (k1 * 1e-3) / 1e12,
                          "peak": VALU_INT_PEAK_TOPS, "unit": "T sign-accumulations/s vs T int32-op/s",
                          "frac": k1_intops / (k1 * 1e-3) / 1e12 / VALU_INT_PEAK_TOPS,
                          # instruction-issue bound: 22.9 VALU per (hash, 64-dim block) at the issue costs measured by
                          # tools/microbench/valu_rates (profiles/r01_valu_rates_microbench.txt): 35.5 ns per
                          # (64 hashes x block) per SIMD, 1024 SIMDs
                          "issue_bound_ms": total_hashes * ((D + 63) // 64) / 64.0 / 1024.0 * 35.5e-6,
                          "issue_bound_frac": total_hashes * ((D + 63) // 64) / 64.0 / 1024.0 * 35.5e-6 / k1},
        "roofline_pairwise": {"kernel": "k_pairwise_mfma", "bound": "mfma",
                              "achieved": k2_flops / (k2 * 1e-3) / 1e12, "peak": INT8_MFMA_PEAK_TOPS,
                              "unit": "TFLOP/s", "frac": k2_flops / (k2 * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
                              # matrix-core work actually issued: passes per cell (1 limb: 1, Karatsuba: 3,
                              # two base-256 limbs: 4) x share of the tiles the symmetric schedule computes
                              # (the two-stage comparison issues one pass, on the coarse plane, and re-checks
                              # `candidates` pairs on the vector ALU)
                              "two_stage": state["candidates"] > 0, "candidates": state["candidates"],
                              "issued_frac": k2_flops * (1 if state["candidates"] > 0 else
                                                         {1: 1, 0x103: 3, 2: 4}.get(limbs, 0)) *
                              (0.5 + 0.5 * 128.0 / S if world == 1 or state["schedule"] == "symmetric"
                               else 1.0 - 0.5 / world + 0.5 * 128.0 / N_total)
                              / (k2 * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
                              "traffic": traffic["k_pairwise_mfma"]},
    }

    if args.pairwise_extra and world == 1:
        n = args.pairwise_extra
        sk = synth.make_sketches_torch(n, D, NH, seed=2345, device=dev)
        ss_dev = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.sumsq(sk, out=ss_dev)
        n2x = torch.from_numpy(fast_norm_sq(ss_dev.cpu().numpy(), D)).to(dev)
        sset = ctx.sketch_set(sk)
        cellsx = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device=dev)
        ctx.pairwise_rows(sset, n2x, cells_out=cellsx)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, cntx = ctx.pairwise_rows(sset, n2x, cells_out=cellsx)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t1
        kx = ctx.kernel_ms(1)
        res["pairwise_extra"] = {"workload": "pairwise only, %d synthesised sketches, d=%d" % (n, D),
                                 "seconds": dt, "kernel_ms": kx, "cells_per_s": float(n) * n / dt,
                                 "kept_cells": cntx, "limbs": sset.limbs,
                                 "algorithmic_tflops": 2.0 * D * n * n / (kx * 1e-3) / 1e12}
        sset.close()

    if args.host_input and world == 1:
        h_host = hashes.cpu().numpy().view(np.uint64)                                      # pageable, as a caller's vector would be
        o_host = np.asarray(offsets)
        sk_host = np.empty((S, D), dtype=np.int32)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.project_csr(h_host, o_host, D, out=sk_host)               # H2D hashes, K1, D2H sketches
            ts.append(time.perf_counter() - t1)
        res["pcie_inclusive"] = {"workload": "projection with host in/out buffers (pageable)",
                                 "seconds": min(ts), "samples_per_s": S / min(ts),
                                 "h2d_gb": h_host.nbytes / 1e9, "d2h_gb": sk_host.nbytes / 1e9}

    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(hashes, offsets, S, NH, D)

    print(json.dumps(res))
    if world > 1:
        dist.destroy_process_group()


def cpu_baseline(hashes, offsets, S, NH, D):
    """The oracle (CPU port of the reference path) timed on this host on a bounded sample of the same
    workload and extrapolated: projection is linear in samples, pairwise quadratic in N."""
    from oracle import pyoracle as orc
    cores = orc.max_threads()
    ns = 96
    o = np.asarray(offsets[:ns + 1])
    h = hashes[:int(o[ns])].cpu().numpy().view(np.uint64)
    per_sample = float(offsets[S]) / S / (float(o[ns]) / ns)       # ragged input: the sample's mean size vs the job's
    t0 = time.perf_counter()
    sk_lit = orc.project_csr(h[:int(o[32])], o[:33], D, threads=cores, native=True)     # literal loop nest
    t_lit = (time.perf_counter() - t0) / 32 * (float(o[ns]) / ns) / (float(o[32]) / 32)
    t0 = time.perf_counter()
    sk = orc.project_csr(h, o, D, threads=cores, fast=True, native=True)
    t_fast = (time.perf_counter() - t0) / ns
    assert np.array_equal(sk[:32], sk_lit)
    t_proj = min(t_lit, t_fast) * per_sample                     # seconds per sample of the job's mean size
    # pairwise: N = 2048 synthetic sketches of the same magnitude
    from metagenome_vector_sketches_amd import synth
    npw = 2048
    skp = synth.make_sketches_numpy(npw, D, NH, seed=2345)
    n2 = np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(r))) for r in skp])
    t0 = time.perf_counter()
    cells = orc.pairwise_rows(skp, n2, chunk=192, threads=cores, native=True)
    t_pw = time.perf_counter() - t0
    cells_per_s = npw * npw / t_pw
    t_job = S * t_proj + float(S) * S / cells_per_s
    return {"value": S / t_job, "unit": "samples/s (projected and compared all-vs-all)", "cores": cores,
            "kind": "port",
            "sample": "projection: %d samples x %d hashes (literal port %.1f samples/s, restructured port %.1f "
                      "samples/s, faster one used); pairwise: N=%d, d=%d, chunk 192 -> %.3g cells/s (%d kept); "
                      "extrapolated to %d samples (linear + quadratic)" %
                      (ns, NH, 1 / t_lit, 1 / t_fast, npw, D, cells_per_s, len(cells), S),
            "projection_samples_per_s": 1 / t_proj, "pairwise_cells_per_s": cells_per_s}


if __name__ == "__main__":
    main()
