#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sketch + pairwise hot path on MI355X.

  python bench.py --gpus N --steps K --warmup W

N > 1 either way: under a launcher (`python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`: RANK /
WORLD_SIZE come from the environment) or plain (`python bench.py --gpus N ...`: this process starts N fresh workers
itself before it touches torch or HIP, relays rank 0's line and returns the worst exit code -- launch_workers()).

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
FP64_MATRIX_PEAK_TFLOPS = 78.6  # nominal fp64 matrix peak of the part (datasheet figure, not measured here)
VALU_INT_PEAK_TOPS = 78.6      # 256 CU x 4 SIMD-32 x 32 lanes x 2.4 GHz int32 lane-ops/s (MI355X_MICROARCH.md)
VALU_ISSUE_PEAK_GIPS = 1228.8  # wave64 VALU instructions/s: 1024 SIMDs x 2.4 GHz / 2 cycles per instruction
K1_VALU_PER_HASH_BLOCK = 22.4  # fallback only: VALU instructions per (hash, 64-dim block) and lane in k_project as counted in
                               # round 2 (SQ_INSTS_VALU 5.596e9 per launch = 22.4 x 1.6e10 / 64).  The figure reported is
                               # read from the newest profiles/rNN_pmc_traffic.json while its kernel_source_sha matches the
                               # sources this run was built from (k1_valu_per_hash_block()).


def source_sha():
    """identity of the kernel sources this run was built from: the PMC traffic figures under profiles/ carry the
    same hash and are only reported when it matches (a later kernel change must not inherit stale HBM bytes)"""
    import hashlib
    h = hashlib.sha256()
    base = os.path.join(ROOT, "metagenome_vector_sketches_amd", "csrc")
    for fn in ("mvs_project.hip", "mvs_pairwise.hip", "mvs_pairwise_dev.h", "mvs_recode.hip", "mvs_cells.hip", "mvs_internal.h"):
        with open(os.path.join(base, fn), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_file():
    """the newest profiles/rNN_pmc_traffic.json (by round number) -> (relative path, parsed) or (None, None)"""
    import glob
    import re
    best = None
    for path in glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")):
        m = re.match(r"r(\d+)_pmc_traffic\.json$", os.path.basename(path))
        if m and (best is None or int(m.group(1)) > best[0]):
            best = (int(m.group(1)), path)
    if best is None:
        return None, None
    try:
        with open(best[1]) as f:
            return "profiles/" + os.path.basename(best[1]), json.load(f)
    except Exception:   # noqa: BLE001
        return None, None


def pmc_traffic(workload_key):
    """HBM bytes per launch from the rocprofv3 PMC passes committed under profiles/ (rocprofv3 cannot run inside this
    process) -> ({kernel: bytes}, provenance).  Dropped (None) unless the file was collected on this workload AND on
    these kernel sources."""
    name, pmc = pmc_file()
    try:
        ent = pmc["workloads"][workload_key]
        prov = {"file": name, "kernel_source_sha": pmc.get("kernel_source_sha"),
                "collected": pmc.get("collected"), "command": ent.get("command")}
        if pmc.get("kernel_source_sha") != source_sha():
            prov["dropped"] = "kernel sources changed since the counters were collected (now %s)" % source_sha()
            return {}, prov
        return {k: v["hbm_bytes_per_launch_corrected"] for k, v in ent["kernels"].items()}, prov
    except Exception as e:   # no file yet / other workload
        return {}, {"file": None, "dropped": "no PMC pass for this workload (%s)" % type(e).__name__}


def k1_valu_per_hash_block(total_hashes, blocks):
    """VALU instructions per (hash, 64-dim block) and lane of k_project = SQ_INSTS_VALU per launch of the configs[1] PMC
    pass / (hashes x blocks / 64 lanes) -> (value, provenance).  Taken from the counter file only while it was collected
    on these kernel sources; otherwise the round-2 constant, and the provenance says so."""
    name, pmc = pmc_file()
    try:
        if pmc.get("kernel_source_sha") != source_sha():
            return K1_VALU_PER_HASH_BLOCK, "round-2 constant (kernel sources changed since %s was collected)" % name
        v = pmc["workloads"]["configs[1]"]["kernels"]["k_project"]["SQ_INSTS_VALU_per_launch"]
        return v / (total_hashes * blocks / 64.0), "%s: SQ_INSTS_VALU %.4g per launch" % (name, v)
    except Exception as e:   # noqa: BLE001
        return K1_VALU_PER_HASH_BLOCK, "round-2 constant (no counter file: %s)" % type(e).__name__


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


def launch_workers(n, argv, script=None, python=None, env=None, poll_s=0.05, grace_s=20.0, deadline_s=None):
    """`bench.py --gpus N` started WITHOUT a launcher (no RANK / WORLD_SIZE in the environment): this process becomes
    the parent of N fresh workers -- the same script with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
    set, one per GPU, which is the reference's own way of distributing (one process per shard,
    src/pairwise_comp_optimized.cpp:937-940).  The parent never imports torch and never touches HIP (a process that has
    initialised the GPU must not be replaced or forked); it relays rank 0's stdout (the ONE JSON line) and returns the
    worst child exit code.  When a worker dies, the others -- which would wait in the next collective for ever, RCCL has
    no timeout -- get `grace_s` seconds and are then terminated (by the exact PIDs started here)."""
    import socket
    import subprocess
    script = script or os.path.abspath(__file__)
    python = python or sys.executable
    base = dict(os.environ if env is None else env)
    base.setdefault("MASTER_ADDR", "127.0.0.1")
    if "MASTER_PORT" not in base:
        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
            sk.bind(("127.0.0.1", 0))
            base["MASTER_PORT"] = str(sk.getsockname()[1])
    base["WORLD_SIZE"] = base["LOCAL_WORLD_SIZE"] = str(n)
    base.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: RCCL between processes needs it on this driver
    procs = []
    for r in range(n):
        e = dict(base, RANK=str(r), LOCAL_RANK=str(r), GROUP_RANK="0")
        # rank 0's stdout is the bench line; whatever the other ranks print must not land next to it
        procs.append(subprocess.Popen([python, script] + list(argv), env=e,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr))
    out0 = []
    import threading
    reader = threading.Thread(target=lambda: out0.extend(procs[0].stdout.read().decode("utf-8", "replace").splitlines()))
    reader.start()
    first_fail = None
    if deadline_s is None:          # a collective that never completes (RCCL has no timeout) must not hold the caller for ever
        deadline_s = float(os.environ.get("MVS_BENCH_DEADLINE_S", "1500"))
    t_start = time.monotonic()
    while any(p.poll() is None for p in procs):
        codes = [p.poll() for p in procs]
        if first_fail is None and any(c not in (None, 0) for c in codes):
            first_fail = time.monotonic()
        if first_fail is None and time.monotonic() - t_start > deadline_s:
            print("bench.py: workers still running after %.0f s: terminating them" % deadline_s, file=sys.stderr)
            first_fail = time.monotonic() - grace_s
        if first_fail is not None and time.monotonic() - first_fail > grace_s:
            for p in procs:
                if p.poll() is None:
                    p.terminate()
            first_fail = time.monotonic() + 3600.0          # terminate once; kill below if that is ignored too
            deadline = time.monotonic() + 10.0
            while any(p.poll() is None for p in procs) and time.monotonic() < deadline:
                time.sleep(poll_s)
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(poll_s)
    reader.join()
    for line in out0:       # stdout carries the bench line and nothing else (gloo, for one, announces itself on stdout)
        print(line, flush=True, file=sys.stdout if line.lstrip().startswith("{") else sys.stderr)
    codes = [p.returncode for p in procs]
    worst = 124 if (time.monotonic() - t_start > deadline_s and all(c == 0 for c in codes)) else 0
    for c in codes:
        if c != 0:
            worst = max(worst, c if c > 0 else 128 - c)      # killed by signal s: 128 + s, as a shell reports it
    if worst:
        print("bench.py: worker exit codes %s" % codes, file=sys.stderr)
    return worst


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
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4, 5],
                    help="2 (default): configs[1], 10k samples x 50k hashes per GPU, projection + pairwise, weak scaling; "
                         "3 / 4 / 5: pairwise only on 100k x 2048 / 100k x 4096 / 1M x 2048 synthesised sketches split "
                         "over the ranks (strong scaling: BASELINE.json configs[2] / [3] / [4])")
    ap.add_argument("--pairwise-samples", type=int, default=100_000,
                    help="N=1: size of the configs[2] leg (pairwise only on synthesised sketches); 0 skips it")
    ap.add_argument("--pairwise-reps", type=int, default=10, help="timed repetitions of the configs[2] leg")
    ap.add_argument("--pairwise-dim", type=int, default=2048)
    ap.add_argument("--cluster", type=int, default=16, help="related samples per cluster (256: the dense variant)")
    ap.add_argument("--lognormal-sigma", type=float, default=0.0,
                    help="> 0: ragged samples, sizes ~ lognormal(ln hashes, sigma) clipped to [100, 2e6] (SURVEY 8d)")
    ap.add_argument("--stream-samples", type=int, default=30_000,
                    help="N=1: size of the streamed-output leg at the density of the reference's toy set (clusters of N/3 "
                         "samples: a third of all cells kept), mvs_pairwise_stream with a counting callback; 0 skips it")
    ap.add_argument("--density-samples", type=int, default=100_000,
                    help="N=1: size of the density leg (clusters of 16 / 1024 / 10000 related samples: sparse to 10 %% of the "
                         "cells kept, streamed as device-encoded rows); 0 skips it")
    ap.add_argument("--search-samples", type=int, default=500_000,
                    help="N=1: database size of the search leg (1 / 16 / 64 / 256 / 1024 query sketches against that many resident "
                         "sketches, mvs_search_block); 0 skips it")
    ap.add_argument("--strong-steps", type=int, default=10,
                    help="default mode: timed steps of the strong-scaled pairwise legs (configs[2] 100k x 2048 and configs[3] "
                         "100k x 4096 split over the N ranks, reported as `strong`); 0 skips them")
    ap.add_argument("--cpp-step-ranks", type=int, default=8,
                    help="N=1: `strong_cpp` record -- every rank's step of a split of configs[2] into this many ranks through the "
                         "C++ host of the step (csrc/host/mvs_step.hpp, bin/mvs_step_bench: the code pairwise_comp_optimized runs), "
                         "each rank timed alone with the exchange's bytes in place, beside the one-GPU step; 0 skips it")
    ap.add_argument("--strong-timeout", type=int, default=300,
                    help="N > 1: seconds the strong legs may take before every rank gives up on them (the line is printed "
                         "without them, `strong.error` says why)")
    ap.add_argument("--overlap-parts", type=int, default=2,
                    help="N > 1: pieces the rank's samples are projected in; the all-gather of a finished piece's limb "
                         "planes runs beside the projection of the next (1: no overlap)")
    ap.add_argument("--require-native-collectives", action="store_true",
                    help="N > 1: exit 3 on every rank if the library's own RCCL communicator (mvs_comm) cannot be created.  Default: "
                         "torch.distributed's nccl backend -- the same RCCL -- then carries the exchange, and the line says so in "
                         "config.collectives / collectives_note (a driver-launched scaling run must not lose its line to the set-up of a "
                         "node this build could never try).  Rehearsals (MVS_BENCH_REHEARSAL=1: all ranks on one card, file transport) "
                         "are not affected")
    ap.add_argument("--allow-torch-collectives", action="store_true", help="accepted for older scripts: this is the default now")
    ap.add_argument("--host-input", action="store_true",
                    help="also time the step with the hash lists handed over as host buffers (PCIe inclusive; "
                         "reported as pcie_inclusive, never as value)")
    args = ap.parse_args()

    if args.gpus > 1 and "RANK" not in os.environ and "WORLD_SIZE" not in os.environ:
        # plain `python bench.py --gpus N`: become the launcher (before torch / HIP are touched in this process)
        sys.exit(launch_workers(args.gpus, sys.argv[1:]))

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
    from metagenome_vector_sketches_amd import parallel, _capi
    coll, coll_note = None, None

    def shutdown():
        """collective teardown while every rank is still alive: the native communicator (ncclCommDestroy, or the file
        transport's last hand-shake) before the process group"""
        comm = getattr(coll, "comm", None)
        if comm is not None:
            comm.close()
        if world > 1:
            dist.destroy_process_group()
    if world > 1:
        # The data path's collectives go through the C ABI's communicator.  It gets a context of its own on a SIDE stream
        # so that the exchange of the rows a rank has finished can run beside the projection of the rest (parallel.py:
        # begin / feed / finish); ShardedComparison orders the two streams around every exchange.  RCCL: the 128-byte id
        # travels over torch.distributed; should creating the communicator fail on a node this build could never try,
        # torch.distributed's own collectives carry the exchange (on the same side stream) and the line says so.
        side = torch.cuda.Stream(device=dev)
        ctx_comm = pkg.Context(dev_index)
        ctx_comm.set_stream(side)

        def agree(ok):
            """True if every rank says ok (a collective over torch.distributed: every rank must call it the same number of times)"""
            t = torch.tensor([1 if ok else 0], device=dev if not rehearsal else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MIN)
            return int(t.item()) == 1
        # (1) what a rank can find out ALONE, agreed on before anything collective: does the library bind an RCCL here, and can
        # rank 0 draw the id?  Creating a communicator is itself a collective (ncclCommInitRank, the file transport's hand-shake): a
        # rank that gave up before it would leave the others waiting in it for ever -- RCCL has no timeout.
        uid_bytes = None
        try:
            if os.environ.get("MVS_BENCH_FAIL_NATIVE_COMM") == str(rank):      # test hook: this rank's library "finds no RCCL"
                raise RuntimeError("MVS_BENCH_FAIL_NATIVE_COMM")
            if not rehearsal:
                _capi.comm_library()
                if rank == 0:
                    uid_bytes = _capi.comm_unique_id()
        except Exception as e:      # noqa: BLE001 -- reported below, after the agreement every rank takes part in
            coll_note = "native communicator unavailable (%s: %s)" % (type(e).__name__, e)
        local_ok = coll_note is None
        all_ok = agree(local_ok)
        if all_ok:
            # (2) the collective creation.  An error return is reported by the library on every rank (bad id, mismatched world);
            # the agreement behind it covers the case where only some ranks see one
            try:
                if rehearsal:
                    comm = ctx_comm.comm_files(os.path.join(os.environ.get("TMPDIR", "/tmp"), "mvs_bench_%s" %
                                                            os.environ.get("MASTER_PORT", "0")), rank, world)
                else:
                    uid = torch.zeros(_capi.COMM_ID_BYTES, dtype=torch.uint8, device=dev)
                    if rank == 0:
                        uid.copy_(torch.frombuffer(bytearray(uid_bytes), dtype=torch.uint8))
                    dist.broadcast(uid, src=0)
                    comm = ctx_comm.comm_rccl(bytes(uid.cpu().numpy().tobytes()), rank, world)
                coll = parallel.NativeCollectives(comm, stream=side)
            except Exception as e:      # noqa: BLE001
                coll_note = "native communicator failed (%s: %s)" % (type(e).__name__, e)
                coll = None
            all_ok = agree(coll is not None)
        if not all_ok:   # all ranks use the same transport
            if coll is not None:
                coll.comm.close()
                coll = None
            if coll_note is None:
                coll_note = "another rank could not create the native communicator"
            if args.require_native_collectives:
                print("bench.py rank %d: %s -- not falling back (--require-native-collectives)" % (rank, coll_note), file=sys.stderr)
                dist.destroy_process_group()
                sys.exit(3)
            # never silently: the line names the communicator that carried the exchange (config.collectives) and why
            coll_note += "; torch.distributed's %s backend carries the exchange instead of the library's communicator" % (
                "gloo" if rehearsal else "nccl (RCCL)")
            print("bench.py rank %d: %s" % (rank, coll_note), file=sys.stderr, flush=True)
            coll = parallel.TorchCollectives(dist, rank, world, stream=side)

    if args.config != 2:
        res = strong_run(args, ctx, dev, rank, world, dist if world > 1 else None, coll, args.config, args.steps, args.warmup)
        res["config"]["collectives_note"] = coll_note
        if rank == 0:
            print(json.dumps(res))
        shutdown()
        return

    # ---- synthetic input, resident in HBM ----
    if args.lognormal_sigma > 0:
        hashes, offsets = synth.make_csr_torch_ragged(S, NH, args.lognormal_sigma, seed=1234 + rank, device=dev,
                                                      cluster=args.cluster, shared=0.4)
    else:
        hashes, offsets = synth.make_csr_torch(S, NH, seed=1234 + rank, device=dev, cluster=args.cluster, shared=0.4)
    sketches = torch.empty((S, D), dtype=torch.int32, device=dev)
    sumsq = torch.empty(S, dtype=torch.int64, device=dev)
    n2_local = torch.empty(S, dtype=torch.float64, device=dev)
    N_total = S * world
    cap = max(1 << 20, max(64, 4 * args.cluster) * S)
    cells = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    torch.cuda.synchronize()

    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, dev), rank, world, collectives=coll)
    sc.time_gather = world > 1
    state = {}

    o_host = np.ascontiguousarray(np.asarray(offsets), dtype=np.int64)
    # N > 1: the rank's samples are projected in `overlap_parts` pieces and the limb planes of a finished piece go into
    # the all-gather while the next piece is being projected; N = 1: one piece (nothing to overlap) -- the same step()
    n_parts = 1 if world == 1 else max(1, args.overlap_parts)
    bounds = sc.part_bounds(N_total, n_parts)

    def step():
        sc.begin(sketches, n2_local, N_total)
        k1 = 0.0
        for (p0, p1) in bounds:
            q0, q1 = min(p0, S), min(p1, S)
            m = 0
            if q1 > q0:
                # K1; the sums of squares and max |v| come out of the same kernel.  The library's events around K1 are recorded
                # in EVERY step (the roofline's kernel time is measured live over the timed region); the block plan's own five
                # events -- each costs the stream ~6 us -- only in the probe steps behind the timed region (state["probe"])
                ctx.set_timing(True)
                m = ctx.project_csr_stats(hashes, o_host[q0:q1 + 1], D, sketches[q0:q1], sumsq[q0:q1])
                k1 += ctx.kernel_ms(0)
                if not state.get("probe"):
                    ctx.set_timing(False)
                # text round trip of the norms (vector_norms.txt), on the device
                ctx.norms_sq_text(sumsq[q0:q1], D, out=n2_local[q0:q1])
            # limb split of these rows, [all-gather of exactly these rows of every rank's block, on the side stream]
            sc.feed(p0, p1, m)
        # [all-gather of the norms, streams joined], K2 on this rank's share of the block plan, kept cells sorted
        _, cnt, info = sc.finish(cells_out=cells)
        state["k1_ms"] = k1
        state["cnt"] = cnt
        state["limbs"] = info["limbs"]
        state["schedule"] = info.get("schedule", "rows x all columns")
        state["allgather_bytes_per_rank"] = info["allgather_bytes_per_rank"]
        state["overlap"] = info["overlap"]

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    k1_ms, k2_ms, gather_ms = [], [], []
    if os.environ.get("MVS_BENCH_PLAN_EVENTS") == "1":      # A/B: the plan's events in the timed steps too (as up to round 5's first session)
        state["probe"] = True
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
        k1_ms.append(state["k1_ms"])
        if world > 1:
            gather_ms.append(sc.last_gather_ms())    # events recorded before the comparison's end: already complete
    sync_all()
    elapsed = time.perf_counter() - t0
    # probe steps behind the timed region: the comparison kernels of the rank's block plan (filter launches + re-check + exact
    # kernel on flagged tiles) from the plan's own events, which the timed steps do without
    state["probe"] = True
    for _ in range(3):
        step()
        ps = ctx.plan_stats()
        k2_ms.append(ps["filter_ms"] + ps["recheck_ms"] + ps["tiles_ms"])
        state["plan"] = ps
    sync_all()
    ctx.set_timing(True)
    ps = state["plan"]
    state["candidates"] = ps["candidates"]
    state["filter_ms"] = ps["filter_ms"] if not ps["exact_mode"] else None
    state["filter_info"] = None if ps["exact_mode"] else (8, 256, ps["filter_tiles"], ps["d_pad"])
    # outside the timed region: the device's text round trip against the host's (printf / strtod semantics)
    if not np.array_equal(n2_local.cpu().numpy(), fast_norm_sq(sumsq.cpu().numpy(), D)):
        raise SystemExit("device norm text round trip differs from the host's")
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        c = torch.tensor([state["cnt"]], dtype=torch.int64, device=dev)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        kept_total = int(c.item())
    else:
        kept_total = state["cnt"]

    # ---- the strong-scaled pairwise step on every rank count: configs[2] and configs[3] split over the N ranks ----
    total_hashes = float(offsets[S])                       # == S * NH unless --lognormal-sigma

    def run_strong():
        strong = {}
        for cfg in (3, 4):
            torch.cuda.empty_cache()
            rec = strong_run(args, ctx, dev, rank, world, dist if world > 1 else None, coll, cfg, args.strong_steps, 3)
            strong["configs[%d]" % (cfg - 1)] = {"workload": rec["config"]["workload"], "cells_per_s": rec["value"],
                                                 "ms_per_step": rec["ms_per_step"], "kept_cells": rec["config"]["kept_cells"],
                                                 "cells_checksum": rec["config"]["cells_checksum"],
                                                 "schedule": rec["config"]["schedule"], "overlap": rec["config"]["overlap"],
                                                 "rccl_ranks": rec["config"].get("rccl_ranks", 0), "stages": rec["stages"],
                                                 "roofline": rec["roofline"]}
        return strong

    def run_strong_guarded(res):
        """N > 1: the strong legs behind a guard.  The headline of this line is complete when they start; a leg that raises
        or hangs on a node this build could never try (RCCL with more than one rank has not run anywhere yet) must not
        cost it.  A rank that raises reports and leaves; the ranks left waiting in a collective leave when the timer
        fires; rank 0 prints the line either way, with the reason in `strong.error`."""
        import threading
        done = threading.Event()

        def leave(why):
            if rank == 0:
                res["strong"] = {"error": why}
                print(json.dumps(res), flush=True)
            else:
                print("bench.py rank %d: strong legs: %s" % (rank, why), file=sys.stderr, flush=True)
            # no collective teardown: the peers may be gone or stuck.  Exit code: 0 by default -- the headline of a launcher-run
            # multi-GPU bench is complete and printed, an optional leg must not void it --, 3 with MVS_BENCH_STRICT=1 (CI that
            # wants a failed or hung strong leg to show in the status)
            os._exit(3 if os.environ.get("MVS_BENCH_STRICT") == "1" else 0)

        def fire():
            if not done.is_set():
                leave("the strong legs did not finish within %d s (--strong-timeout); headline unaffected" % args.strong_timeout)
        timer = threading.Timer(args.strong_timeout, fire)
        timer.daemon = True
        timer.start()
        try:
            out = run_strong()
        except BaseException as e:      # noqa: BLE001 -- reported in the line
            done.set()
            leave("%s: %s" % (type(e).__name__, e))
        done.set()
        timer.cancel()
        return out

    strong = {}
    if args.strong_steps > 0 and world > 1 and rank != 0:  # every rank takes part (rank 0: below, once its headline is assembled)
        del hashes, sketches, cells, sc
        run_strong_guarded(None)

    if rank != 0:
        shutdown()
        return

    ms_per_step = elapsed / args.steps * 1e3
    samples_per_s = N_total / (elapsed / args.steps)
    cells_per_step = float(N_total) * float(N_total)
    k1 = float(np.mean(k1_ms))
    k2 = float(np.mean(k2_ms))
    limbs = state["limbs"]

    # roofline of the dominant kernel (K1, projection): algorithmic bytes = 8*n_i + 4*d per sample
    k1_bytes = 8.0 * total_hashes + 4.0 * D * S
    k1_gbs = k1_bytes / (k1 * 1e-3) / 1e9
    k1_intops = total_hashes * D             # sign accumulations
    k2_flops = 2.0 * D * S * N_total         # this rank's rows x all columns
    default_workload = (S, NH, D, world, args.cluster, args.lognormal_sigma) == (10_000, 50_000, 2048, 1, 16, 0.0)
    traffic, traffic_src = pmc_traffic("configs[1]") if default_workload else ({}, {"file": None, "dropped": "non-default workload"})
    if default_workload:
        k1_vphb, k1_vphb_src = k1_valu_per_hash_block(total_hashes, (D + 63) // 64)
    else:
        k1_vphb, k1_vphb_src = K1_VALU_PER_HASH_BLOCK, "round-2 constant (non-default workload)"
    k1_valu_instr = total_hashes * ((D + 63) // 64) / 64.0 * k1_vphb     # wave64 VALU instructions per launch
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
                   "schedule": state["schedule"], "projection_parts": len(bounds), "overlap": state["overlap"],
                   "collectives": (coll.kind if coll is not None else "none"), "collectives_note": coll_note,
                   **comm_facts(coll, world)},
        "cells_per_s": cells_per_step / (elapsed / args.steps),
        "stages": {"projection_kernel_ms": k1, "projection_samples_per_s_per_gpu": S / (k1 * 1e-3),
                   "pairwise_kernel_ms": k2, "pairwise_cells_per_s_per_gpu": S * float(N_total) / (k2 * 1e-3),
                   "allgather_ms": float(np.mean(gather_ms)) if world > 1 else 0.0,
                   "allgather_bytes_per_rank": state.get("allgather_bytes_per_rank", 0),
                   "allgather_bytes_received_per_rank": state.get("allgather_bytes_per_rank", 0) * (world - 1),
                   "other_ms": ms_per_step - k1 - k2},
        # the dominant kernel of the step.  The contract prices it against HBM (algorithmic bytes = 8 n_i + 4 d per
        # sample); the +-1 matrix is generated from the hashes, so the kernel's binding bound is integer-VALU issue:
        # `binding` says so and roofline_valu_issue carries that fraction.
        "roofline": {"kernel": "k_project", "bound": "hbm", "achieved": k1_gbs, "peak": HBM_PEAK_GBS,
                     "unit": "GB/s", "frac": k1_gbs / HBM_PEAK_GBS, "traffic": traffic.get("k_project"),
                     "traffic_source": traffic_src, "algorithmic_bytes": k1_bytes,
                     "kernel_ms": k1, "binding": "valu_issue (see roofline_valu_issue): arithmetic intensity d/8 = "
                                                 "256 sign-accumulations per byte, nothing to stage from HBM"},
        "roofline_valu_issue": {"kernel": "k_project", "bound": "valu_issue",
                                "achieved": k1_valu_instr / (k1 * 1e-3) / 1e9, "peak": VALU_ISSUE_PEAK_GIPS,
                                "unit": "G wave64 VALU instructions/s",
                                "frac": k1_valu_instr / (k1 * 1e-3) / 1e9 / VALU_ISSUE_PEAK_GIPS,
                                "instructions_per_launch": k1_valu_instr,
                                "instructions_per_hash_block": k1_vphb, "instructions_source": k1_vphb_src,
                                "note": "%.1f VALU instructions per (hash, 64-dim block) x 2 cycles each on a SIMD-32 "
                                        "at 2.4 GHz nominal; the hash's 64-bit multiplies and shifts issue at 4 cycles, "
                                        "which is what keeps this fraction near one half" % k1_vphb,
                                # NOT a roofline fraction: bit-slicing does 64 sign-accumulations in ~4.4 instructions
                                "sign_accumulations_per_s_T": k1_intops / (k1 * 1e-3) / 1e12,
                                "int32_lane_op_peak_T": VALU_INT_PEAK_TOPS},
        "roofline_pairwise_step": step_pairwise_roofline(state, S, N_total, D, k2, k2_flops, traffic),
    }

    if args.strong_steps > 0 and world > 1:
        del sketches, cells, sc
        strong = run_strong_guarded(res)

    if args.pairwise_samples and world == 1:
        pw = pairwise_leg(ctx, dev, args.pairwise_samples, args.pairwise_dim, NH, args.pairwise_reps)
        res["config"]["workload_pairwise"] = pw.pop("workload")
        res["pairwise"] = pw.pop("leg")
        res["roofline_pairwise"] = pw.pop("roofline")

    if args.stream_samples and world == 1:
        res["streamed_dense"] = stream_leg(ctx, dev, args.stream_samples, args.pairwise_dim, NH)

    if args.density_samples and world == 1:
        res["density"] = density_leg(ctx, dev, args.density_samples, args.pairwise_dim, NH)
    if args.search_samples and world == 1:
        res["search"] = search_leg(ctx, dev, args.search_samples, args.pairwise_dim, NH)

    if args.host_input and world == 1:
        h_host = hashes.cpu().numpy().view(np.uint64)                                      # pageable, as a caller's vector would be
        o_host = np.asarray(offsets)
        sk_host = np.empty((S, D), dtype=np.int32)
        ts = []
        for _ in range(3):
            t1 = time.perf_counter()
            ctx.project_csr(h_host, o_host, D, out=sk_host)               # H2D hashes, K1, D2H sketches
            ts.append(time.perf_counter() - t1)
        # the bare link time for the same bytes: pinned host memory -> device and back, nothing else
        pin_in = torch.empty(h_host.shape, dtype=torch.int64).pin_memory()
        pin_out = torch.empty((S, D), dtype=torch.int32).pin_memory()
        dev_in = torch.empty(h_host.shape, dtype=torch.int64, device=dev)
        bare = []
        for _ in range(3):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            dev_in.copy_(pin_in, non_blocking=True)
            pin_out.copy_(sketches, non_blocking=True)
            torch.cuda.synchronize()
            bare.append(time.perf_counter() - t1)
        res["pcie_inclusive"] = {"workload": "projection with host in/out buffers (pageable)",
                                 "seconds": min(ts), "samples_per_s": S / min(ts),
                                 "h2d_gb": h_host.nbytes / 1e9, "d2h_gb": sk_host.nbytes / 1e9,
                                 "bare_link_seconds": min(bare), "ratio_to_bare_link": min(ts) / min(bare)}

    # (the GPU legs first: the CPU baseline's thread teams keep the host cores busy for a while after they return, and a
    # strong-scaled step has two host round trips in it -- measured once behind the baseline: 11.05 ms instead of 10.25)
    if args.strong_steps > 0 and world == 1:
        del sketches, cells, sc
        strong = run_strong()
    if strong:
        res["strong"] = strong
    if args.cpp_step_ranks > 1 and args.strong_steps > 0 and world == 1:
        res["strong_cpp"] = cpp_step_leg(dev, args.cpp_step_ranks, args.hashes)

    if not args.no_cpu_baseline and world == 1:
        res["cpu_baseline"] = cpu_baseline(hashes, offsets, S, NH, D, dev)
        if "pairwise" in res:
            res["pairwise"]["vs_cpu_port_all_cores"] = res["pairwise"]["cells_per_s"] / res["cpu_baseline"]["pairwise_cells_per_s"]
    print(json.dumps(res))
    shutdown()


def cpp_step_leg(dev, ranks, nh):
    """The strong-scaled step through its C++ host (csrc/host/mvs_step.hpp: what `pairwise_comp_optimized` runs with
    MVS_COLLECTIVE=rccl or --shard_idx -1; src/pairwise_comp_optimized.cpp:937-982 is what it replaces): configs[2]'s sketches are
    written as a DB folder, `bin/mvs_step_bench` times the one-GPU step and EVERY rank's step of a `ranks`-way split alone on this
    card, every byte of the exchange in place (40 warm-up + 40 timed steps per rank, no events on the stream).  The exchange itself
    is not in these numbers (tools/strong_model.py --from-cpp models it); `speedup_before_exchange` = one-GPU step / slowest rank."""
    import shutil
    import subprocess
    import tempfile
    import torch
    from metagenome_vector_sketches_amd import synth
    exe = os.path.join(ROOT, "metagenome_vector_sketches_amd", "bin", "mvs_step_bench")
    if not os.path.exists(exe):
        return {"error": "bin/mvs_step_bench is not built"}
    n, d, seed = STRONG[3]
    tmp = tempfile.mkdtemp(prefix="mvs_bench_db_")
    try:
        db = tmp + "/"
        sk = synth.make_sketches_torch_rows(n, d, nh, seed=seed, device=dev, row_begin=0, row_end=n)
        ss = (sk.to(torch.int64) ** 2).sum(dim=1)
        norms = np.sqrt(ss.cpu().numpy().astype(np.float64) / d)
        sk.cpu().numpy().astype("<i4").tofile(db + "vectors.bin")
        del sk
        torch.cuda.empty_cache()
        with open(db + "vector_norms.txt", "w") as f:
            f.write("".join("s%d %s\n" % (i, "%g" % v) for i, v in enumerate(norms)))
        with open(db + "dimension.txt", "w") as f:
            f.write("%d\n" % d)
        with open(db + "dtype.txt", "w") as f:
            f.write("int32\n")
        out = {}
        for g in (1, ranks):
            r = subprocess.run([exe, "--db", db, "--ranks", str(g), "--steps", "40", "--warmup", "40"], capture_output=True, text=True,
                               timeout=240)
            if r.returncode != 0:
                return {"error": "mvs_step_bench --ranks %d: rc %d: %s" % (g, r.returncode, r.stderr.strip()[-300:])}
            out[g] = json.loads(r.stdout.strip().split("\n")[-1])
        one, split = out[1]["per_rank"][0], out[ranks]
        slow = max(split["per_rank"], key=lambda p: p["wall_ms_median"])
        return {"workload": "configs[2]: %d synthetic samples, d=%d, the step of `pairwise_comp_optimized` (C++ host), 1 rank and every "
                            "rank of a %d-way split timed alone on one GPU, exchange bytes in place" % (n, d, ranks),
                "one_gpu_step_ms": one["wall_ms_median"], "ranks": ranks,
                "per_rank_step_ms": [p["wall_ms_median"] for p in split["per_rank"]],
                "slowest_rank": slow["rank"], "slowest_rank_step_ms": slow["wall_ms_median"],
                "speedup_before_exchange": one["wall_ms_median"] / slow["wall_ms_median"],
                "slowest_rank_stages": {k: slow[k] for k in ("prepare_own_rows_ms", "diag_filter_ms", "peer_filters_ms", "finish_ms",
                                                             "cells_route_exchange_sort_ms", "filter_ms", "recheck_ms", "flagged_tiles_ms",
                                                             "filter_launches", "filter_tiles", "candidates", "flagged_tiles",
                                                             "own_cells", "foreign_cells") if k in slow},
                "kept_cells_one_gpu": one["own_cells"],
                "note": "not a multi-GPU measurement: per-rank compute of the C++ step on ONE card; the exchange is modelled in "
                        "tools/strong_model.py --from-cpp (DESIGN.md section 7: 6.3-6.6 x at 61 GB/s per link and 20 us per collective)"}
    except Exception as e:      # noqa: BLE001 -- a leg beside the headline: reported, never fatal
        return {"error": "%s: %s" % (type(e).__name__, e)}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def step_pairwise_roofline(state, S, N_total, D, k2_ms, k2_flops, traffic):
    """MFMA utilisation of the filter kernel inside the configs[1] step: int8 operations issued (the tiles the launch
    computed, in the kernel's own tile size, x one pass of 2 * edge^2 * d_pad) / the filter kernel's time / 5 POP/s.
    The 2 d flop per cell of the rank's rows x all columns over ALL comparison kernels stays beside it as
    `algorithmic_credit` (not a utilisation)."""
    fi = state.get("filter_info")          # (variant, tile edge, tiles, d_pad) of the step's last filter launch, or None
    credit = {"flops": k2_flops, "kernels_ms": k2_ms, "tflops": k2_flops / (k2_ms * 1e-3) / 1e12,
              "ratio_to_peak": k2_flops / (k2_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
              "note": "2 d flop per cell of this rank's rows x all columns over filter + re-check; not a utilisation"}
    rec = {"bound": "mfma", "workload": "the %d x %d comparison inside the step" % (S, N_total),
           "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8 operations issued to the matrix cores)",
           "two_stage": state["candidates"] > 0, "candidates": state["candidates"], "algorithmic_credit": credit,
           "traffic": traffic.get("k_pairwise_pp_filter", traffic.get("k_pairwise_mfma_filter"))}
    f_ms = state.get("filter_ms")
    if not fi or not f_ms:
        rec.update(kernel="exact kernel on every cell (no filter pass in this step)", achieved=None, frac=None)
        return rec
    variant, edge, tiles, d_pad = fi
    issued = tiles * 2.0 * edge * edge * d_pad
    rec.update(kernel="%s, %d x %d tiles" % ("k_pairwise_pp<filter>" if variant in (7, 8, 9, 10, 40, 41, 42) else
                                             "k_search_filter" if variant == 50 else "k_pairwise_mfma<filter> (ring)", edge, edge),
               achieved=issued / (f_ms * 1e-3) / 1e12, frac=issued / (f_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
               issued_ops=issued, tiles=tiles, kernel_ms=f_ms)
    return rec


def comm_facts(coll, world):
    """what mvs_comm_info says about the communicator that carried the exchange: `rccl_ranks` answers "did RCCL see N
    ranks" (0: the exchange did not go through the library's RCCL communicator -- one rank, the file transport of a
    rehearsal, or torch.distributed's collectives after a fallback)"""
    comm = getattr(coll, "comm", None)
    if comm is None:
        return {"rccl_ranks": 0, "comm_world": world if coll is not None else 1}
    facts = {"rccl_ranks": comm.world if comm.is_rccl else 0, "comm_world": comm.world, "comm_rank0": comm.rank}
    if comm.is_rccl:           # which RCCL: the file the library's dlopen bound (mvs_comm_library) and its version
        try:
            from metagenome_vector_sketches_amd import _capi
            facts["rccl_library"], facts["rccl_version"] = _capi.comm_library()
        except Exception as e:      # noqa: BLE001
            facts["rccl_library"] = "unknown (%s)" % e
    return facts


STRONG = {3: (100_000, 2048, 2345), 4: (100_000, 4096, 3456), 5: (1_000_000, 2048, 4567)}   # --config -> (N, d, seed)


def strong_run(args, ctx, dev, rank, world, dist, coll, config, steps, warmup):
    """BASELINE.json configs[2] / [3] / [4] -- pairwise only, a FIXED number of synthesised sketches split over the ranks by
    the reference's shard formula (src/pairwise_comp_optimized.cpp:937-940).  One step = parallel.ShardedComparison.run():
    re-code this rank's rows (limb planes + the filter's coarse plane and row statistics, OWN rows only) -> all-gather of
    statistics + norms, coarse plane in chunks, limb planes (on the communicator's stream) -> this rank's block plan of the
    symmetric schedule (the diagonal block's filter starts at once, the peers' blocks as their chunks land) -> re-check,
    flagged tiles -> kept cells routed, the mirror images exchanged, collected, sorted.  Sketches and norms are resident in HBM
    when the clock starts.  Returns the rank-0 record (every rank takes part in its reductions)."""
    import torch
    from metagenome_vector_sketches_amd import parallel, synth
    n_total, d, seed = STRONG[config]
    rb, re = parallel.shard_rows(n_total, world, rank)
    sk = synth.make_sketches_torch_rows(n_total, d, args.hashes, seed=seed, device=dev, row_begin=rb, row_end=re)
    ss = torch.empty(re - rb, dtype=torch.int64, device=dev)
    _, max_abs = ctx.stats(sk, out=ss)
    n2_local = torch.from_numpy(fast_norm_sq(ss.cpu().numpy(), d)).to(dev)
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, dev), rank, world, collectives=coll)
    sc.time_gather = True
    sc.trace = []
    cap = max(1 << 21, 40 * (re - rb) + (1 << 20))
    cells = torch.empty((cap, 4), dtype=torch.int32, device=dev)
    state = {}

    def step():
        _, cnt, info = sc.run(sk, n2_local, n_total, cells_out=cells, max_abs_local=max_abs)
        state.update(cnt=cnt, info=info)

    def sync_all():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    # The K timed steps run WITHOUT instrumentation: every event recorded between two kernels costs the stream ~6 us
    # (profiles/r05_g8_step_kernels.txt: 14 records = 82 us of the 1.7 ms a rank of an 8-way split spends on its step).
    # The stage breakdown comes from `probe` instrumented steps behind the timed region.
    timing_was = True
    ctx.set_timing(False)
    sc.time_gather, sc.trace = False, None
    for _ in range(2):
        step()
    sync_all()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync_all()
    elapsed = time.perf_counter() - t0
    ctx.set_timing(timing_was)
    sc.time_gather, sc.trace = True, []
    acc = {k: [] for k in ("filter", "recheck", "tiles", "prepare", "plan_span", "cells", "gather")}
    probe = max(2, min(5, steps))
    sync_all()
    t1 = time.perf_counter()
    for _ in range(probe):
        sc.trace = []
        step()
        ps = ctx.plan_stats()                          # waits for the plan's last kernel, not for the sort behind it
        acc["filter"].append(ps["filter_ms"])
        acc["recheck"].append(ps["recheck_ms"])
        acc["tiles"].append(ps["tiles_ms"])
        acc["gather"].append(sc.last_gather_ms())
        state["plan"] = ps
    sync_all()
    instrumented_ms = (time.perf_counter() - t1) / probe * 1e3
    if os.environ.get("MVS_BENCH_STEP_TIMES") and rank == 0:       # per-step kernel times of the leg (diagnosis)
        print("strong_run config %d: per-step ms %s" % (config, {k: [round(x, 3) for x in v] for k, v in acc.items()}), file=sys.stderr)
    # stage spans of the LAST step from the events the step left on its streams
    tr = dict(sc.trace)
    def span(a, b):
        return tr[a].elapsed_time(tr[b]) if a in tr and b in tr else 0.0
    ready = [k for k in tr if k.startswith("own rows")][-1]
    prepare_ms = span("step begin", ready)
    plan_span_ms = span("plan begin", "plan finished")
    cells_ms = span("plan finished", "cells sorted")
    kept = state["cnt"]
    # an order-independent checksum of the shard (sum over its cells of a 64-bit mix of row, col, dot, q; wraps mod 2^64), summed
    # over the ranks: the same whatever the rank count, since the union of the shards is the same matrix
    cc = cells[:kept].to(torch.int64)
    mix = (cc[:, 0] * 1000003 + cc[:, 1]) * 2654435761 + cc[:, 2] * 40503 + cc[:, 3]
    digest = torch.stack([mix.sum(), (mix * mix).sum()])
    timeline = [[k, round(tr["step begin"].elapsed_time(ev), 4)] for k, ev in sc.trace] if "step begin" in tr else []
    k2 = float(np.mean(acc["filter"]) + np.mean(acc["recheck"]) + np.mean(acc["tiles"]))
    vals = [elapsed, k2, float(np.mean(acc["gather"])), prepare_ms, plan_span_ms, cells_ms, float(np.mean(acc["filter"]))]
    if world > 1:
        t = torch.tensor(vals, dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        vals = [float(x) for x in t.tolist()]
        c = torch.cat([torch.tensor([kept], dtype=torch.int64, device=dev), digest])
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        kept = int(c[0].item())
        digest = c[1:]
    digest = "%016x%016x" % (int(digest[0].item()) & (2 ** 64 - 1), int(digest[1].item()) & (2 ** 64 - 1))
    elapsed, k2_max, gather_max, prepare_ms, plan_span_ms, cells_ms, filter_ms = vals
    info, ps = state["info"], state["plan"]
    cells_total = float(n_total) * n_total
    per_step = elapsed / steps
    ms = per_step * 1e3
    # MFMA utilisation of this rank's filter launches: tiles computed x one int8 pass / their time / 5 POP/s (never > 1);
    # 2 d flop per cell of the rank's share of the N x N matrix stays beside it as algorithmic credit
    issued = ps["filter_tiles"] * 2.0 * 256 * 256 * ps["d_pad"] if not ps["exact_mode"] else 0.0
    roof = {"kernel": "k_pairwise_pp<filter> over the rank's block plan (%d launches)" % ps["filter_launches"], "bound": "mfma",
            "achieved": issued / (filter_ms * 1e-3) / 1e12 if filter_ms > 0 else None, "peak": INT8_MFMA_PEAK_TOPS,
            "unit": "TOP/s per GPU (int8 operations issued to the matrix cores, rank 0's tile count over the slowest rank's time)",
            "frac": issued / (filter_ms * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS if filter_ms > 0 else None,
            "issued_ops": issued, "tiles": ps["filter_tiles"], "kernel_ms": filter_ms,
            "algorithmic_credit": {"flops_per_gpu": 2.0 * d * cells_total / world, "kernels_ms": k2_max,
                                   "ratio_to_peak": 2.0 * d * cells_total / world / (k2_max * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS
                                   if k2_max > 0 else None,
                                   "note": "2 d flop per cell of the rank's share of the full matrix; not a utilisation (one "
                                           "triangle computed, one of four limb passes in the filter): may exceed 1"},
            "traffic": None}
    return {"metric": "samples projected/sec + pairwise Jaccard cells/sec, d=2048, 1/2/4/8 GPUs",
            "value": cells_total / per_step, "unit": "pairwise cells/s (ordered pairs of the full N x N matrix, whole job)",
            "n_gpus": world, "steps": steps, "warmup": warmup, "ms_per_step": ms,
            "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "int8 limbs x int32 MFMA accumulate; fp64 keep test", "data": "synthetic",
            "config": {"workload": "configs[%d]: %d synthetic samples, d=%d, pairwise, row shards over %d GPU(s)" %
                                   (config - 1, n_total, d, world),
                       "total_samples": n_total, "d": d, "limbs": info["limbs"], "kept_cells": kept, "cells_checksum": digest,
                       "schedule": info.get("schedule"), "overlap": info.get("overlap"), "gather_chunks": sc.gather_chunks,
                       "wire": info.get("wire"),
                       "collectives": info.get("collectives"), **comm_facts(coll, world)},
            "stages": {"prepare_own_rows_ms": prepare_ms,
                       "comparison_kernels_ms_max_over_ranks": k2_max,
                       "filter_ms": filter_ms, "recheck_ms": float(np.mean(acc["recheck"])), "flagged_tiles_ms": float(np.mean(acc["tiles"])),
                       "filter_launches": ps["filter_launches"], "filter_tiles": ps["filter_tiles"], "candidates": ps["candidates"],
                       "flagged_tiles": ps["flagged_tiles"],
                       "plan_span_ms": plan_span_ms,
                       # what the comparison waited for on top of its own kernels between plan begin and plan end: the
                       # exchange that the diagonal block did not cover, the mid-plan host synchronisation, launch gaps
                       "exposed_in_plan_ms": max(0.0, plan_span_ms - k2_max),
                       "allgather_ms_max_over_ranks": gather_max,
                       "allgather_bytes_per_rank": info["allgather_bytes_per_rank"],
                       "allgather_bytes_received_per_rank": info["allgather_bytes_per_rank"] * (world - 1),
                       "cells_route_exchange_sort_ms": cells_ms,
                       "exchanged_cells": info.get("exchanged_cells", 0),
                       "other_ms": ms - prepare_ms - plan_span_ms - cells_ms,
                       "instrumented_ms_per_step": instrumented_ms, "instrumented_steps": probe,
                       "note": "ms_per_step: K steps without instrumentation; the stage spans are from `instrumented_steps` more "
                               "steps behind the timed region with events on the step's streams (the last one's spans, max over "
                               "ranks; an event between two kernels costs the stream ~6 us, so instrumented_ms_per_step is the "
                               "larger and `other_ms` = ms_per_step - prepare - plan span - cells may come out negative); host "
                               "synchronisations per step: 1 in the steady state (cell counts; a plan of the same shape as the "
                               "previous step's runs ahead of its own read-backs), 2 on a first step"},
            # rank 0's last step: (what, ms since the step began) from events on the compute stream and on the exchange's
            # stream -- "filter launched" / "gathered" events complete when the work queued before them has
            "timeline": timeline,
            "roofline": roof}


def pairwise_leg(ctx, dev, n, d, nh, reps):
    """BASELINE.json configs[2]: all-vs-all comparison of n synthesised sketches (magnitudes of nh-hash samples,
    clusters of 16), sketches already resident in HBM as limb planes.  `reps` timed repetitions after 3 warm-up
    runs, each bracketed by torch.cuda.synchronize(); kernel durations from HIP events recorded by the library on
    the launch stream.  Timed twice: the default two-stage comparison and the exact kernel on every cell."""
    import torch
    from metagenome_vector_sketches_amd import synth
    sk = synth.make_sketches_torch(n, d, nh, seed=2345, device=dev)
    ss_dev = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.sumsq(sk, out=ss_dev)
    n2 = torch.from_numpy(fast_norm_sq(ss_dev.cpu().numpy(), d)).to(dev)
    sset = ctx.sketch_set(sk)
    del sk
    cells = torch.empty((max(1 << 22, 64 * n), 4), dtype=torch.int32, device=dev)
    out = {}
    for name, filt, r in (("two_stage", 1, reps), ("exact", 0, max(3, reps // 2))):
        with ctx.options(pairwise_filter=filt):
            for _ in range(3):
                _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
            torch.cuda.synchronize()
            wall, kern, filt_ms, chk_ms, tile_ms = [], [], [], [], []
            for _ in range(r):
                t1 = time.perf_counter()
                _, cnt = ctx.pairwise_rows(sset, n2, cells_out=cells)
                torch.cuda.synchronize()
                wall.append((time.perf_counter() - t1) * 1e3)
                kern.append(ctx.kernel_ms(1))
                if filt:
                    filt_ms.append(ctx.kernel_ms(2))
                    chk_ms.append(ctx.kernel_ms(3))
                    try:
                        tile_ms.append(ctx.kernel_ms(4))
                    except Exception:      # noqa: BLE001 -- no tile was flagged
                        pass
            out[name] = {"reps": r, "wall_ms": float(np.mean(wall)), "wall_ms_min": float(np.min(wall)),
                         "kernels_ms": float(np.mean(kern)), "kept_cells": int(cnt),
                         "candidates": ctx.pairwise_candidates()}
            if filt:
                out[name]["filter_kernel_ms"] = float(np.mean(filt_ms))
                out[name]["recheck_kernel_ms"] = float(np.mean(chk_ms))
                # tile-granular comparison: the 256 x 256 tiles whose waves held more than tile_dense_thr candidates went to
                # the exact kernel whole (here: the diagonal tiles, where the clusters of 16 sit)
                _, out[name]["flagged_tiles"], out[name]["filter_tiles"] = ctx.pairwise_stats()
                out[name]["flagged_tiles_kernel_ms"] = float(np.mean(tile_ms)) if tile_ms else 0.0
    limbs = sset.limbs
    sset.close()
    cells_total = float(n) * n
    flops = 2.0 * d * cells_total
    two, ex = out["two_stage"], out["exact"]
    traffic, src = pmc_traffic("configs[2]") if (n, d) == (100_000, 2048) else ({}, {"file": None, "dropped": "non-default size"})
    t_f = two.get("filter_kernel_ms", two["kernels_ms"])
    # MFMA UTILISATION: int8 operations the matrix cores were actually given / kernel time / 5 POP/s.
    # Filter: the 256 x 256 tiles the launch computed (the library's count: tiles on and above the diagonal of the
    # symmetric square) x ONE pass of 2 * 256^2 * d_pad operations.  Exact kernel: the 128 x 128 tiles on and above the
    # diagonal x the four limb-pair passes.  Neither can exceed 1.  What the result is WORTH -- 2 d operations per cell
    # of the N x N matrix the reference computes, over the same time -- is `algorithmic_credit`: it counts both
    # triangles although one is computed and four passes although the filter runs one, so it may exceed 1.
    d_pad = (d + 127) // 128 * 128
    n256, n128 = (n + 255) // 256, (n + 127) // 128
    filter_tiles = two.get("filter_tiles") or n256 * (n256 + 1) // 2
    issued_filter = filter_tiles * 2.0 * 256 * 256 * d_pad
    issued_exact = n128 * (n128 + 1) // 2 * 4 * 2.0 * 128 * 128 * d_pad
    roof = {"kernel": "k_pairwise_pp<filter> (ping-pong wave groups, 256 x 256 tiles, one int8 pass)", "bound": "mfma",
            "workload": "configs[2]", "achieved": issued_filter / (t_f * 1e-3) / 1e12,
            "peak": INT8_MFMA_PEAK_TOPS, "unit": "TOP/s (int8 operations issued to the matrix cores)",
            "frac": issued_filter / (t_f * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
            "issued_ops": issued_filter, "tiles": filter_tiles, "kernel_ms": t_f,
            "cross_check": "tiles x 2 x 256^2 x d_pad / 32768 = v_mfma_i32_16x16x64_i8 instructions per launch "
                           "(SQ_INSTS_VALU_MFMA_I8 in profiles/*_c2_pmc_summary.txt)",
            "mfma_instructions": issued_filter / 32768.0,
            "algorithmic_credit": {"flops": flops, "kernels_ms": two["kernels_ms"],
                                   "tflops": flops / (two["kernels_ms"] * 1e-3) / 1e12,
                                   "ratio_to_peak": flops / (two["kernels_ms"] * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
                                   "note": "2 d flop per cell of the full N x N matrix over filter + re-check + flagged "
                                           "tiles; NOT a utilisation: the symmetric schedule computes one triangle and "
                                           "the filter one of the four limb passes, so this ratio can exceed 1"},
            "exact_kernel": {"kernel": "k_pairwise_pp<exact> (4 limb-pair passes per cell)", "kernel_ms": ex["kernels_ms"],
                             "frac": issued_exact / (ex["kernels_ms"] * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS,
                             "issued_ops": issued_exact,
                             "algorithmic_credit_ratio": flops / (ex["kernels_ms"] * 1e-3) / 1e12 / INT8_MFMA_PEAK_TOPS},
            # the only single-pass exact matrix type would be fp64 (78.6 TFLOP/s dense on MI355X)
            "vs_fp64_matrix_peak": flops / (two["kernels_ms"] * 1e-3) / 1e12 / FP64_MATRIX_PEAK_TFLOPS,
            "traffic": traffic.get("k_pairwise_pp_filter"),
            "traffic_recheck": traffic.get("k_exact_pairs_tree", traffic.get("k_exact_pairs")),
            "traffic_source": src,
            "algorithmic_bytes": float(n) * d * limbs + 16.0 * two["kept_cells"]}
    leg = {"cells_per_s": cells_total / (two["wall_ms"] * 1e-3), "cells_per_s_kernels": cells_total / (two["kernels_ms"] * 1e-3),
           "cells_counted": "N^2 ordered pairs (the symmetric schedule computes the upper triangle and mirrors)",
           "limbs": limbs, "two_stage": two, "exact": ex,
           "exact_cells_per_s": cells_total / (ex["wall_ms"] * 1e-3)}
    return {"workload": "configs[2]: %d synthetic samples, d=%d, pairwise" % (n, d), "leg": leg, "roofline": roof}


def density_leg(ctx, dev, n, d, nh, clusters=(16, 1024, 10000), reps=2):
    """the comparison between "sparse" and "dense".  n samples in clusters of c related samples, c from
    16 (0.016 % of the cells kept) to n / 10 (10 %), streamed out as device-encoded rows (what the executable does); per
    point the comparison kernels' time, candidates and flagged tiles, against
        bound = two-stage kernels of the sparse point + flagged share of the tiles x exact kernel on every tile.
    The reference's cost is flat in the density (src/pairwise_comp_optimized.cpp:135-147); up to round 3 ours jumped from
    10 to 32+ ms once 1/128 of the cells were candidates."""
    import ctypes
    import torch
    from metagenome_vector_sketches_amd import _capi, synth
    seen = {"cells": 0}

    def count(_user, bp):
        seen["cells"] += bp.contents.n_cells
        return 0
    ecb = _capi.ENCODED_ROWS_CB(count)

    def stream(sset, n2):
        best = None
        for r in range(reps + 1):
            seen["cells"] = 0
            cnt = ctypes.c_int64()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            rc = ctx.lib.mvs_pairwise_stream_encoded(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0,
                                                     ecb, None, ctypes.byref(cnt))
            wall = (time.perf_counter() - t0) * 1e3
            if rc != 0 or seen["cells"] != cnt.value:
                raise SystemExit("density leg: streamed comparison failed (rc %d)" % rc)
            st = ctx.stream_stats()
            if r and (best is None or wall < best["wall_ms"]):
                best = {"wall_ms": wall, "kernels_ms": st["kernel_ms"], "bytes_to_host": st["bytes"], "row_blocks": st["row_blocks"],
                        "path": int(st["two_stage"]), "kept_cells": int(cnt.value)}
        return best
    points, exact_ms, sparse_ms = [], None, None
    for c in clusters:
        sk = synth.make_sketches_torch(n, d, nh, seed=2345, device=dev, cluster=c)
        ss = torch.empty(n, dtype=torch.int64, device=dev)
        ctx.sumsq(sk, out=ss)
        n2 = torch.from_numpy(fast_norm_sq(ss.cpu().numpy(), d)).to(dev)
        sset = ctx.sketch_set(sk)
        del sk
        if exact_ms is None:
            with ctx.options(pairwise_filter=0):
                exact_ms = stream(sset, n2)["kernels_ms"]
        p = stream(sset, n2)
        cand, flagged, tiles = ctx.pairwise_stats()
        if sparse_ms is None:
            sparse_ms = p["kernels_ms"]
        bound = sparse_ms + flagged / max(tiles, 1) * exact_ms
        p.update(cluster=c, density=p["kept_cells"] / float(n) / n, candidates=cand, flagged_tiles=flagged, filter_tiles=tiles,
                 bound_ms=bound, kernels_over_bound=p["kernels_ms"] / bound)
        points.append(p)
        sset.close()
        del sset
        # (no torch.cuda.empty_cache() here: handing torch's cached blocks back to the driver between the points makes the
        # NEXT stream's device-to-host copies run at ~30 instead of 55 GB/s on this stack -- 10 % point 55 ms instead of 38,
        # same kernels; tools/exp/link_state3.py variant E)
    return {"workload": "%d synthetic samples, d=%d, clusters of c samples, rows encoded on the device and streamed to the host"
                        % (n, d), "exact_kernel_every_tile_ms": exact_ms, "points": points}


def stream_leg(ctx, dev, n, d, nh, reps=3):
    """The comparison with a DENSE result -- clusters of n/3 samples, a third of all cells kept, the density of the
    reference's own toy set (1291 of 3721) -- streamed to the host as CSR pieces (mvs_pairwise_stream; the callback only
    counts): compare + download wall against the comparison kernels' own time and against the bare link time of the same
    bytes (pinned D2H in 32 MiB pieces), i.e. how much of the run the PCIe link accounts for."""
    import ctypes
    import torch
    from metagenome_vector_sketches_amd import _capi, synth
    cluster = max(16, n // 3)
    sk = synth.make_sketches_torch(n, d, nh, seed=2345, device=dev, cluster=cluster)
    ss = torch.empty(n, dtype=torch.int64, device=dev)
    ctx.sumsq(sk, out=ss)
    n2 = torch.from_numpy(fast_norm_sq(ss.cpu().numpy(), d)).to(dev)
    sset = ctx.sketch_set(sk)
    del sk
    seen = {"cells": 0, "rows": 0}

    def count(_user, bp):
        b = bp.contents
        seen["cells"] += b.n_cells
        seen["rows"] += b.row_end - b.row_begin
        return 0
    cb = _capi.ROW_BLOCK_CB(count)
    walls, stats = [], None
    for r in range(reps + 1):
        seen.update(cells=0, rows=0)
        cnt = ctypes.c_int64()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx.lib.mvs_pairwise_stream(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, cb, None,
                                         ctypes.byref(cnt))
        wall = (time.perf_counter() - t0) * 1e3
        if rc != 0 or seen["cells"] != cnt.value or seen["rows"] != n:
            raise SystemExit("streamed comparison failed: rc %d, %d of %d cells, %d of %d rows" %
                             (rc, seen["cells"], cnt.value, seen["rows"], n))
        if r:
            walls.append(wall)
            stats = ctx.stream_stats()
    # the same rows ENCODED on the device in the shard codec (what the executable does): fewer bytes, no host work per cell
    ecb = _capi.ENCODED_ROWS_CB(count)
    ewalls, estats = [], None
    for r in range(reps + 1):
        seen.update(cells=0, rows=0)
        cnt2 = ctypes.c_int64()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rc = ctx.lib.mvs_pairwise_stream_encoded(ctx._h, sset._h, n2.data_ptr(), _capi.MEM_DEVICE, _capi.KEEP_INT32, 0, n, 0, ecb,
                                                 None, ctypes.byref(cnt2))
        ewall = (time.perf_counter() - t0) * 1e3
        if rc != 0 or cnt2.value != cnt.value or seen["rows"] != n:
            raise SystemExit("streamed comparison (encoded rows) failed: rc %d, %d of %d cells" % (rc, cnt2.value, cnt.value))
        if r:
            ewalls.append(ewall)
            estats = ctx.stream_stats()
    sset.close()
    nbytes = max(stats["bytes"], 1)
    piece = min(nbytes, 32 << 20)
    src = torch.empty(piece, dtype=torch.uint8, device=dev)
    dst = [torch.empty(piece, dtype=torch.uint8).pin_memory() for _ in range(2)]
    link = []
    for _ in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range((nbytes + piece - 1) // piece):
            dst[i & 1].copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        link.append((time.perf_counter() - t0) * 1e3)
    wall = float(np.mean(walls))
    return {"workload": "%d synthetic samples in clusters of %d, d=%d: all-vs-all streamed to the host as CSR pieces" % (n, cluster, d),
            "kept_cells": int(cnt.value), "density": cnt.value / float(n) / n, "wall_ms": wall,
            "comparison_kernels_ms": stats["kernel_ms"], "row_blocks": stats["row_blocks"], "pieces": stats["pieces"],
            "path": {0: "exact kernel on every tile, dense byte matrix -> CSR on the device",
                     1: "two-stage comparison, one packed list sorted on the device",
                     2: "two-stage comparison feeding the dense byte matrix: flagged tiles by the exact kernel row block by row "
                        "block, the re-check's cells scattered, row passes over the tiles that can hold cells",
                     3: "the same with the filter itself running row block by row block"}[int(stats["two_stage"])],
            "candidates_flagged_filter_tiles": list(ctx.pairwise_stats()),
            "bytes_to_host": int(nbytes), "bytes_per_kept_cell": nbytes / max(cnt.value, 1),
            "bare_link_ms": min(link), "link_GBps": nbytes / (min(link) * 1e-3) / 1e9,
            "pcie_share_of_wall": min(link) / wall, "wall_over_max_kernel_link": wall / max(stats["kernel_ms"], min(link)),
            "cells_per_s": float(n) * n / (wall * 1e-3), "kept_cells_per_s": cnt.value / (wall * 1e-3),
            "encoded_rows": {"what": "mvs_pairwise_stream_encoded: rows in the shard codec (compact_vector of q + rice_sequence of the "
                                     "column deltas) encoded on the device, the bytes matrix.bin holds",
                             "wall_ms": float(np.mean(ewalls)), "bytes_to_host": int(estats["bytes"]),
                             "bytes_per_kept_cell": estats["bytes"] / max(cnt.value, 1),
                             "kept_cells_per_s": cnt.value / (float(np.mean(ewalls)) * 1e-3)}}


def search_leg(ctx, dev, n, d, nh, reps=5):
    """SURVEY 8f row 4, device part: q query sketches against n resident database sketches (mvs_search_block; the reference
    runs FAISS IndexFlatIP on float32 copies on the CPU, src/jaccard.py:117-200).  Every tenth query is a database sample
    (every twentieth an exact copy, the others a copy + an independent sample: Jaccard ~ 0.5), so the hit path is timed
    too.  The set stays resident, as a search service would keep it: from the second search on the coarse plane exists and
    up to 640 queries go through the streaming filter (k_search_filter: the query rows' coarse plane in LDS, the database's
    coarse plane streamed once per group of 64 rows into the matrix cores) + exact re-check of its candidates; 1024
    queries take the tile filter.  Every timed search uploads its query sketches into the scratch rows behind the database
    first (mvs_sketch_set_fill: what search.SearchIndex does per call -- the library re-derives the coarse rows of exactly
    those rows and keeps the database's); `wall_ms_resident_queries` is the same search on query rows already in place.
    Per row the roofline of the dominant kernel: HBM with the bytes that kernel has to read
    once (the coarse plane, d_pad per sketch; the exact streaming kernel would need both limb planes), or MFMA with
    2 d q N flops where the matrix cores are the nearer bound."""
    import torch
    from metagenome_vector_sketches_amd import synth
    nq_max = 1024
    sset = ctx.sketch_set_alloc(n + nq_max, d, 2)
    ss_all = torch.empty(n + nq_max, dtype=torch.int64, device=dev)
    step = 100_000
    donors = None
    for r0 in range(0, n, step):                             # the database
        rows = min(step, n - r0)
        sk = synth.make_sketches_torch(rows, d, nh, seed=4567 + r0, device=dev)
        if donors is None:
            donors = sk[:nq_max // 10 + 1].clone()
        ctx.sumsq(sk, out=ss_all[r0:r0 + rows])
        sset.fill(sk, r0)
        del sk
    q_sk = synth.make_sketches_torch(nq_max, d, nh, seed=99, device=dev)          # the queries are the last rows
    for k, qi in enumerate(range(0, nq_max, 10)):
        q_sk[qi] = donors[k] if k % 2 == 0 else donors[k] + q_sk[qi]
    ctx.sumsq(q_sk, out=ss_all[n:])
    sset.fill(q_sk, n)
    del donors
    n2 = ss_all.double() / d
    cells = torch.empty((1 << 20, 4), dtype=torch.int32, device=dev)
    coarse_bytes = float(n) * sset.d_pad
    out = {"workload": "%d resident synthetic sketches, d=%d, two limbs: q query sketches (a tenth of them database samples) "
                       "against all of them, Jaccard > 0.1" % (n, d),
           "algorithmic_bytes": coarse_bytes, "limb_plane_bytes": 2 * coarse_bytes, "queries": {}}
    for nq in (1, 16, 64, 256, 1024):
        walls, kern, walls_res = [], [], []
        for r in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            hits = ctx.search_block(sset, n2, 0.1, n, n + nq, 0, n, cells)
            torch.cuda.synchronize()
            if r >= 2:
                walls_res.append((time.perf_counter() - t0) * 1e3)
        for r in range(reps + 2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            sset.fill(q_sk[:nq], n)                           # fresh queries: their rows are re-coded and re-derived
            hits = ctx.search_block(sset, n2, 0.1, n, n + nq, 0, n, cells)
            torch.cuda.synchronize()
            if r >= 2:
                walls.append((time.perf_counter() - t0) * 1e3)
                kern.append(ctx.kernel_ms(1))
        k = float(np.mean(kern))
        cand = int(ctx.pairwise_candidates())
        rec = {"wall_ms": float(np.mean(walls)), "kernel_ms": k, "hits": int(hits),
               "wall_ms_resident_queries": float(np.mean(walls_res)),
               "pairs_per_s": float(n) * nq / (float(np.mean(walls)) * 1e-3), "two_stage_candidates": cand}
        if cand:
            rec["filter_ms"], rec["recheck_ms"] = float(ctx.kernel_ms(2)), float(ctx.kernel_ms(3))
        kt = rec.get("filter_ms", k)                          # the dominant kernel's own time
        hbm = coarse_bytes * (1 if cand else 2) / (kt * 1e-3) / 1e9
        flop = 2.0 * d * nq * float(n) / (kt * 1e-3) / 1e12
        if flop / 5000.0 > hbm / 8000.0:
            rec["roofline"] = {"bound": "mfma", "achieved": flop, "peak": 5000.0, "unit": "TFLOP/s", "frac": flop / 5000.0,
                               "kernel": "filter on the coarse plane (int8 MFMA), 2 d q N operations"}
        else:
            rec["roofline"] = {"bound": "hbm", "achieved": hbm, "peak": 8000.0, "unit": "GB/s", "frac": hbm / 8000.0,
                               "kernel": ("streaming filter: the coarse plane once" if cand else
                                          "streaming exact kernel: both limb planes once")}
        out["queries"][str(nq)] = rec
    del q_sk
    sset.close()
    return out


def usable_cores(visible):
    """threads worth starting: the cgroup CPU quota when there is one (a GPU box hands a 1-GPU job a share of the
    host's cores; 128 OpenMP threads on a 16-core quota run slower than 16), else every visible core"""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:                      # cgroup v2
            quota, period = f.read().split()[:2]
        if quota != "max":
            return max(1, min(visible, int(round(float(quota) / float(period)))))
    except Exception:
        pass
    try:
        with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f:         # cgroup v1
            quota = int(f.read())
        with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as f:
            period = int(f.read())
        if quota > 0:
            return max(1, min(visible, int(round(quota / period))))
    except Exception:
        pass
    try:
        return max(1, min(visible, len(os.sched_getaffinity(0))))
    except Exception:
        return visible


def reference_sketch_leg(hashes, o_all, S, D, cores, orc, ns=128):
    """-> dict for cpu_baseline.detail.projection.reference_binary (None-valued when the binary is not there)"""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "project_everything")
    if not os.path.exists(exe):
        return {"available": False, "why": "oracle/_ref/project_everything not built (needs /root/reference at build time)"}
    ns = min(ns, S)
    o = o_all[:ns + 1]
    h = hashes[:int(o[ns])].cpu().numpy().view(np.uint64)
    tmp = tempfile.mkdtemp(prefix="mvs_refleg_")
    try:
        hf = os.path.join(tmp, "hashes.txt")
        with open(hf, "w") as f:
            for i in range(ns):
                f.write("s%d: " % i + " ".join(map(str, h[int(o[i]):int(o[i + 1])].tolist())) + "\n")
        env = dict(os.environ, OMP_NUM_THREADS=str(cores))
        t0 = time.perf_counter()
        r = subprocess.run([exe, "sketch", hf, os.path.join(tmp, "db"), "-d", str(D)], env=env, capture_output=True,
                           text=True, timeout=300)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"available": False, "why": "reference binary failed: " + r.stderr[-200:]}
        span = None
        for line in r.stdout.splitlines():
            if line.startswith("Time to compute all projected vectors:"):
                span = float(line.split(":")[1].split()[0])
        got = np.fromfile(os.path.join(tmp, "db", "vectors.bin"), dtype="<i4").reshape(ns, D)
        same = bool(np.array_equal(got, orc.project_csr(h, o, D, threads=cores, fast=True, native=True)))
        return {"available": True, "samples": ns, "hashes": int(o[ns]), "threads": cores,
                "reference_span_s": span, "wall_s": wall,
                "samples_per_s": (ns / span) if span else None,
                "vectors_equal_to_port": same,
                "note": "parse + projection as the reference times them; the ports above time the projection alone"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def reference_pairwise_leg(skp, n2_text_lines, D, cores, orc, port_cells, n=4096):
    """-> dict for cpu_baseline.detail.pairwise.reference_functions: the reference's OWN block loader, int32 Eigen product and
    keep test (oracle/_ref/ref_pairwise32: src/pairwise_comp_optimized.cpp:33-160 compiled from line ranges, loop restated in
    oracle/ref_pairwise_driver.inc) on the first n sketches, --max_memory_gb 12 (chunk 192), all cores: its rate calibrates the
    port, and its kept (i, j, P) list must equal the port's."""
    import shutil
    import subprocess
    import tempfile
    exe = os.path.join(os.path.dirname(os.path.abspath(__file__)), "oracle", "_ref", "ref_pairwise32")
    if not os.path.exists(exe):
        return {"available": False, "why": "oracle/_ref/ref_pairwise32 not built (needs /root/reference at build time)"}
    tmp = tempfile.mkdtemp(prefix="mvs_refpw_")
    try:
        np.ascontiguousarray(skp[:n], dtype="<i4").tofile(os.path.join(tmp, "vectors.bin"))
        with open(os.path.join(tmp, "vector_norms.txt"), "w") as f:
            f.write("".join("s%d %s\n" % (i, t) for i, t in enumerate(n2_text_lines[:n])))
        t0 = time.perf_counter()
        r = subprocess.run([exe, os.path.join(tmp, "vectors.bin"), os.path.join(tmp, "vector_norms.txt"), str(D), "12.0", "1", "0",
                            str(cores)], capture_output=True, text=True, timeout=600)
        wall = time.perf_counter() - t0
        if r.returncode != 0:
            return {"available": False, "why": "reference functions failed: " + r.stderr[-200:]}
        got = np.array([[int(t) for t in line.split()] for line in r.stdout.splitlines() if line], dtype=np.int64).reshape(-1, 3)
        want = np.stack([port_cells["row"], port_cells["col"], port_cells["dot"]], axis=1).astype(np.int64).reshape(-1, 3)
        return {"available": True, "samples": n, "threads": cores, "wall_s": wall, "cells_per_s": float(n) * n / wall,
                "kept": int(len(got)), "kept_cells_equal_to_port_in_order": bool(np.array_equal(got, want)),
                "note": "whole process: tile loads from the file, Eigen int32 product, keep test, printing the kept cells"}
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def cpu_baseline(hashes, offsets, S, NH, D, dev):
    """The oracle (CPU port of the reference path, kind "port") timed on this host's cores on bounded samples of the
    same workloads, as BASELINE.md section 3 lays out: projection and pairwise each with 8 threads (the reference's
    example flag) and with all cores; pairwise at N = 4096 and N = 16384 (a row stripe where the full square would
    take too long); the toy set (configs[0]) with 8 threads.  Projection extrapolates linearly in samples, pairwise
    quadratically in N.  `value` is the configs[1] job on all cores.  The pairwise port has never been timed against
    the real reference (its translation unit needs the absent `bits` submodule): BASELINE.md has the one calibration
    point there is (survey-session build)."""
    from oracle import pyoracle as orc
    from metagenome_vector_sketches_amd import synth
    visible = orc.max_threads()               # before any call changes the OpenMP thread count
    cores = usable_cores(visible)
    thr_list = [8, cores] if cores > 8 else [cores]
    o_all = np.asarray(offsets)
    mean_size = float(o_all[S]) / S
    detail = {"threads": thr_list, "projection": {}, "pairwise": {}}
    t_begin = time.perf_counter()

    # ---- projection: both ports; sample sizes chosen for ~1-2 s each ----
    def proj_rate(ns, thr, fast):
        ns = min(ns, S)
        o = o_all[:ns + 1]
        h = hashes[:int(o[ns])].cpu().numpy().view(np.uint64)
        t0 = time.perf_counter()
        sk = orc.project_csr(h, o, D, threads=thr, fast=fast, native=True)
        dt = time.perf_counter() - t0
        return ns / dt * (float(o[ns]) / ns) / mean_size, sk      # samples/s at the job's mean sample size

    for thr in thr_list:
        n_fast = 1024 if thr > 8 else 128
        n_lit = 256 if thr > 8 else 48
        r_lit, sk_lit = proj_rate(n_lit, thr, False)
        r_fast, sk_fast = proj_rate(n_fast, thr, True)
        assert np.array_equal(sk_fast[:len(sk_lit)], sk_lit)
        detail["projection"]["%d_threads" % thr] = {
            "literal_port_samples_per_s": r_lit, "restructured_port_samples_per_s": r_fast,
            "samples_timed": [min(n_lit, S), min(n_fast, S)]}
    best_proj = max(max(v["literal_port_samples_per_s"], v["restructured_port_samples_per_s"])
                    for v in detail["projection"].values())

    # ---- projection by the reference's OWN executable (oracle/_ref/project_everything, compiled from its sources in
    # the dev container and shipped as a binary): `sketch` on a text file of the first samples.  Its span "Time to
    # compute all projected vectors" covers the serial text parse and the OpenMP projection (project_everything.cpp:
    # 255-303); -t is ignored by the reference's sketch (:373-408), so OMP_NUM_THREADS carries the core quota. ----
    detail["projection"]["reference_binary"] = reference_sketch_leg(hashes, o_all, S, D, cores, orc)

    # ---- pairwise: N = 4096 (full square) and N = 16384 (row stripe sized from the first rate) ----
    skp = synth.make_sketches_torch(16384, D, NH, seed=2345, device=dev).cpu().numpy()
    n2 = fast_norm_sq((skp.astype(np.int64) ** 2).sum(1), D)
    best_pw = 0.0
    for thr in thr_list:
        t0 = time.perf_counter()
        c4 = orc.pairwise_rows(skp[:4096], n2[:4096], chunk=192, threads=thr, native=True)
        r4 = 4096.0 * 4096.0 / (time.perf_counter() - t0)
        rows = int(min(16384, max(192, (4.0 * r4 / 16384.0) // 192 * 192)))       # ~4 s of work
        t0 = time.perf_counter()
        c16 = orc.pairwise_rows(skp, n2, row_begin=0, row_end=rows, chunk=192, threads=thr, native=True)
        r16 = rows * 16384.0 / (time.perf_counter() - t0)
        detail["pairwise"]["%d_threads" % thr] = {
            "N4096_cells_per_s": r4, "N4096_kept": len(c4), "N16384_cells_per_s": r16, "N16384_rows_timed": rows,
            "N16384_kept_in_stripe": len(c16), "extrapolated_100k_x_2048_s": 1e10 / r16,
            "extrapolated_1M_x_2048_s": 1e12 / r16}
        best_pw = max(best_pw, r4, r16)

    # ---- the reference's own pairwise functions on the same first 4096 sketches (calibration + parity on this host) ----
    try:
        # the norms as vector_norms.txt would carry them: "%g" of sqrt(sum v^2 / d); both sides get the squares of the
        # parsed text (stod(text)^2, :893-901)
        n2_lines = ["%g" % x for x in np.sqrt((skp[:4096].astype(np.int64) ** 2).sum(1).astype(np.float64) / D)]
        n2_ref = np.array([float(t) * float(t) for t in n2_lines])
        c4_all = orc.pairwise_rows(skp[:4096], n2_ref, chunk=192, threads=cores, native=True)
        detail["pairwise"]["reference_functions"] = reference_pairwise_leg(skp, n2_lines, D, cores, orc, c4_all)
        detail["pairwise"]["reference_functions"]["norms_equal_to_fast_norm_sq"] = bool(np.array_equal(n2_ref, n2[:4096]))
        rf = detail["pairwise"]["reference_functions"]
        if rf.get("available"):
            rf["port_over_reference"] = detail["pairwise"]["%d_threads" % thr_list[-1]]["N4096_cells_per_s"] / rf["cells_per_s"]
    except Exception as e:      # noqa: BLE001 -- a calibration leg must not take the bench line down
        detail["pairwise"]["reference_functions"] = {"available": False, "why": repr(e)}

    # ---- configs[0]: the reference's toy set, 8 threads, sketch + all-vs-all (fixture committed under tests/golden) ----
    try:
        g = np.load(os.path.join(ROOT, "tests", "golden", "toy_hashes.npz"))
        offs = g["offsets"].astype(np.int64)
        hh = g["deltas"].astype(np.uint64).copy()
        for i in range(len(offs) - 1):
            hh[offs[i]:offs[i + 1]] = np.cumsum(hh[offs[i]:offs[i + 1]], dtype=np.uint64)
        t0 = time.perf_counter()
        tsk = orc.project_csr(hh, offs, 2048, threads=min(8, cores), native=True)
        t_sk = time.perf_counter() - t0
        tn2 = fast_norm_sq((tsk.astype(np.int64) ** 2).sum(1), 2048)
        t0 = time.perf_counter()
        tc = orc.pairwise_rows(tsk, tn2, chunk=192, threads=min(8, cores), native=True)
        detail["toy_8_threads"] = {"samples": len(offs) - 1, "hashes": int(offs[-1]), "sketch_s": t_sk,
                                   "pairwise_s": time.perf_counter() - t0, "kept_cells": len(tc)}
    except Exception as e:      # fixture missing: say so rather than fail the bench line
        detail["toy_8_threads"] = {"error": repr(e)}

    t_job = S / best_proj + float(S) * S / best_pw
    # the same job priced with the REFERENCE's own code where it could be built (oracle/_ref): its `sketch` executable (text
    # parse + OpenMP projection, as it times itself) and its own pairwise functions (tile loads, Eigen int32 product, keep test)
    ref_value = None
    try:
        rp = detail["projection"]["reference_binary"].get("samples_per_s")
        rf = detail["pairwise"]["reference_functions"].get("cells_per_s")
        if rp and rf:
            ref_value = S / (S / rp + float(S) * S / rf)
    except Exception:      # noqa: BLE001
        pass
    detail["value_with_reference_code"] = ref_value
    detail["seconds_spent"] = time.perf_counter() - t_begin
    detail["calibration"] = ("projection port calibrated against the reference binary in the dev container (BASELINE.md "
                             "section 3 table: 1.0-2.2x the reference's speed, the faster port is used) and on this host "
                             "(detail.projection.reference_binary); pairwise port calibrated on this host against the reference's "
                             "own loader / Eigen product / keep test compiled from line ranges (detail.pairwise.reference_functions: "
                             "port_over_reference); the reference's writers need the absent `bits` submodule and are in neither")
    detail["host_threads_visible"] = visible
    try:                                                  # state the host CPU the baseline ran on
        with open("/proc/cpuinfo") as f:
            models = [l.split(":", 1)[1].strip() for l in f if l.startswith("model name")]
        detail["host_cpu_model"] = models[0] if models else None
    except Exception:   # noqa: BLE001
        detail["host_cpu_model"] = None
    return {"value": S / t_job, "unit": "samples/s (projected and compared all-vs-all)", "cores": cores,
            "kind": "port", "value_reference": ref_value,
            "sample": "projection: %s samples x %d hashes per thread count %s, faster of the literal and the restructured "
                      "port, and the reference's own sketch executable on 128 samples where oracle/_ref is present "
                      "(detail.projection.reference_binary); pairwise: N=4096 full square and a row stripe of N=16384, d=%d, chunk 192; toy set with 8 "
                      "threads; best all-core rates extrapolated to %d samples (linear + quadratic)" %
                      ("/".join(str(x) for x in detail["projection"]["%d_threads" % thr_list[-1]]["samples_timed"]), NH,
                       thr_list, D, S),
            "projection_samples_per_s": best_proj, "pairwise_cells_per_s": best_pw, "detail": detail}


if __name__ == "__main__":
    main()
