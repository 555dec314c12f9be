"""GPU, several ranks sharing the one card of the GPU box (gloo for the collectives; the driver's multi-GPU
runs use RCCL): the real HIP back end under both multi-rank schedules -- rows x all-columns, and the
symmetric one (every unordered block pair once, mirrored cells exchanged).  Each rank's shard must equal
the oracle's rows for that shard, cell for cell."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _make(n, d):
    from metagenome_vector_sketches_amd import synth
    from oracle import pyoracle as orc
    sk = synth.make_sketches_numpy(n, d, 3000, seed=321, cluster=8)
    n2 = np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(r))) for r in sk])
    return sk, n2


def _worker(rank, world, port, n, d, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import metagenome_vector_sketches_amd as pkg
    from metagenome_vector_sketches_amd import parallel
    torch.cuda.set_device(0)
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream())
    sk, n2 = _make(n, d)
    b, e = parallel.shard_rows(n, world, rank)
    local = torch.from_numpy(sk[b:e]).to("cuda:0")
    out = torch.empty((n * 40, 4), dtype=torch.int32, device="cuda:0")
    for sym in (True, False):
        sc = parallel.ShardedComparison(parallel.GpuOps(ctx, torch.device("cuda", 0)), rank, world, dist)
        sc.symmetric = sym
        _, cnt, info = sc.run(local, n2[b:e], n, cells_out=out)
        torch.cuda.synchronize()
        assert (info.get("schedule") == "symmetric") == sym
        np.save(os.path.join(out_dir, "cells_%d_%d.npy" % (int(sym), rank)), out[:cnt].cpu().numpy())
    dist.barrier()
    ctx.close()
    dist.destroy_process_group()


@pytest.mark.parametrize("n,world,two_stage", [(700, 2, False), (650, 3, False), (700, 2, True), (650, 3, True)])
def test_ranks_on_one_gpu_match_oracle(tmp_path, monkeypatch, n, world, two_stage):
    # two_stage: force the coarse filter + exact re-check on these small blocks (symmetric, mirror-all and
    # plain block calls all go through it); the spawned ranks inherit the environment
    monkeypatch.setenv("MVS_PAIRWISE_FILTER", "2" if two_stage else "0")
    d, port = 512, 29800 + (os.getpid() + n + 7 * two_stage) % 1000
    mp.spawn(_worker, args=(world, port, n, d, str(tmp_path)), nprocs=world, join=True)
    from oracle import pyoracle as orc
    sk, n2 = _make(n, d)
    want = orc.pairwise_rows(sk, n2, chunk=192, threads=8)
    want = want[np.lexsort((want["col"], want["row"]))]
    want = np.stack([want[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    assert len(want) > 6 * n
    for sym in (1, 0):
        got = np.concatenate([np.load(os.path.join(str(tmp_path), "cells_%d_%d.npy" % (sym, r))) for r in range(world)])
        assert np.array_equal(got, want), sym


def _native_worker(rank, world, n, d, out_dir):
    """no torch.distributed at all: the exchange goes through the C ABI's communicator (file transport: the ranks
    share the one card, which RCCL refuses)"""
    import metagenome_vector_sketches_amd as pkg
    from metagenome_vector_sketches_amd import parallel
    torch.cuda.set_device(0)
    ctx = pkg.Context(0)
    ctx.set_stream(torch.cuda.current_stream())
    comm = ctx.comm_files(os.path.join(out_dir, "xchg"), rank, world)
    assert (comm.rank, comm.world, comm.is_rccl) == (rank, world, False)
    sk, n2 = _make(n, d)
    b, e = parallel.shard_rows(n, world, rank)
    local = torch.from_numpy(sk[b:e]).to("cuda:0")
    out = torch.empty((n * 40, 4), dtype=torch.int32, device="cuda:0")
    for sym in (True, False):
        sc = parallel.ShardedComparison(parallel.GpuOps(ctx, torch.device("cuda", 0)), rank, world,
                                        collectives=parallel.NativeCollectives(comm))
        sc.symmetric = sym
        _, cnt, info = sc.run(local, n2[b:e], n, cells_out=out)
        torch.cuda.synchronize()
        assert "file transport" in info["collectives"]
        np.save(os.path.join(out_dir, "ncells_%d_%d.npy" % (int(sym), rank)), out[:cnt].cpu().numpy())
    comm.close()
    # local rows arriving in parts, the communicator on a context and stream of its own (what bench.py --gpus N does):
    # mvs_allgather_rows moves a part of every rank's block while the compute stream goes on
    side = torch.cuda.Stream()
    ctx2 = pkg.Context(0)
    ctx2.set_stream(side)
    comm2 = ctx2.comm_files(os.path.join(out_dir, "xchg"), rank, world)
    sc = parallel.ShardedComparison(parallel.GpuOps(ctx, torch.device("cuda", 0)), rank, world,
                                    collectives=parallel.NativeCollectives(comm2, stream=side))
    n2_dev = torch.from_numpy(n2[b:e].copy()).to("cuda:0")
    for parts in (2, 3):
        sc.begin(local, n2_dev, n)
        bounds = sc.part_bounds(n, parts)              # on multiples of 256 storage rows: a small block is one part
        for (p0, p1) in bounds:
            q0, q1 = min(p0, e - b), min(p1, e - b)
            sc.feed(p0, p1, int(local[q0:q1].abs().max()) if q1 > q0 else 0)
        _, cnt, info = sc.finish(cells_out=out)
        torch.cuda.synchronize()
        assert ("exchange of a part" in info["overlap"]) == (len(bounds) > 1)
        np.save(os.path.join(out_dir, "ncells_p%d_%d.npy" % (parts, rank)), out[:cnt].cpu().numpy())
    comm2.close()
    ctx2.close()
    ctx.close()


@pytest.mark.parametrize("n,world,stale", [(700, 2, False), (650, 3, False), (700, 2, True)])
def test_native_communicator_ranks_on_one_gpu(tmp_path, n, world, stale):
    d = 512
    if stale:
        # what a killed earlier job with the same prefix leaves behind: blocks of the first sequence numbers (one of
        # them exactly the 8 bytes of the max|v| all-reduce), hello / ready files with that job's numbers, an
        # acknowledgement.  None of it may be read as this job's data, and all of it is gone afterwards.
        for name, data in (("xchg_0_0", b"\x7f" * 8), ("xchg_0_1", b"\x7f" * 8), ("xchg_1_1", b"\x01" * 4096),
                           ("xchg_hello_0", b"\x11" * 8), ("xchg_hello_1", b"\x22" * 8), ("xchg_ready_0", b"\x33" * 8),
                           ("xchg_ready_1", b"\x33" * 8), ("xchg_0_1.ack0", b"")):
            with open(os.path.join(str(tmp_path), name), "wb") as f:
                f.write(data)
    mp.spawn(_native_worker, args=(world, n, d, str(tmp_path)), nprocs=world, join=True)
    from oracle import pyoracle as orc
    sk, n2 = _make(n, d)
    want = orc.pairwise_rows(sk, n2, chunk=192, threads=8)
    want = want[np.lexsort((want["col"], want["row"]))]
    want = np.stack([want[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    for sym in (1, 0):
        got = np.concatenate([np.load(os.path.join(str(tmp_path), "ncells_%d_%d.npy" % (sym, r))) for r in range(world)])
        assert np.array_equal(got, want), sym
    for parts in (2, 3):
        got = np.concatenate([np.load(os.path.join(str(tmp_path), "ncells_p%d_%d.npy" % (parts, r))) for r in range(world)])
        assert np.array_equal(got, want), parts
    assert not [f for f in os.listdir(str(tmp_path)) if f.startswith("xchg")]      # the transport cleans up after itself


def test_rccl_communicator_first_contact():
    """RCCL through the C ABI with a world of one (all a one-GPU box can do): the library is found and bound at run
    time, ncclGetUniqueId / ncclCommInitRank succeed, the collectives return at once and leave the data alone"""
    import metagenome_vector_sketches_amd as pkg
    from metagenome_vector_sketches_amd import _capi
    ctx = pkg.Context(0)
    uid = _capi.comm_unique_id()
    assert len(uid) == _capi.COMM_ID_BYTES and any(uid)
    comm = ctx.comm_rccl(uid, 0, 1)
    assert (comm.rank, comm.world, comm.is_rccl) == (0, 1, True)
    buf = torch.arange(1024, dtype=torch.int8, device="cuda:0")
    comm.allgather_bytes(buf, 1024)
    ctx.synchronize()
    assert torch.equal(buf.cpu(), torch.arange(1024, dtype=torch.int8))
    assert comm.allreduce_max(41) == 41
    comm.close()
    ctx.close()


def test_rccl_beside_torchs_nccl_backend(tmp_path):
    """8-GPU readiness that one card can check: a process that has ALREADY initialised torch.distributed's nccl backend (as
    every bench.py rank has: the barrier and the id broadcast go through it) creates the library's own RCCL communicator
    from an id that travelled through a torch collective, and runs its collectives.  Both use the same librccl -- the
    loader hands the library's dlopen the copy torch has mapped -- which mvs_comm_library reports; the bench line carries
    it (config.rccl_library / rccl_version)."""
    import subprocess
    script = tmp_path / "coexist.py"
    script.write_text('''
import os, sys, json
sys.path.insert(0, %r)
import torch
import torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29611", RANK="0", WORLD_SIZE="1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.ones(4, device="cuda")
dist.all_reduce(x)                                   # torch's communicator exists and has done work
import metagenome_vector_sketches_amd as pkg
from metagenome_vector_sketches_amd import _capi
ctx = pkg.Context(0)
uid = torch.zeros(_capi.COMM_ID_BYTES, dtype=torch.uint8, device="cuda")
uid.copy_(torch.frombuffer(bytearray(_capi.comm_unique_id()), dtype=torch.uint8))
dist.broadcast(uid, src=0)                           # the way bench.py hands the id to the other ranks
comm = ctx.comm_rccl(bytes(uid.cpu().numpy().tobytes()), 0, 1)
buf = torch.arange(4096, dtype=torch.int8, device="cuda")
comm.allgather_bytes(buf, 4096)
ctx.synchronize()
ok = bool(torch.equal(buf.cpu(), torch.arange(4096, dtype=torch.int8))) and comm.allreduce_max(7) == 7
dist.all_reduce(x)                                   # and torch's still works afterwards
torch.cuda.synchronize()
path, version = _capi.comm_library()
mapped = [l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l]
comm.close(); ctx.close(); dist.destroy_process_group()
print(json.dumps({"ok": ok, "is_rccl": True, "x": float(x[0]), "path": path, "version": version, "mapped": sorted(set(mapped))}))
''' % ROOT)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    import json
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert d["ok"] and d["x"] == 1.0 and d["version"] >= 20000
    assert os.path.basename(d["path"]).startswith("librccl") and d["path"] in d["mapped"]
    assert len(d["mapped"]) == 1, d["mapped"]          # ONE copy of RCCL in the process, shared by torch and the library


def _bare_bench(extra, timeout=600):
    """`python3 bench.py --gpus 2 ...` exactly as the round driver types it for a SCALE run -- no launcher, no RANK /
    WORLD_SIZE in the environment -- with MVS_BENCH_REHEARSAL=1 putting both ranks on the one card (gloo + the file
    transport instead of RCCL, which refuses two ranks on one device)"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVS_PAIRWISE_FILTER")}
    env["MVS_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"] + extra, env=env, capture_output=True, text=True, timeout=timeout)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]                     # ONE JSON line, nothing else on stdout
    return json.loads(lines[0])


def test_bare_bench_gpus2_launches_itself():
    d = _bare_bench(["--samples", "2000", "--hashes", "4000"])
    assert d["n_gpus"] == 2 and d["steps"] == 2 and d["scaling"] == "weak" and d["value"] > 0
    assert d["config"]["total_samples"] == 4000 and d["config"]["schedule"] == "symmetric"
    assert d["config"]["comm_world"] == 2 and d["config"]["rccl_ranks"] == 0      # rehearsal: file transport, not RCCL
    assert d["config"]["kept_cells"] >= 4000 * 10
    # per rank: a block of 2048 storage rows (2000 samples padded to the tile grid) x (low limbs + the coarse plane: the
    # receiver rebuilds the high limbs) x 2048 bytes, + 24 bytes of statistics and norm per row
    assert d["stages"]["allgather_bytes_per_rank"] == 2048 * 2 * 2048 + 2048 * 24 and d["stages"]["allgather_ms"] > 0


def test_bare_bench_gpus2_keeps_its_headline_when_the_strong_legs_do_not_finish():
    """the strong legs of a multi-GPU line run behind a guard (bench.py: run_strong_guarded): with a limit they cannot meet
    every rank gives up on them, rank 0 still prints the ONE line with the configs[1] headline, `strong.error` says why,
    and the exit code is 0 -- RCCL with more than one rank has never run anywhere: a failure there must not cost the line"""
    d = _bare_bench(["--samples", "2000", "--hashes", "4000", "--strong-timeout", "1"])
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["config"]["kept_cells"] >= 4000 * 10
    assert "did not finish within 1 s" in d["strong"]["error"]


def test_bare_bench_gpus2_moves_to_torchs_collectives_when_one_rank_finds_no_native_communicator():
    """bench.py's communicator set-up agrees in two phases (what a rank finds out alone, then the collective creation): a rank
    whose library binds no RCCL (test hook MVS_BENCH_FAIL_NATIVE_COMM=<rank>) takes every rank to torch.distributed's collectives
    -- before the change the other rank waited inside the communicator's creation for ever --, the line says so, the cells are the
    same; with --require-native-collectives every rank exits 3 instead"""
    import subprocess
    want = _bare_bench(["--samples", "2000", "--hashes", "4000", "--strong-steps", "0"])
    os.environ["MVS_BENCH_FAIL_NATIVE_COMM"] = "1"
    try:
        d = _bare_bench(["--samples", "2000", "--hashes", "4000", "--strong-steps", "0"])
        env = {k: v for k, v in os.environ.items()
               if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVS_PAIRWISE_FILTER")}
        env["MVS_BENCH_REHEARSAL"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                            "--no-cpu-baseline", "--samples", "2000", "--hashes", "4000", "--strong-steps", "0",
                            "--require-native-collectives"], env=env, capture_output=True, text=True, timeout=300)
    finally:
        del os.environ["MVS_BENCH_FAIL_NATIVE_COMM"]
    assert d["config"]["collectives"] == "torch.distributed" and d["config"]["rccl_ranks"] == 0
    assert "carries the exchange instead of the library's communicator" in d["config"]["collectives_note"]
    assert d["config"]["kept_cells"] == want["config"]["kept_cells"] and want["config"]["collectives"].startswith("libmvs_hip mvs_comm")
    assert r.returncode == 3 and not [l for l in r.stdout.splitlines() if l.startswith("{")], (r.returncode, r.stderr[-1500:])
    assert "not falling back (--require-native-collectives)" in r.stderr


def test_bare_bench_gpus2_config4_launches_itself():
    d = _bare_bench(["--config", "4"])
    assert d["n_gpus"] == 2 and d["scaling"] == "strong" and d["config"]["total_samples"] == 100_000
    assert d["config"]["d"] == 4096 and d["config"]["comm_world"] == 2
    assert d["config"]["kept_cells"] >= 100_000 * 10
    assert d["stages"]["filter_launches"] >= 2 and d["config"]["schedule"] == "symmetric"


def test_strong_scaled_step_gives_the_same_cells_for_every_rank_count():
    """configs[2] (100k x 2048) through bench.py with 1, 2 and 4 ranks (the ranks share the card: MVS_BENCH_REHEARSAL):
    the same number of kept cells and the same order-independent checksum over all shards -- the union of the shards is
    bit for bit the one-rank result -- and every rank count reports its stage times"""
    import json
    import subprocess
    env = {k: v for k, v in os.environ.items()
           if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT", "MVS_PAIRWISE_FILTER")}
    env["MVS_BENCH_REHEARSAL"] = "1"
    seen = {}
    for g in (1, 2, 4):
        r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(g), "--config", "3", "--steps", "2",
                            "--warmup", "1"], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stderr[-2000:]
        d = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
        seen[g] = (d["config"]["kept_cells"], d["config"]["cells_checksum"])
        assert d["stages"]["filter_tiles"] > 0 and d["roofline"]["frac"] is not None and d["roofline"]["frac"] <= 1.0
        assert [x for x in d["timeline"] if x[0].startswith("filter launched: diagonal block")]
    assert seen[1] == seen[2] == seen[4] and seen[1][0] > 1_000_000, seen
