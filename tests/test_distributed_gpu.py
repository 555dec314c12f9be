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
    ctx.close()


@pytest.mark.parametrize("n,world", [(700, 2), (650, 3)])
def test_native_communicator_ranks_on_one_gpu(tmp_path, n, world):
    d = 512
    mp.spawn(_native_worker, args=(world, n, d, str(tmp_path)), nprocs=world, join=True)
    from oracle import pyoracle as orc
    sk, n2 = _make(n, d)
    want = orc.pairwise_rows(sk, n2, chunk=192, threads=8)
    want = want[np.lexsort((want["col"], want["row"]))]
    want = np.stack([want[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    for sym in (1, 0):
        got = np.concatenate([np.load(os.path.join(str(tmp_path), "ncells_%d_%d.npy" % (sym, r))) for r in range(world)])
        assert np.array_equal(got, want), sym
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
