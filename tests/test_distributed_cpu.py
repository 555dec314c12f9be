"""CPU, world_size 2, gloo: the row-sharded comparison (metagenome_vector_sketches_amd/parallel.py).

The orchestration code is the product's; the numeric back end is replaced by a stand-in built on the
oracle (tests may use the oracle; the product may not), so what is verified here is the sharding, the
plane-buffer layout that the all-gather assembles, the padding of an uneven last shard and the norms
exchange: the union of the ranks' shards must equal the single-process result cell for cell."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class OracleOps:
    """CPU stand-in for parallel.GpuOps with the same plane layout as the HIP limb-split kernel:
    planes[(row*limbs + limb)*d_pad + k], signed base-256 digits."""

    def __init__(self):
        from oracle import pyoracle
        self.orc = pyoracle

    def max_abs(self, sk):
        return int(np.abs(np.asarray(sk, dtype=np.int64)).max()) if sk.size else 0

    def limbs_for(self, m):
        return 1 if m <= 127 else 2 if m <= 32639 else 3 if m <= 8355711 else 4

    def limb_geometry(self, n, d, limbs):
        n_alloc = (n + 127) // 128 * 128 + 128
        d_pad = (d + 127) // 128 * 128
        return n_alloc, d_pad, n_alloc * limbs * d_pad

    def new_planes(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.int8)

    def limb_split(self, sk, limbs, planes, d_pad, row_offset):
        v = np.asarray(sk, dtype=np.int64).copy()
        n, d = v.shape
        view = planes.numpy().reshape(-1, limbs, d_pad)
        for l in range(limbs):
            digit = ((v + 128) % 256) - 128
            view[row_offset:row_offset + n, l, :d] = digit.astype(np.int8)
            v = (v - digit) // 256

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def _sketches(self, planes, n, d, d_pad, limbs):
        view = planes.numpy().reshape(-1, limbs, d_pad).astype(np.int64)
        return sum(view[:n, l, :d] * (256 ** l) for l in range(limbs)).astype(np.int32)

    def compare(self, planes, n, n_alloc, d, d_pad, limbs, norms_sq, rb, re, keep_mode, cells_out):
        sk = self._sketches(planes, n, d, d_pad, limbs)
        cells = self.orc.pairwise_rows(sk, norms_sq.numpy(), row_begin=rb, row_end=re, chunk=192, threads=2)
        order = np.lexsort((cells["col"], cells["row"]))
        if cells_out is None:
            return cells[order], len(cells)
        arr = np.stack([cells[order][k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
        cells_out[:len(arr)] = torch.from_numpy(arr)
        return cells_out, len(arr)

    # ---- block interface of the symmetric schedule ----
    def new_cells(self, capacity):
        return torch.empty((capacity, 4), dtype=torch.int32)

    def open_set(self, planes, n, n_alloc, d, d_pad, limbs):
        return (self._sketches(planes, n, d, d_pad, limbs), n)

    def close_set(self, sset):
        pass

    def compare_block(self, sset, norms_sq, rb, re, cb, ce, flags, keep_mode, raw, n_raw):
        sk, n = sset
        cells = self.orc.pairwise_rows(sk, norms_sq.numpy(), row_begin=rb, row_end=re, chunk=192, threads=2)
        cells = cells[(cells["col"] >= cb) & (cells["col"] < ce)]
        arr = np.stack([cells[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
        if flags & 2:      # MVS_BLOCK_MIRROR_ALL
            arr = np.concatenate([arr, arr[:, [1, 0, 2, 3]]])
        arr = arr[np.random.default_rng(n_raw).permutation(len(arr))]      # the device appends in no particular order
        raw[n_raw:n_raw + len(arr)] = torch.from_numpy(arr)
        return n_raw + len(arr)

    def sort_cells(self, cells_in, n, cells_out):
        a = cells_in[:n].numpy()
        cells_out[:n] = torch.from_numpy(a[np.lexsort((a[:, 1], a[:, 0]))])


def _make(n, d, seed):
    from metagenome_vector_sketches_amd import synth
    from oracle import pyoracle as orc
    sk = synth.make_sketches_numpy(n, d, 3000, seed=seed, cluster=8)
    n2 = np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(r))) for r in sk])
    return sk, n2


def _worker(rank, world, port, n, d, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metagenome_vector_sketches_amd import parallel
    from oracle import pyoracle as orc_mod
    sk, n2 = _make(n, d, seed=99)
    b, e = parallel.shard_rows(n, world, rank)
    sc = parallel.ShardedComparison(OracleOps(), rank, world, dist)
    cells, cnt, info = sc.run(sk[b:e], n2[b:e], n)          # no output buffer -> plain rows x all-columns schedule
    cells2, cnt2, _ = sc.run(sk[b:e], n2[b:e], n)          # second step reuses the plane buffer
    assert cnt == cnt2 and np.array_equal(cells, cells2)
    np.save(os.path.join(out_dir, "cells_%d.npy" % rank), cells)
    # symmetric schedule: every unordered block pair once + exchange of the mirrored cells
    out = torch.empty((n * n, 4), dtype=torch.int32)
    _, cnt3, info3 = sc.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert info3.get("schedule") == "symmetric"
    plain = np.stack([cells[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
    assert cnt3 == cnt and np.array_equal(out[:cnt3].numpy(), plain), (rank, cnt3, cnt)
    # local rows arriving in parts (begin / feed / finish): the exchange of a part starts when it is fed; same cells
    local, n2l = torch.from_numpy(sk[b:e].copy()), n2[b:e]
    sc.begin(local, n2l, n)
    bounds = sc.part_bounds(n, 3)
    assert bounds[0][0] == 0 and bounds[-1][1] == (n + world - 1) // world and len(bounds) in (2, 3)
    for (p0, p1) in bounds:                                  # block coordinates: the same on every rank
        q0, q1 = min(p0, e - b), min(p1, e - b)
        sc.feed(p0, p1, int(np.abs(sk[b + q0:b + q1]).max()) if q1 > q0 else 0)
    _, cnt4, info4 = sc.finish(cells_out=out)
    assert cnt4 == cnt and np.array_equal(out[:cnt4].numpy(), plain) and info4["overlap"].startswith("exchange of a part")
    # a limb guess that does not hold on ONE rank only: every rank falls back to the plain exchange, same cells
    sc.begin(local, n2l, n, limbs_guess=1)
    sc.feed(0, (n + world - 1) // world, int(np.abs(sk[b:e]).max()) if rank == world - 1 else 1)
    _, cnt5, info5 = sc.finish(cells_out=out)
    assert cnt5 == cnt and np.array_equal(out[:cnt5].numpy(), plain) and "did not hold" in info5["overlap"]
    # a guess that is too LARGE is no reason to redo anything: one-limb data (max |v| <= 127) coded with the default guess
    # of two limbs keeps the overlapped exchange -- same cells as the plain run, which picks one limb
    small = np.clip(sk, -100, 100).astype(np.int32)
    n2s = np.array([orc_mod.norm_sq_from_text(orc_mod.format_norm(orc_mod.norm(r))) for r in small])
    out_s = torch.empty((n * n, 4), dtype=torch.int32)
    _, cnt_s, info_s = sc.run(small[b:e], n2s[b:e], n, cells_out=out_s)
    want_s = out_s[:cnt_s].numpy().copy()
    assert info_s["limbs"] == 1
    sc.begin(torch.from_numpy(small[b:e].copy()), n2s[b:e], n)
    for (p0, p1) in bounds:
        q0, q1 = min(p0, e - b), min(p1, e - b)
        sc.feed(p0, p1, int(np.abs(small[b + q0:b + q1]).max()) if q1 > q0 else 0)
    _, cnt_s2, info_s2 = sc.finish(cells_out=out_s)
    assert info_s2["limbs"] == 2 and info_s2["overlap"].startswith("exchange of a part"), info_s2
    assert cnt_s2 == cnt_s and np.array_equal(out_s[:cnt_s2].numpy(), want_s)
    # an output buffer that holds exactly this shard (the mirrored cells in flight live in internal buffers)
    tight = torch.empty((cnt, 4), dtype=torch.int32)
    _, cnt6, _ = sc.run(sk[b:e], n2[b:e], n, cells_out=tight)
    assert cnt6 == cnt and np.array_equal(tight.numpy(), plain)
    # a second, larger problem that pads to the same plane geometry: the gathered-norm buffer follows the row count
    n_big = n + 3
    sk2, n22 = _make(n_big, d, seed=99)
    b2, e2 = parallel.shard_rows(n_big, world, rank)
    cells7, cnt7, _ = sc.run(sk2[b2:e2], n22[b2:e2], n_big)
    np.save(os.path.join(out_dir, "cells_big_%d.npy" % rank), cells7)
    dist.barrier()
    dist.destroy_process_group()


def test_block_plan_covers_every_pair_once():
    from metagenome_vector_sketches_amd import parallel
    for world in (1, 2, 3, 4, 5, 8):
        for n in (1, 7, 64, 101):
            seen = np.zeros((n, n), dtype=np.int32)
            for rank in range(world):
                for (rb, re, cb, ce, flags) in parallel.block_plan(n, world, rank):
                    seen[rb:re, cb:ce] += 1
                    if flags & 2:                      # mirrored into the transposed block
                        seen[cb:ce, rb:re] += 1
                    assert parallel.shard_rows(n, world, rank)[0] <= rb and re <= parallel.shard_rows(n, world, rank)[1]
            assert np.all(seen == 1), (world, n)
    # per-rank work is balanced: G/2 blocks each (even G)
    work = [sum((re - rb) * (ce - cb) * (0.5 if f & 1 else 1.0) for rb, re, cb, ce, f in parallel.block_plan(8000, 8, r))
            for r in range(8)]
    assert max(work) == min(work) == 4 * 1000 * 1000


@pytest.mark.parametrize("n,world", [(96, 2), (101, 2), (101, 3), (130, 4)])   # uneven last shards -> padded blocks
def test_multi_rank_shards_equal_single_process(tmp_path, n, world):
    d, port = 256, 29500 + (os.getpid() + n + world) % 2000
    mp.spawn(_worker, args=(world, port, n, d, str(tmp_path)), nprocs=world, join=True)
    from metagenome_vector_sketches_amd import parallel
    from oracle import pyoracle as orc
    sk, n2 = _make(n, d, seed=99)
    want = orc.pairwise_rows(sk, n2, chunk=192)
    want = want[np.lexsort((want["col"], want["row"]))]
    got = np.concatenate([np.load(os.path.join(str(tmp_path), "cells_%d.npy" % r)) for r in range(world)])
    assert len(want) > n * 4
    assert np.array_equal(got, want)
    for r in range(world):
        b, e = parallel.shard_rows(n, world, r)
        part = np.load(os.path.join(str(tmp_path), "cells_%d.npy" % r))
        assert np.all((part["row"] >= b) & (part["row"] < e))
    sk2, n22 = _make(n + 3, d, seed=99)
    want2 = orc.pairwise_rows(sk2, n22, chunk=192)
    want2 = want2[np.lexsort((want2["col"], want2["row"]))]
    got2 = np.concatenate([np.load(os.path.join(str(tmp_path), "cells_big_%d.npy" % r)) for r in range(world)])
    assert np.array_equal(got2, want2)


def test_shard_rows_matches_reference_formula():
    from metagenome_vector_sketches_amd import parallel
    from oracle import pyoracle as orc
    for n, s in [(61, 2), (100, 8), (3, 8), (1000, 7), (0, 4)]:
        for k in range(s):
            assert parallel.shard_rows(n, s, k) == orc.shard_rows(n, s, k)
