"""CPU, world_size 2-4, gloo: the row-sharded comparison (metagenome_vector_sketches_amd/parallel.py).

The orchestration code is the product's; the numeric back end is replaced by a stand-in built on the
oracle (tests may use the oracle; the product may not), so what is verified here is the sharding, the storage layout
the all-gathers assemble (per-rank blocks padded to 256 rows), the block plan of the symmetric schedule, the ORDER of the
step (a filter launch may only read columns whose coarse rows and statistics have landed; the re-check needs the limb
planes), the routing / exchange of the mirrored cells with its header protocol (status, largest |v|, overflow), and the
padding of an uneven last shard: the union of the ranks' shards must equal the single-process result cell for cell."""
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
    """CPU stand-in for parallel.GpuOps with the same plane layout as the HIP limb-split kernel
    (planes[(row*limbs + limb)*d_pad + k], signed base-256 digits) and the same call protocol as the block-plan entry
    points of the C ABI.  prepare_rows() leaves a marker in the coarse / statistics buffers instead of real filter
    inputs; plan_filter() insists on finding it for every row and column it is asked to compare."""

    def __init__(self, coll=None):
        from oracle import pyoracle
        self.orc = pyoracle
        self.log = []
        self.coll = coll                  # lazy collectives: the re-check must not run while exchanges are outstanding
        self.stale_next = 0               # plans to report as MVS_PLAN_STALE
        self.exact = False                # True: a plan without a filter -- its blocks are compared when they are handed over

    # ---- geometry / buffers ----
    def layout(self, n_total, world):
        rps = (n_total + world - 1) // world
        return rps, max(256, (rps + 255) // 256 * 256)

    def max_abs(self, sk):
        return int(np.abs(np.asarray(sk, dtype=np.int64)).max()) if sk.size else 0

    def limbs_for(self, m):
        return 1 if m <= 127 else 2 if m <= 32639 else 3 if m <= 8355711 else 4

    def limb_geometry(self, n, d, limbs):
        n_alloc = (n + 255) // 256 * 256 + 256
        d_pad = (d + 127) // 128 * 128
        return n_alloc, d_pad, n_alloc * limbs * d_pad

    def new_planes(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.int8)

    def new_bytes(self, nbytes):
        return torch.zeros(nbytes, dtype=torch.uint8)

    def new_cells(self, capacity):
        return torch.zeros((capacity, 4), dtype=torch.int32)

    def new_counter(self, rows=0):
        return torch.zeros(1, dtype=torch.int64)

    def zero_count(self):
        return [0]

    def zero_(self, t):
        t.zero_()

    def to_device(self, a):
        return torch.from_numpy(np.ascontiguousarray(a))

    def to_host_cells(self, cells, n):
        a = cells[:n].numpy()
        out = np.zeros(n, dtype=[("row", "<i4"), ("col", "<i4"), ("dot", "<i4"), ("q", "<i4")])
        for k, name in enumerate(("row", "col", "dot", "q")):
            out[name] = a[:, k]
        return out

    def limb_split(self, sk, limbs, planes, d_pad, row_offset):
        v = np.asarray(sk, dtype=np.int64).copy()
        n, d = v.shape
        view = planes.numpy().reshape(-1, limbs, d_pad)
        for l in range(limbs):
            digit = ((v + 128) % 256) - 128
            view[row_offset:row_offset + n, l, :d] = digit.astype(np.int8)
            v = (v - digit) // 256

    # ---- sets and their derived data ----
    def open_set(self, planes, n, n_alloc, d, d_pad, limbs, coarse_fm, stats):
        return {"planes": planes, "n": n, "n_alloc": n_alloc, "d": d, "d_pad": d_pad, "limbs": limbs, "coarse": coarse_fm,
                "stats": stats}

    def close_set(self, sset):
        pass

    def touch_set(self, sset):
        sset["coarse"].zero_()           # a new step: nothing has landed yet
        sset["stats"].zero_()

    def prepare_rows(self, sset, first, count):
        """the filter's inputs of these rows.  The stand-in keeps a real coarse plane (row-major, c + 128 so that a byte that has
        landed is never 0) with the radix that just avoids clamping in the first four statistics bytes: what the receiver of
        a low limb needs to rebuild the high one (csrc/mvs_pairwise.hip "The high limb on the wire")"""
        assert first % 16 == 0 and count % 16 == 0
        d_pad, limbs = sset["d_pad"], sset["limbs"]
        view = sset["planes"].numpy().reshape(-1, limbs, d_pad).astype(np.int64)[first:first + count]
        v = sum(view[:, l, :] * (256 ** l) for l in range(limbs))
        mx = np.abs(v).max(axis=1)
        m = np.maximum(1, (mx + 126) // 127)
        c = np.clip(np.rint(v.astype(np.float32) * (np.float32(1.0) / m.astype(np.float32))[:, None]), -127, 127).astype(np.int64)
        sset["coarse"].numpy()[first * d_pad:(first + count) * d_pad] = (c + 128).astype(np.uint8).reshape(-1)
        stats = sset["stats"].numpy()[first * 16:(first + count) * 16].reshape(count, 16)
        stats[:, :4] = m.astype("<i4").view(np.uint8).reshape(count, 4)
        stats[:, 4:] = 7

    def wire_rows(self, planes, lo_wire, d_pad, first, count):
        lo_wire.numpy().reshape(-1, d_pad)[first:first + count] = planes.numpy().reshape(-1, 2, d_pad)[first:first + count, 0, :]

    def planes_from_wire(self, sset, lo_wire, first, count):
        """numpy restatement of k_planes_from_wire: v = the value congruent to the low limb mod 256 near m c"""
        assert sset["limbs"] == 2 and self._landed(sset, first, first + count), "low limbs rebuilt before coarse plane / statistics arrived"
        d_pad = sset["d_pad"]
        l0 = lo_wire.numpy().reshape(-1, d_pad)[first:first + count].astype(np.int64)
        c = sset["coarse"].numpy()[first * d_pad:(first + count) * d_pad].reshape(count, d_pad).astype(np.int64) - 128
        m = sset["stats"].numpy()[first * 16:(first + count) * 16].reshape(count, 16)[:, :4].copy().view("<i4").reshape(count).astype(np.int64)
        edge = (127 * m - (m + 1) // 2 + 127)[:, None]
        t = np.where(c == 127, edge, np.where(c == -127, -edge, m[:, None] * c))
        v = t + (((l0 - t) + 128) % 256 - 128)
        planes = sset["planes"].numpy().reshape(-1, 2, d_pad)
        planes[first:first + count, 0, :] = l0.astype(np.int8)
        planes[first:first + count, 1, :] = ((v - l0) // 256).astype(np.int8)
        self.log.append("rebuilt %d" % first)

    def recode_rows(self, sset, sketches, limbs, planes, d_pad, first, count):
        if sketches.shape[0]:
            self.limb_split(sketches, limbs, planes, d_pad, first)
        self.prepare_rows(sset, first, count)

    def _landed(self, sset, r0, r1):
        d_pad = sset["d_pad"]
        c = sset["coarse"].numpy()[r0 * d_pad:r1 * d_pad].reshape(r1 - r0, d_pad)
        s = sset["stats"].numpy()[r0 * 16:r1 * 16].reshape(r1 - r0, 16)
        return bool(np.all(c[:, 0] != 0) and np.all(c[:, -1] != 0) and np.all(s[:, 4:] == 7))

    def _sketches(self, sset):
        planes, n, d, d_pad, limbs = sset["planes"], sset["n"], sset["d"], sset["d_pad"], sset["limbs"]
        view = planes.numpy().reshape(-1, limbs, d_pad).astype(np.int64)
        return sum(view[:n, l, :d] * (256 ** l) for l in range(limbs)).astype(np.int32)

    # ---- block plans ----
    def plan_begin(self, sset, norms_sq, f0, f1, mirror_outside, raw, keep_mode):
        assert f0 % 256 == 0 and f1 % 256 == 0
        assert self._landed(sset, f0, f1), "the frame's own rows were not prepared"
        self.plan = {"sset": sset, "n2": norms_sq, "f0": f0, "f1": f1, "mirror": mirror_outside, "raw": raw, "blocks": []}
        self.log.append("begin")

    def plan_exact_mode(self):
        return self.exact

    def plan_wire(self, lo_wire):
        """the device rebuilds the rows its candidates name, inside mvs_plan_finish; the stand-in, which compares everything
        with everything, rebuilds every peer's rows (after the exchanges have been waited for: finish checks)"""
        p = self.plan
        P = p["f1"] - p["f0"]
        for q in range(p["sset"]["n"] // P):
            if q * P != p["f0"]:
                self.planes_from_wire(p["sset"], lo_wire, q * P, P)

    def plan_filter(self, blocks):
        p = self.plan
        for (rb, re, cb, ce) in blocks:
            if self.exact and self.coll is not None and not (cb >= p["f0"] and ce <= p["f1"]):
                assert self.coll.outstanding() == 0, "exact kernel on a peer's block while its limb planes are still on the way"
            assert p["f0"] <= rb <= re <= p["f1"] and rb % 256 == 0 and cb % 256 == 0
            inside = cb >= p["f0"] and ce <= p["f1"]
            assert inside or ce <= p["f0"] or cb >= p["f1"]
            assert self._landed(p["sset"], cb, ce), "filter launched on columns [%d,%d) that have not arrived" % (cb, ce)
            p["blocks"].append((rb, re, cb, ce))
        self.log.append("filter %d" % len(blocks))

    def plan_finish(self):
        p = self.plan
        sset = p["sset"]
        if self.coll is not None:
            assert self.coll.outstanding() == 0, "re-check while %d exchanges have not been waited for" % self.coll.outstanding()
        sk = self._sketches(sset)                       # storage rows (padding rows: zeros)
        n2 = p["n2"].numpy()[:sset["n"]]
        out = []
        for (rb, re, cb, ce) in p["blocks"]:
            cells = self.orc.pairwise_rows(sk, n2, row_begin=rb, row_end=re, chunk=192, threads=2)
            cells = cells[(cells["col"] >= cb) & (cells["col"] < ce)]
            arr = np.stack([cells[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)
            inside = cb >= p["f0"] and ce <= p["f1"]
            if inside:                                  # symmetric schedule: one triangle computed, the other mirrored
                upper = arr[arr[:, 1] > arr[:, 0]]
                arr = np.concatenate([arr[arr[:, 1] == arr[:, 0]], upper, upper[:, [1, 0, 2, 3]]])
            elif p["mirror"]:
                arr = np.concatenate([arr, arr[:, [1, 0, 2, 3]]])
            out.append(arr)
        arr = np.concatenate(out) if out else np.zeros((0, 4), dtype=np.int32)
        arr = arr[np.random.default_rng(len(arr)).permutation(len(arr))]      # the device appends in no particular order
        raw = p["raw"]
        keep = min(len(arr), raw.shape[0])
        raw[:keep] = torch.from_numpy(arr[:keep])
        self.log.append("finish")
        if self.stale_next:                             # a plan that ran on the previous step's sizes and found them too small
            self.stale_next -= 1
            raw[:keep] = torch.from_numpy(arr[:keep][::-1].copy()) // 2          # whatever it appended is not to be used
            from metagenome_vector_sketches_amd import _capi as capi
            return [capi.PLAN_STALE + 5]
        return [len(arr)]                               # the count may exceed the capacity, as on the device

    # ---- kept cells -> shard ----
    def cells_route(self, raw, d_n_raw, P, rps, n_total, own, own_out, d_own, send, cap_f, status, max_abs):
        total = int(d_n_raw[0])
        n = min(total, raw.shape[0])
        a = raw[:n].numpy().astype(np.int64)
        br, orow, bc, ocol = a[:, 0] // P, a[:, 0] % P, a[:, 1] // P, a[:, 1] % P
        row, col = br * rps + orow, bc * rps + ocol
        valid = (orow < rps) & (ocol < rps) & (row < n_total) & (col < n_total)
        g = np.stack([row, col, a[:, 2], a[:, 3]], axis=1).astype(np.int32)
        mine = valid & (row >= own[0]) & (row < own[1])
        mc = g[mine]
        keep = min(len(mc), own_out.shape[0])
        own_out[:keep] = torch.from_numpy(mc[:keep])
        d_own[0] = len(mc)
        self._own_rows = own_out         # (the report looks at it for the widest row)
        fc = g[valid & ~mine]
        hdr = np.zeros(8, dtype=np.int64)
        hdr[:5] = [len(fc), status, max_abs, total, raw.shape[0]]
        send[:64] = torch.from_numpy(hdr.view(np.uint8).copy())
        k = min(len(fc), cap_f)
        if k:
            send[64:64 + 16 * k] = torch.from_numpy(np.ascontiguousarray(fc[:k]).view(np.uint8).reshape(-1).copy())

    def cells_collect(self, recv, world, rank, cap_f, own, own_out, d_own):
        stride = 64 + 16 * cap_f
        n_own = int(d_own[0])
        for p in range(world):
            if p == rank:
                continue
            buf = recv[p * stride:(p + 1) * stride].numpy()
            n = min(int(buf[:8].view(np.int64)[0]), cap_f)
            cells = buf[64:64 + 16 * n].view(np.int32).reshape(n, 4)
            got = cells[(cells[:, 0] >= own[0]) & (cells[:, 0] < own[1])]
            room = max(0, min(len(got), own_out.shape[0] - n_own))
            if room:
                own_out[n_own:n_own + room] = torch.from_numpy(got[:room].copy())
            n_own += len(got)
        d_own[0] = n_own

    def sort_cells_ahead(self, cells_in, cells_out, own, d_own):
        """the sort before the report: the count is the one the route / collect left in the state block"""
        n = min(int(d_own[0]), cells_in.shape[0])
        a = cells_in[:n].numpy()
        a = a[np.lexsort((a[:, 1], a[:, 0]))]
        k = min(n, cells_out.shape[0])
        cells_out[:k] = torch.from_numpy(a[:k].copy())
        self.log.append("sorted ahead")

    def cells_report(self, recv, world, cap_f, own, d_own):
        stride = 64 + 16 * cap_f
        heads = []
        for p in range(world):
            h = recv[p * stride:p * stride + 64].numpy().view(np.int64)
            heads.append(tuple(int(x) for x in h[:5]))
        n_own = min(int(d_own[0]), self._own_rows.shape[0]) if getattr(self, "_own_rows", None) is not None else 0
        widest = int(np.bincount(self._own_rows[:n_own, 0].numpy() - own[0]).max()) if n_own else 0
        return int(d_own[0]), heads, widest

    def sort_cells(self, cells_in, n, cells_out, own=None, d_own=None, max_row=None):
        a = cells_in[:n].numpy()
        cells_out[:n] = torch.from_numpy(a[np.lexsort((a[:, 1], a[:, 0]))])


def lazy_collectives(dist_mod, rank, world):
    """torch.distributed collectives whose exchanges happen as LATE as the step allows: submit() only queues the calls, a
    handle's wait() runs everything queued up to it.  With them a step that reads a buffer before waiting for the exchange
    that fills it finds the buffer empty (OracleOps checks the markers) -- on a GPU the same mistake is a race."""
    from metagenome_vector_sketches_amd import parallel

    class Lazy(parallel.TorchCollectives):
        def __init__(self):
            super().__init__(dist_mod, rank, world)
            self.queue = []

        def submit(self, fn):
            self.queue.append(fn)
            me, upto = self, len(self.queue)

            class H:
                def wait(self):
                    while me.queue and upto > me.done:
                        me.queue[me.done]()
                        me.done += 1
            return H()
        done = 0

        def outstanding(self):
            return len(self.queue) - self.done
    return Lazy()


def _make(n, d, seed):
    from metagenome_vector_sketches_amd import synth
    from oracle import pyoracle as orc
    sk = synth.make_sketches_numpy(n, d, 3000, seed=seed, cluster=8)
    n2 = np.array([orc.norm_sq_from_text(orc.format_norm(orc.norm(r))) for r in sk])
    return sk, n2


def _plain(cells):
    return np.stack([cells[k] for k in ("row", "col", "dot", "q")], axis=1).astype(np.int32)


def _worker(rank, world, port, n, d, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from metagenome_vector_sketches_amd import parallel, _capi
    from oracle import pyoracle as orc_mod
    sk, n2 = _make(n, d, seed=99)
    b, e = parallel.shard_rows(n, world, rank)
    ops = OracleOps()
    sc = parallel.ShardedComparison(ops, rank, world, dist)
    assert sc.symmetric
    cells, cnt, info = sc.run(sk[b:e], n2[b:e], n)          # no output buffer -> the shard comes back as a host array
    assert info["schedule"] == ("symmetric" if world > 1 else "rows x all columns")
    # the order of the step: the diagonal block's filter first, then one launch per arrived chunk, then the rest
    steps = [x for x in ops.log if not x.startswith("rebuilt")]
    assert steps[0] == "begin" and steps[1] == "filter 1" and steps[-1] == "finish"
    # two bytes per entry on the wire: the other ranks' limb planes were rebuilt from low limbs + coarse plane, once per peer
    assert info["wire"].startswith("coarse plane + low limbs") == (world > 1)
    rebuilt = [int(x.split()[1]) for x in ops.log if x.startswith("rebuilt")]
    has_peers = len(parallel.block_plan(world, rank, ops.layout(n, world)[1])) > 1      # (else nobody's limb planes are read)
    assert sorted(set(r // ops.layout(n, world)[1] for r in rebuilt)) == ([p for p in range(world) if p != rank] if has_peers else [])
    if len(parallel.block_plan(world, rank, ops.layout(n, world)[1])) > 1:
        assert len([x for x in ops.log if x.startswith("filter")]) >= 2
    cells2, cnt2, info_2 = sc.run(sk[b:e], n2[b:e], n)      # second step reuses the gathered buffers
    assert cnt == cnt2 and np.array_equal(cells, cells2) and not info_2["sorted_ahead"]      # (no caller's buffer to sort into)
    np.save(os.path.join(out_dir, "cells_%d.npy" % rank), cells)
    plain = _plain(cells)
    # into a caller's buffer; the plain rows x all-columns schedule (no mirroring, no cell exchange) gives the same shard
    out = torch.empty((n * n, 4), dtype=torch.int32)
    _, cnt3, info3 = sc.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert cnt3 == cnt and np.array_equal(out[:cnt3].numpy(), plain), (rank, cnt3, cnt)
    # the previous steps' rows held at most 64 cells: this one sorted in front of its host synchronisation
    widest = int(np.bincount(plain[:, 0] - b).max()) if cnt else 0
    assert info3["sorted_ahead"] == (widest <= 64) and ("sorted ahead" in ops.log) == (widest <= 64), (widest, info3["sorted_ahead"], ops.log[-6:])
    sc_rows = parallel.ShardedComparison(OracleOps(), rank, world, dist)
    sc_rows.symmetric = False
    _, cnt3b, info3b = sc_rows.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert info3b["schedule"] == "rows x all columns" and info3b["exchanged_cells"] == 0
    assert cnt3b == cnt and np.array_equal(out[:cnt3b].numpy(), plain)
    # exchanges that complete as late as the step allows: nothing may be read before its handle has been waited for
    lazy = lazy_collectives(dist, rank, world)
    sc_lazy = parallel.ShardedComparison(OracleOps(lazy), rank, world, collectives=lazy)
    _, cnt_l, _ = sc_lazy.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert cnt_l == cnt and np.array_equal(out[:cnt_l].numpy(), plain)
    ops_x = OracleOps(lazy)
    ops_x.exact = True                    # a plan without a filter reads the peers' limb planes when a block is handed over
    sc_x = parallel.ShardedComparison(ops_x, rank, world, collectives=lazy)
    _, cnt_x, _ = sc_x.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert cnt_x == cnt and np.array_equal(out[:cnt_x].numpy(), plain)
    sc_lazy.begin(torch.from_numpy(sk[b:e].copy()), n2[b:e], n)
    for (p0, p1) in sc_lazy.part_bounds(n, 2):
        q0, q1 = min(p0, e - b), min(p1, e - b)
        sc_lazy.feed(p0, p1, int(np.abs(sk[b + q0:b + q1]).max()) if q1 > q0 else 0)
    _, cnt_l, _ = sc_lazy.finish(cells_out=out)
    assert cnt_l == cnt and np.array_equal(out[:cnt_l].numpy(), plain)
    # local rows arriving in parts (begin / feed / finish): the exchange of a part starts when it is fed; same cells
    local, n2l = torch.from_numpy(sk[b:e].copy()), n2[b:e]
    sc.begin(local, n2l, n)
    bounds = sc.part_bounds(n, 3)
    assert bounds[0][0] == 0 and bounds[-1][1] == ops.layout(n, world)[1]
    for (p0, p1) in bounds:                                  # storage rows of the block: the same on every rank
        q0, q1 = min(p0, e - b), min(p1, e - b)
        sc.feed(p0, p1, int(np.abs(sk[b + q0:b + q1]).max()) if q1 > q0 else 0)
    _, cnt4, info4 = sc.finish(cells_out=out)
    assert cnt4 == cnt and np.array_equal(out[:cnt4].numpy(), plain)
    # a limb guess that does not hold on ONE rank only: every rank learns it from the exchange's headers and redoes the step
    sc.begin(local, n2l, n, limbs_guess=1)
    sc.feed(0, ops.layout(n, world)[1], int(np.abs(sk[b:e]).max()) if rank == world - 1 else 1)
    _, cnt5, info5 = sc.finish(cells_out=out)
    assert cnt5 == cnt and np.array_equal(out[:cnt5].numpy(), plain)
    assert "did not hold" in info5["overlap"] or world == 1 and rank != world - 1, info5
    # a guess that is too LARGE is no reason to redo anything: one-limb data (max |v| <= 127) coded with two limbs is exact
    small = np.clip(sk, -100, 100).astype(np.int32)
    n2s = np.array([orc_mod.norm_sq_from_text(orc_mod.format_norm(orc_mod.norm(r))) for r in small])
    out_s = torch.empty((n * n, 4), dtype=torch.int32)
    _, cnt_s, info_s = sc.run(small[b:e], n2s[b:e], n, cells_out=out_s, limbs_guess=1)
    want_s = out_s[:cnt_s].numpy().copy()
    assert info_s["limbs"] == 1
    _, cnt_s2, info_s2 = sc.run(small[b:e], n2s[b:e], n, cells_out=out_s)
    assert info_s2["limbs"] == 2 and "did not hold" not in info_s2["overlap"], info_s2
    assert cnt_s2 == cnt_s and np.array_equal(out_s[:cnt_s2].numpy(), want_s)
    # values beyond what a low limb pins (|v| > 32 004, still two limbs) on ONE rank: the rebuilt planes are not trusted, every
    # rank learns it from the headers, the step is redone with the limb planes themselves on the wire -- for good
    big = sk.copy()
    big[n - 1, 1] = 32500
    n2b = np.array([orc_mod.norm_sq_from_text(orc_mod.format_norm(orc_mod.norm(r))) for r in big])
    sc_ref = parallel.ShardedComparison(OracleOps(), rank, world, dist)
    sc_ref.wire = False
    out_b = torch.empty((n * n, 4), dtype=torch.int32)
    _, cnt_ref, info_ref = sc_ref.run(big[b:e], n2b[b:e], n, cells_out=out_b)
    want_b = out_b[:cnt_ref].numpy().copy()
    assert info_ref["wire"] == "coarse plane + limb planes"
    sc_b = parallel.ShardedComparison(OracleOps(), rank, world, dist)
    _, cnt_b, info_b = sc_b.run(big[b:e], n2b[b:e], n, cells_out=out_b)
    assert cnt_b == cnt_ref and np.array_equal(out_b[:cnt_b].numpy(), want_b)
    assert ("beyond" in info_b["overlap"]) == (world > 1) and sc_b.wire == (world == 1)
    # ONE rank's plan ran ahead of its read-backs on sizes that did not hold (mvs_plan_finish, plan_speculate): its header says
    # so, it compares again, everybody exchanges again
    ops_st = OracleOps()
    ops_st.stale_next = 1 if rank == world - 1 else 0
    sc_st = parallel.ShardedComparison(ops_st, rank, world, dist)
    _, cnt_st, info_st = sc_st.run(sk[b:e], n2[b:e], n, cells_out=out)
    assert cnt_st == cnt and np.array_equal(out[:cnt_st].numpy(), plain)
    assert info_st["plan_respeculated"] == 1 and ops_st.log.count("finish") == (2 if rank == world - 1 else 1)
    # an output buffer that holds exactly this shard: the raw list and the exchange buffers start too small for the mirrored
    # cells in flight and are regrown from what the headers report (every rank goes through the same retries)
    sc_tight = parallel.ShardedComparison(OracleOps(), rank, world, dist)
    sc_tight._cap_f = 4
    tight = torch.empty((max(cnt, 1), 4), dtype=torch.int32)
    _, cnt6, _ = sc_tight.run(sk[b:e], n2[b:e], n, cells_out=tight)
    assert cnt6 == cnt and np.array_equal(tight[:cnt].numpy(), plain)
    # a buffer that cannot hold the shard: MVS_E_CAPACITY with the number of cells there are
    if cnt > 1:
        with pytest.raises(_capi.MvsError) as ei:
            sc_tight.run(sk[b:e], n2[b:e], n, cells_out=torch.empty((cnt - 1, 4), dtype=torch.int32))
        assert ei.value.code == _capi.MVS_E_CAPACITY and ei.value.needed == cnt
    # no output buffer, and ONE rank's default shard capacity is too small while its raw list fits: that rank
    # routes and collects again into a larger buffer on its own -- the kept cells are still in the raw list, the peers'
    # mirror images still in the exchange buffer -- and no peer takes part (a collective repeat only one rank knows about would
    # hang); with every capacity too small on one rank the collective regrowth and the local one follow each other
    for raw_small in (False, True):
        ops_o = OracleOps()
        sc_o = parallel.ShardedComparison(ops_o, rank, world, dist)
        if rank == world - 1:
            sc_o.own_capacity_default = lambda rows: 3
        sc_o.raw_capacity_default = (lambda rows: 3) if (raw_small and rank == world - 1) else (lambda rows: n * n)
        cells_o, cnt_o, info_o = sc_o.run(sk[b:e], n2[b:e], n)
        assert cnt_o == cnt and np.array_equal(cells_o, cells), (rank, cnt_o, cnt, info_o)
        assert info_o.get("own_regrown", 0) == (1 if (rank == world - 1 and cnt > 3) else 0), info_o
        # (the raw list keeps 1024 cells of slack: whether it overflows as well depends on the shard)
        assert ops_o.log.count("finish") in ((1, 2) if (raw_small and rank == world - 1) else (1,)), ops_o.log
    # a second, larger problem: other geometry, other buffers
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
        for P in (256, 512, 768, 1280):
            t = P // 256                                     # in units of tiles
            seen = np.zeros((world * t, world * t), dtype=np.int32)
            for rank in range(world):
                plan = parallel.block_plan(world, rank, P)
                assert plan[0] == (rank * P, (rank + 1) * P, rank * P, (rank + 1) * P)     # the diagonal block comes first
                for k, (rb, re, cb, ce) in enumerate(plan):
                    assert rank * P <= rb <= re <= (rank + 1) * P and all(x % 256 == 0 for x in (rb, re, cb, ce))
                    seen[rb // 256:re // 256, cb // 256:ce // 256] += 1
                    if k > 0:                                # mirrored into the transposed block
                        seen[cb // 256:ce // 256, rb // 256:re // 256] += 1
            assert np.all(seen == 1), (world, P)
            # the plain schedule: every rank its rows against everything, nothing mirrored
            seen[:] = 0
            for rank in range(world):
                for (rb, re, cb, ce) in parallel.block_plan(world, rank, P, symmetric=False):
                    seen[rb // 256:re // 256, cb // 256:ce // 256] += 1
            assert np.all(seen == 1)
    # per-rank work is balanced: G/2 blocks each (even G, an even number of tile rows)
    work = [sum((re - rb) * (ce - cb) * (0.5 if k == 0 else 1.0) for k, (rb, re, cb, ce) in enumerate(parallel.block_plan(8, r, 1024)))
            for r in range(8)]
    assert max(work) == min(work) == 4 * 1024 * 1024
    # chunks: multiples of 256, in order, covering the block; clipping keeps the rows and cuts the columns per rank block
    assert parallel.chunk_bounds(1280, 2) == [(0, 512), (512, 1280)] and parallel.chunk_bounds(256, 4) == [(0, 256)]
    plan = parallel.block_plan(4, 3, 512)
    got = parallel.clip_blocks(plan[1:], 512, 256, 512)
    assert got == [(1536, 2048, 256, 512), (1536, 2048, 768, 1024)]      # rank 0's block, second half of rank 1's


@pytest.mark.parametrize("n,world", [(96, 2), (101, 2), (101, 3), (130, 4), (700, 2)])   # uneven last shards -> padded blocks
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


def test_single_rank_goes_through_the_same_step():
    """world 1: no collectives, the same plan / route / report path"""
    from metagenome_vector_sketches_amd import parallel
    from oracle import pyoracle as orc
    n, d = 300, 128
    sk, n2 = _make(n, d, seed=5)
    sc = parallel.ShardedComparison(OracleOps(), 0, 1)
    cells, cnt, info = sc.run(sk, n2, n)
    want = orc.pairwise_rows(sk, n2, chunk=192)
    want = want[np.lexsort((want["col"], want["row"]))]
    assert cnt == len(want) and np.array_equal(cells, want) and info["collectives"] == "none"
    # a default shard capacity that is too small while the raw list fits: the raw buffer -- which holds the plan's kept cells --
    # must survive the regrowth of the shard buffer (it was replaced by an uninitialised one without a new comparison)
    for own_cap in (5, cnt - 1):
        ops = OracleOps()
        sc2 = parallel.ShardedComparison(ops, 0, 1)
        sc2.own_capacity_default = lambda rows, c=own_cap: c
        sc2.raw_capacity_default = lambda rows: cnt
        cells2, cnt2, info2 = sc2.run(sk, n2, n)
        assert cnt2 == cnt and np.array_equal(cells2, want) and info2["own_regrown"] == 1 and ops.log.count("finish") == 1


def test_shard_rows_matches_reference_formula():
    from metagenome_vector_sketches_amd import parallel
    from oracle import pyoracle as orc
    for n, s in [(61, 2), (100, 8), (3, 8), (1000, 7), (0, 4)]:
        for k in range(s):
            assert parallel.shard_rows(n, s, k) == orc.shard_rows(n, s, k)
