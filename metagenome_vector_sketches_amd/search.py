"""Query-by-hashes search over a sketch DB folder: the exact, brute-force GPU counterpart of the reference's
FAISS path (src/jaccard.py).  `search_index` keeps the reference function's meaning -- for every query line
"name: h1 h2 ..." report the database samples whose Jaccard estimate exceeds j, best first -- but there is
nothing to build beforehand: the reference's `index` step (L2-normalise + IndexFlatIP, :18-61) has no
counterpart because the comparison kernel works on the integer sketches of vectors.bin directly.

  reference                                   here
  ---------                                   ----
  standalone_projection per query (:98-118)   Context.project_csr (same kernel as `sketch`)
  faiss IndexFlatIP.search, growing k (:131)  mvs_search_block: exact dots of every (query, sample) pair on the
                                              matrix cores, Jaccard test fused into the epilogue
  jaccard = ip*qn*nn/(nn^2+qn^2-ip*qn*nn)     the same formula with ip*qn*nn = dot/d exactly (:199)

Estimates agree with the float32 reference path to ~1e-6 relative (its inner products are float32).
"""
import os

import numpy as np

from . import _capi


def read_db(index_folder):
    """-> (names, norms float64, vectors int32/int16 [N, d])"""
    if not index_folder.endswith("/"):
        index_folder += "/"
    with open(index_folder + "dimension.txt") as f:
        d = int(f.readline().strip())
    dtype = "int32"
    if os.path.exists(index_folder + "dtype.txt"):
        with open(index_folder + "dtype.txt") as f:
            dtype = f.readline().strip() or "int32"
    names, norms = [], []
    with open(index_folder + "vector_norms.txt") as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split()
            names.append(parts[0])
            norms.append(float(parts[1]))
    # mapped, not read: search_index hands it to the device in row chunks straight from the page cache
    vec = np.memmap(index_folder + "vectors.bin", dtype="<i2" if dtype == "int16" else "<i4", mode="r")
    return names, np.array(norms, dtype=np.float64), vec.reshape(-1, d)


def read_queries(query_file):
    """src/jaccard.py:75-90: one 'name: hashes' record per non-empty line, exactly one ':'"""
    names, lists = [], []
    with open(query_file) as f:
        for line in f:
            line = line.strip()
            if not line:
                continue
            parts = line.split(":")
            if len(parts) != 2:
                raise ValueError("ERROR 332: %s %s %d" % (query_file, line[:20], len(parts)))
            names.append(parts[0].strip())
            lists.append(np.array(sorted(set(int(t) for t in parts[1].split())), dtype=np.uint64))
    return names, lists


def search_index(index_folder, query_file, j, ctx=None, verbose=True):
    """-> list of (query_index, neighbor_id, jaccard), per query sorted by jaccard descending
    (what src/jaccard.py:63-224 returns)."""
    import torch
    own = ctx is None
    if own:
        ctx = _capi.Context(0)
    try:
        names, norms, vectors = read_db(index_folder)
        n, d = vectors.shape
        qnames, lists = read_queries(query_file)
        nq = len(lists)
        if nq == 0:
            return []
        offs = np.zeros(nq + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(x) for x in lists])
        flat = np.concatenate(lists) if offs[-1] else np.zeros(0, dtype=np.uint64)
        dev = torch.device("cuda", ctx.device)
        if own:
            ctx.set_stream(torch.cuda.current_stream(dev))          # one stream for torch's copies and the kernels
        q_sk = torch.empty((nq, d), dtype=torch.int32, device=dev)
        q_ss = torch.empty(nq, dtype=torch.int64, device=dev)
        q_max = ctx.project_csr_stats(flat, offs, d, q_sk, q_ss)
        # the database goes up once, in row chunks, re-coded for two limbs unless the queries already need more; a chunk
        # reports its largest |v| with the same upload, and only if that asks for more limbs is the set built again
        # (the way pairwise_comp_optimized loads vectors.bin)
        limbs = max(2, _capi.limbs_for_max_abs(q_max))
        chunk = max(1, (1 << 30) // (d * vectors.dtype.itemsize))
        while True:
            sset = ctx.sketch_set_alloc(n + nq, d, limbs)
            need = limbs
            for r0 in range(0, n, chunk):
                need = max(need, _capi.limbs_for_max_abs(sset.fill_stats(vectors[r0:r0 + chunk], r0)))
                if need > limbs:
                    break
            if need == limbs:
                break
            sset.close()
            limbs = need
        sset.fill(q_sk, n)
        qn2 = q_ss.cpu().numpy().astype(np.float64) / d             # query_norm^2 (:120-121, exact here)
        n2 = torch.from_numpy(np.concatenate([norms * norms, qn2])).to(dev)
        cap = max(1 << 16, 64 * nq)
        while True:
            cells = torch.empty((cap, 4), dtype=torch.int32, device=dev)
            try:
                cnt = ctx.search_block(sset, n2, j, n, n + nq, 0, n, cells)
                break
            except _capi.MvsError as e:
                if e.code != _capi.MVS_E_CAPACITY:
                    raise
                cap *= 4
        ctx.synchronize()
        hits = cells[:cnt].cpu().numpy()
        sset.close()
        out = []
        for qi in range(nq):
            if qn2[qi] == 0:                                        # :204-205 query_norm == 0 -> skipped
                continue
            mine = hits[hits[:, 0] == n + qi]
            inter = mine[:, 2].astype(np.float64) / d
            nn2 = norms[mine[:, 1]] ** 2
            jac = inter / (nn2 + qn2[qi] - inter)                   # :199
            order = np.argsort(-jac, kind="stable")
            if verbose:
                print("Query %d:" % qi)
            for rank, k in enumerate(order):
                if not jac[k] > j:
                    continue
                nid = names[mine[k, 1]]
                if verbose:
                    ip = inter[k] / (np.sqrt(qn2[qi]) * norms[mine[k, 1]])
                    print("  Neighbor %d: %s (jaccard: %.4f), inner_product: %.4f %s %s"
                          % (rank, nid, jac[k], ip, norms[mine[k, 1]], np.sqrt(qn2[qi])))
                out.append((qi, nid, float(jac[k])))
        return out
    finally:
        if own:
            ctx.close()


def main():
    import argparse
    ap = argparse.ArgumentParser(description="search a sketch DB for samples similar to query hash sets")
    ap.add_argument("index_folder")
    ap.add_argument("query_file")
    ap.add_argument("-j", "--jaccard", type=float, default=0.1)
    a = ap.parse_args()
    search_index(a.index_folder, a.query_file, a.jaccard)


if __name__ == "__main__":
    main()
