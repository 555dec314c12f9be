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


class SearchIndex:
    """A sketch DB resident on the device: load once (`SearchIndex(folder)`), query many times (`search`).  What the
    reference keeps in faiss.index (L2-normalised float copies, src/jaccard.py:18-61) is here the integer sketches
    themselves, re-coded as limb planes; rows [n, n + max_queries) of the set are scratch for the queries' sketches."""

    def __init__(self, index_folder, ctx=None, max_queries=1024):
        import torch
        self._own = ctx is None
        self.ctx = _capi.Context(0) if ctx is None else ctx
        self.names, self.norms, vectors = read_db(index_folder)
        self.n, self.d = vectors.shape
        self.max_queries = int(max_queries)
        self.dev = torch.device("cuda", self.ctx.device)
        if self._own:
            self.ctx.set_stream(torch.cuda.current_stream(self.dev))  # one stream for torch's copies and the kernels
        self._vectors = vectors
        self.sset = None
        self.limbs = 0
        self._hits = None                                            # grow-only hit buffer
        self._load(2)

    def _load(self, limbs):
        """the database goes up once, in row chunks, re-coded for `limbs` limbs; a chunk reports its largest |v| with the
        same upload, and only if that asks for more limbs is the set built again (the way pairwise_comp_optimized loads
        vectors.bin)"""
        n, d, vectors = self.n, self.d, self._vectors
        chunk = max(1, (1 << 30) // (d * vectors.dtype.itemsize))
        while True:
            if self.sset is not None:
                self.sset.close()
            self.sset = self.ctx.sketch_set_alloc(n + self.max_queries, d, limbs)
            need = limbs
            for r0 in range(0, n, chunk):
                need = max(need, _capi.limbs_for_max_abs(self.sset.fill_stats(vectors[r0:r0 + chunk], r0)))
                if need > limbs:
                    break
            if need == limbs:
                break
            limbs = need
        self.limbs = limbs

    def close(self):
        if self.sset is not None:
            self.sset.close()
            self.sset = None
        if self._own and self.ctx is not None:
            self.ctx.close()
        self.ctx = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def search(self, query_file, j, verbose=True):
        """-> list of (query_index, neighbor_id, jaccard), per query sorted by jaccard descending"""
        qnames, lists = read_queries(query_file)
        out = []
        for q0 in range(0, len(lists), self.max_queries):
            out += self._search_lists(lists[q0:q0 + self.max_queries], q0, j, verbose)
        return out

    def _search_lists(self, lists, q_first, j, verbose):
        import torch
        ctx, n, d, names, norms, dev = self.ctx, self.n, self.d, self.names, self.norms, self.dev
        nq = len(lists)
        if nq == 0:
            return []
        offs = np.zeros(nq + 1, dtype=np.int64)
        offs[1:] = np.cumsum([len(x) for x in lists])
        flat = np.concatenate(lists) if offs[-1] else np.zeros(0, dtype=np.uint64)
        q_sk = torch.empty((nq, d), dtype=torch.int32, device=dev)
        q_ss = torch.empty(nq, dtype=torch.int64, device=dev)
        q_max = ctx.project_csr_stats(flat, offs, d, q_sk, q_ss)
        if _capi.limbs_for_max_abs(q_max) > self.limbs:             # a query with larger entries than anything in the DB
            self._load(_capi.limbs_for_max_abs(q_max))
        self.sset.fill(q_sk, n)
        qn2 = q_ss.cpu().numpy().astype(np.float64) / d             # query_norm^2 (:120-121, exact here)
        n2 = torch.from_numpy(np.concatenate([norms * norms, qn2, np.zeros(self.max_queries - nq)])).to(dev)
        # hits land in a grow-only buffer; if it is too small the library says how many there are and the block is
        # compared ONCE more with exactly that room (the reference re-queries FAISS with 3x the neighbours, :131-170)
        cap = max(1 << 16, 256 * nq) if self._hits is None else self._hits.shape[0]
        for attempt in range(2):
            if self._hits is None or self._hits.shape[0] < cap:
                self._hits = torch.empty((cap, 4), dtype=torch.int32, device=dev)
            try:
                cnt = ctx.search_block(self.sset, n2, j, n, n + nq, 0, n, self._hits)
                break
            except _capi.MvsError as e:
                if e.code != _capi.MVS_E_CAPACITY or attempt or not e.needed:
                    raise
                cap = int(e.needed)
        ctx.synchronize()
        hits = self._hits[:cnt].cpu().numpy()
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
                print("Query %d:" % (q_first + qi))
            for rank, k in enumerate(order):
                if not jac[k] > j:
                    continue
                nid = names[mine[k, 1]]
                if verbose:
                    ip = inter[k] / (np.sqrt(qn2[qi]) * norms[mine[k, 1]])
                    print("  Neighbor %d: %s (jaccard: %.4f), inner_product: %.4f %s %s"
                          % (rank, nid, jac[k], ip, norms[mine[k, 1]], np.sqrt(qn2[qi])))
                out.append((q_first + qi, nid, float(jac[k])))
        return out


def search_index(index_folder, query_file, j, ctx=None, verbose=True):
    """-> list of (query_index, neighbor_id, jaccard), per query sorted by jaccard descending
    (what src/jaccard.py:63-224 returns).  Loads the DB for this one call, as the reference loads its faiss.index;
    keep a SearchIndex to query a resident DB repeatedly."""
    nq = sum(1 for line in open(query_file) if line.strip())
    with SearchIndex(index_folder, ctx=ctx, max_queries=max(1, min(nq, 4096))) as idx:
        if ctx is not None:
            idx._own = False
        return idx.search(query_file, j, verbose=verbose)


def index_vectors(output_dir, verbose=True):
    """The reference's `index` step (src/jaccard.py:18-61) reads vectors.bin, L2-normalises a float copy and writes
    faiss.index.  Nothing of that is needed here -- the search runs on the integer sketches -- so this only checks that
    the folder is a usable DB and prints the reference's closing line.  Unlike the reference it deletes nothing (the
    reference removes every file but vectors.bin / vector_norms.txt / dimension.txt, dtype.txt included, :24-30)."""
    names, norms, vectors = read_db(output_dir)
    if vectors.shape[0] != len(names):
        raise ValueError("%s: vectors.bin holds %d vectors but vector_norms.txt names %d"
                         % (output_dir, vectors.shape[0], len(names)))
    if verbose:
        print("Indexed %d vectors of dimension %d into %s." % (vectors.shape[0], vectors.shape[1],
                                                               os.path.join(output_dir, "faiss.index")))
        print("(no index file is written: the search works on vectors.bin directly)")
    return vectors.shape


__version__ = "1.1.0"          # the command line mirrors src/jaccard.py 1.1.0 (03/10/2025)
__date__ = "04/10/2026"


def build_parser():
    """src/jaccard.py:334-346, argument for argument"""
    import argparse
    parser = argparse.ArgumentParser(description="Sketch-DB indexer and searcher (GPU brute force; the reference's FAISS front end).")
    subparsers = parser.add_subparsers(dest="command", required=True)
    parser_index = subparsers.add_parser("index", help="Check a vector folder (no index has to be built).")
    parser_index.add_argument("output_index", type=str, help="Path to the index folder [same folder contains the vectors].")
    parser_index.add_argument("-t", "--threads", type=int, default=1, help="Number of threads [1] (accepted, unused)")
    parser_search = subparsers.add_parser("search", help="Search vectors in an index folder.")
    parser_search.add_argument("index_folder", type=str, help="Path to the index folder.")
    parser_search.add_argument("query_file", type=str,
                               help="Path to query file. Formatted as ID: space_separated_hashes, one ID per line per line")
    parser_search.add_argument("-j", type=float, default=0.1, help="Retrieve all datasets with higher Jaccard index")
    parser_search.add_argument("-t", "--threads", type=int, default=1, help="Number of threads [1] (accepted, unused)")
    parser.add_argument("-v", "--version", action="store_true", help="Show version and date")
    return parser


def main(argv=None):
    import sys
    argv = sys.argv[1:] if argv is None else list(argv)
    args = build_parser().parse_args(argv)
    if args.version:
        print("Version: %s, Date: %s" % (__version__, __date__))
        return 0
    print("Version: %s, Date: %s" % (__version__, __date__))
    print("Command line:", " ".join([sys.argv[0]] + argv))
    if args.command == "index":
        index_vectors(args.output_index)
    elif args.command == "search":
        folder = args.index_folder if args.index_folder.endswith("/") else args.index_folder + "/"
        try:
            search_index(folder, args.query_file, args.j)
        except ValueError as e:
            if str(e).startswith("ERROR 332"):                     # the reference prints the line and exits 332 (:82-84)
                print(e)
                return 332
            raise
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
