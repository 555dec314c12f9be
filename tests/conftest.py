import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLD = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # the native pieces are git-ignored build products: build them once if this is a fresh checkout
    pkg = os.path.join(ROOT, "metagenome_vector_sketches_amd")
    need = [os.path.join(pkg, "libmvs_hip.so"), os.path.join(pkg, "bin", "query_pc_mat"),
            os.path.join(pkg, "bin", "project_everything"), os.path.join(ROOT, "oracle", "libmvs_oracle.so")]
    if not all(os.path.exists(p) for p in need):
        import __graft_entry__
        __graft_entry__.build()


class Golden:
    """Committed fixtures (tests/golden/, produced by tests/golden/make_golden.py from the reference)."""

    def __init__(self):
        with open(os.path.join(GOLD, "kat.json")) as f:
            self.kat = json.load(f)
        with open(os.path.join(GOLD, "toy_sketch_digests.json")) as f:
            self.digests = json.load(f)
        db = np.load(os.path.join(GOLD, "toy_db.npz"))
        self.names = [str(x) for x in db["names"]]
        self.vectors = db["vectors"]
        with open(os.path.join(GOLD, "toy_vector_norms.txt")) as f:
            self.norms_txt = f.read()
        h = np.load(os.path.join(GOLD, "toy_hashes.npz"))
        assert [str(x) for x in h["names"]] == self.names
        self.offsets = h["offsets"].astype(np.int64)
        deltas = h["deltas"].astype(np.uint64)
        hashes = deltas.copy()
        for i in range(len(self.names)):
            b, e = self.offsets[i], self.offsets[i + 1]
            hashes[b:e] = np.cumsum(deltas[b:e], dtype=np.uint64)
        self.hashes = hashes

    def cells(self, int16=False):
        fn = "toy_pairwise_cells_int16.txt" if int16 else "toy_pairwise_cells.txt"
        idx = {n: i for i, n in enumerate(self.names)}
        out = []
        with open(os.path.join(GOLD, fn)) as f:
            for line in f:
                if line.startswith("#"):
                    continue
                r, c, dot, q = line.split()
                out.append((idx[r], idx[c], int(dot), int(q)))
        return out

    @staticmethod
    def eigen_blocks(case):
        """the blocks oracle/eigen_gemm_check.cpp generated for a kat.json eigen_gemm_cases entry (same splitmix64-based
        formula): -> (block_i int32 [c_i, d], block_j int32 [c_j, d])"""
        def mix(x):
            x = (x + 0x9e3779b97f4a7c15) & (2**64 - 1)
            x = ((x ^ (x >> 30)) * 0xbf58476d1ce4e5b9) & (2**64 - 1)
            x = ((x ^ (x >> 27)) * 0x94d049bb133111eb) & (2**64 - 1)
            return x ^ (x >> 31)
        d, mag, seed = case["d"], case["magnitude"], case["seed"]

        def block(first, count):
            return np.array([[mix(seed * 1000003 + (first + s) * 65537 + k) % (2 * mag + 1) - mag for k in range(d)]
                             for s in range(count)], dtype=np.int64).astype(np.int32)
        return block(0, case["c_i"]), block(1000, case["c_j"])

    def norm_lines(self):
        return [l for l in self.norms_txt.split("\n") if l]

    @staticmethod
    def formula_sketches(n, d, seed, cluster, amp, shared_amp):
        """the sketches of a ref_pairwise.json case given as a formula (tests/golden/make_golden_pairwise.py wrote the same
        rows to the vectors.bin the reference functions read): noise(row, k) + shared(cluster(row), k), both uniform
        integers from a splitmix64 counter"""
        def mix(x):
            x = x + np.uint64(0x9e3779b97f4a7c15)
            x = (x ^ (x >> np.uint64(30))) * np.uint64(0xbf58476d1ce4e5b9)
            x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94d049bb133111eb)
            return x ^ (x >> np.uint64(31))
        with np.errstate(over="ignore"):
            rows = np.arange(n, dtype=np.uint64)[:, None]
            ks = np.arange(d, dtype=np.uint64)[None, :]
            a = mix(np.uint64(seed) * np.uint64(1000003) + rows * np.uint64(65537) + ks)
            b = mix(np.uint64(seed) * np.uint64(7919) + (rows // np.uint64(cluster)) * np.uint64(2654435761) + ks
                    + np.uint64(1 << 40))
        return ((a % np.uint64(2 * amp + 1)).astype(np.int64) - amp
                + (b % np.uint64(2 * shared_amp + 1)).astype(np.int64) - shared_amp)

    def ref_pairwise_cases(self):
        """kept cells of the REFERENCE's own pairwise functions (oracle/_ref/ref_pairwise32 / 16, compiled from line ranges
        of src/pairwise_comp_optimized*.cpp; tests/golden/make_golden_pairwise.py): name -> dict(elem, d, vectors,
        norm_lines, norms_sq, runs=[dict(max_memory_gb, chunk, num_shards, shard_idx, cells int64 [kept, 3] in the
        reference's append order, cells_sha256)], writer_head=dict(cells int64 [kept, 3] = (row, col, q) for int32 DBs /
        (row, col, round(dot / d)) for int16 DBs as the bits-free first lines of the reference's writers compute them
        (:654-672, _16bits.cpp:260-280), in the writer's emission order; row_order; undefined = cells whose Jaccard is NaN
        in the reference (uint16(round(NaN)) is undefined behaviour there)))"""
        if getattr(self, "_ref_pairwise", None) is None:
            with open(os.path.join(GOLD, "ref_pairwise.json")) as f:
                meta = json.load(f)["cases"]
            data = np.load(os.path.join(GOLD, "ref_pairwise_inputs.npz"))
            out = {}
            for name, c in meta.items():
                dt = np.int32 if c["elem"] == 4 else np.int16
                src = c["vectors"]
                if src == "inline":
                    vec = data[name]
                elif src == "toy_db":
                    vec = self.vectors
                elif src == "toy_db_int16":
                    vec = np.clip(self.vectors, -32768, 32767)
                else:
                    vec = self.formula_sketches(**{k: v for k, v in src.items() if k != "formula"})
                n2 = []
                for l in c["norm_lines"]:
                    x = float(l.split(" ", 1)[1])          # :893-901 stod(text after the first ' ')
                    n2.append(x * x)
                runs = [dict(r, cells=data[r["cells"]]) for r in c["runs"]]
                wh = c["writer_head"]
                head = dict(cells=data[wh["cells"]], row_order=wh["row_order"],
                            undefined={tuple(x) for x in wh["undefined_in_reference"]})
                out[name] = dict(elem=c["elem"], d=c["d"], vectors=np.ascontiguousarray(vec, dtype=dt),
                                 norm_lines=c["norm_lines"], norms_sq=np.array(n2, dtype=np.float64), runs=runs,
                                 writer_head=head)
                assert out[name]["vectors"].shape == (c["n"], c["d"])
            self._ref_pairwise = out
        return self._ref_pairwise


@pytest.fixture(scope="session")
def gold():
    return Golden()


@pytest.fixture(scope="session")
def ctx():
    """GPU context through the C ABI; only -m gpu tests use it."""
    from metagenome_vector_sketches_amd import Context
    c = Context(0)
    yield c
    c.close()


@pytest.fixture
def restore_options(ctx):
    """the context is shared by the whole session: whatever a test switches is switched back"""
    names = ("pairwise_filter", "pairwise_block_cells", "filter_variant", "exact_variant", "pairwise_variant",
             "pairwise_symmetric", "sort", "tile_dense_thr", "stream_list_cells", "stream_block_rows", "stream_dense", "stream_pipeline", "search_stream", "fragment_major", "pairwise_bdirect")
    old = {k: ctx.get_option(k) for k in names}
    yield
    for k, v in old.items():
        ctx.set_option(k, v)
