"""GPU: the HIP projection kernel (through the C ABI) against the oracle, the reference fixtures and
size-independent properties.  Bit-exact: the sketch is integer work."""
import hashlib

import numpy as np
import pytest

from metagenome_vector_sketches_amd import synth
from oracle import pyoracle as orc

pytestmark = pytest.mark.gpu


def _csr(lists):
    offs = np.zeros(len(lists) + 1, dtype=np.int64)
    offs[1:] = np.cumsum([len(x) for x in lists])
    flat = np.concatenate([np.asarray(x, dtype=np.uint64) for x in lists]) if offs[-1] else np.zeros(0, np.uint64)
    return flat, offs


def test_kat(ctx, gold):
    # hashes {1,2,3}, d=8 -> -1 1 -1 -1 3 1 -3 -3
    h, o = _csr([[1, 2, 3]])
    assert ctx.project_csr(h, o, 8).tolist() == [[-1, 1, -1, -1, 3, 1, -3, -3]]


def test_standalone_projection_fixtures(ctx, gold):
    for key, case in gold.kat["standalone_projection"].items():
        d = case["d"]
        lines = case["input"].split("\n")[:-1]
        want = [[float(t) for t in l.split(" ")] for l in case["stdout"].split("\n")[:-1]]
        lists = [sorted(set(int(t) for t in line.split())) for line in lines]
        h, o = _csr(lists)
        got = ctx.project_csr(h, o, d)
        assert got.shape == (len(lines), d)
        assert np.array_equal(got.astype(np.float64), np.array(want)), key


def test_toy_set_bit_exact(ctx, gold):
    """config 1 of BASELINE.json: the 61 toy samples (3 ... 80772 hashes), d=2048, against the
    reference's own vectors.bin"""
    got = ctx.project_csr(gold.hashes, gold.offsets, 2048)
    assert np.array_equal(got, gold.vectors)
    assert hashlib.sha256(got.tobytes()).hexdigest() == gold.kat["toy_vectors_sha256_sorted_by_name"]


@pytest.mark.parametrize("d", [64, 100, 2048, 4096, 2048 + 64])
def test_ragged_vs_oracle(ctx, d):
    rng = np.random.default_rng(d)
    sizes = [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4096 + 300, 0, 7000, 65535, 65536, 65537,
             131072 + 5]
    lists = [rng.integers(0, synth.MAX_HASH, size=s, dtype=np.uint64) for s in sizes]
    h, o = _csr(lists)
    got = ctx.project_csr(h, o, d)
    want = orc.project_csr(h, o, d, threads=8, fast=True)
    assert np.array_equal(got, want)


def test_u64_wraparound(ctx):
    # hash + 64*block wraps mod 2^64 (src/random_projection.cpp:13)
    lists = [[2**64 - 1, 2**64 - 64, 2**64 - 2048, 5], [2**63, 2**63 - 1]]
    h, o = _csr(lists)
    assert np.array_equal(ctx.project_csr(h, o, 4096), orc.project_csr(h, o, 4096))


def test_device_resident_io(ctx):
    import torch
    hashes, offsets = synth.make_csr_torch(96, 5000, seed=7, device="cuda")
    out = torch.empty((96, 2048), dtype=torch.int32, device="cuda")
    ctx.set_stream(torch.cuda.current_stream())
    ctx.project_csr(hashes, offsets, 2048, out=out)
    torch.cuda.synchronize()
    ctx.set_stream(None)
    want = orc.project_csr(hashes.cpu().numpy().view(np.uint64), offsets, 2048, threads=8, fast=True)
    assert np.array_equal(out.cpu().numpy(), want)


def test_full_size_properties(ctx):
    """BASELINE config-2 sample size (50k hashes): properties that need no oracle.
    - parity: v[k] == n (mod 2) and |v[k]| <= n
    - additivity: sketch(A u B) == sketch(A) + sketch(B) for disjoint A, B
    - order independence"""
    import torch
    n, d = 50_000, 2048
    hashes, offsets = synth.make_csr_torch(64, n, seed=11, device="cuda", cluster=1, shared=0.0)
    out = torch.empty((64, d), dtype=torch.int32, device="cuda")
    ctx.project_csr(hashes, offsets, d, out=out)
    ctx.synchronize()
    v = out.cpu().numpy()
    assert np.all((v - n) % 2 == 0) and np.all(np.abs(v) <= n)
    assert np.abs(v).max() < 8 * np.sqrt(n)            # +-1 sums: 8 sigma
    # pairs of consecutive samples as one 100k-hash sample (exercises the multi-unit atomic path)
    out2 = torch.empty((32, d), dtype=torch.int32, device="cuda")
    ctx.project_csr(hashes, offsets[::2].copy(), d, out=out2)
    ctx.synchronize()
    assert np.array_equal(out2.cpu().numpy(), v[0::2] + v[1::2])
    # permute the hashes inside each sample
    perm = torch.randperm(n, device="cuda")
    shuffled = hashes.view(64, n)[:, perm].contiguous().view(-1)
    out3 = torch.empty_like(out)
    ctx.project_csr(shuffled, offsets, d, out=out3)
    ctx.synchronize()
    assert torch.equal(out3, out)
    # spot-check two rows against the oracle
    hh = hashes.cpu().numpy().view(np.uint64)
    for s in (0, 63):
        assert np.array_equal(v[s], orc.project(hh[offsets[s]:offsets[s + 1]], d))


def test_sumsq_and_saturate(ctx, gold):
    got = ctx.sumsq(gold.vectors)
    assert got.tolist() == [gold.digests[n]["sumsq"] for n in gold.names]
    v = np.array([[0, 1, -1, 32767, 32768, -32768, -32769, 2**31 - 1, -2**31, 5]], dtype=np.int32)
    assert np.array_equal(ctx.saturate_i16(v), orc.saturate_i16(v))


def test_bad_arguments(ctx):
    from metagenome_vector_sketches_amd import MvsError
    h, o = _csr([[1, 2, 3]])
    with pytest.raises(MvsError):
        ctx.project_csr(h, o, 0)
    with pytest.raises(MvsError):
        ctx.project_csr(h, np.array([0, 3, 2], dtype=np.int64), 64)


def test_stats_one_pass(ctx, gold):
    ss, m = ctx.stats(gold.vectors)
    assert ss.tolist() == [gold.digests[n]["sumsq"] for n in gold.names]
    assert m == int(np.abs(gold.vectors).max())
    v = np.array([[-2**31, 5, 0], [7, -9, 1]], dtype=np.int32)      # d % 4 != 0 path, INT32_MIN
    ss, m = ctx.stats(v)
    assert ss.tolist() == [2**62 + 25, 131] and m == 2**31


def test_project_with_fused_stats(ctx):
    """mvs_project_csr_stats: statistics fused into the projection kernel (single-unit samples) and the
    fallback pass (a sample of > 65536 hashes spans several units)"""
    import torch
    rng = np.random.default_rng(5)
    for sizes in ([0, 1, 300, 4096, 65536, 20000], [10, 70000, 3, 131073]):
        lists = [rng.integers(0, synth.MAX_HASH, size=s, dtype=np.uint64) for s in sizes]
        h, o = _csr(lists)
        want = orc.project_csr(h, o, 2048, threads=8, fast=True)
        out = torch.empty((len(sizes), 2048), dtype=torch.int32, device="cuda")
        ss = torch.full((len(sizes),), -1, dtype=torch.int64, device="cuda")
        m = ctx.project_csr_stats(torch.from_numpy(h.view(np.int64)).to("cuda"), o, 2048, out, ss)
        ctx.synchronize()
        assert np.array_equal(out.cpu().numpy(), want)
        assert ss.cpu().tolist() == (want.astype(np.int64) ** 2).sum(1).tolist()
        assert m == int(np.abs(want).max())


def test_toy_set_other_dimensions_reference_digests(ctx, gold):
    """the reference binary's own output at d = 4096 (BASELINE config 4) and d = 100 (d % 64 != 0)"""
    for d in (4096, 100):
        got = ctx.project_csr(gold.hashes, gold.offsets, d)
        assert hashlib.sha256(got.tobytes()).hexdigest() == gold.kat["toy_vectors_sha256_d%d" % d]
        for i, n in enumerate(gold.names):
            assert hashlib.sha256(got[i].tobytes()).hexdigest() == gold.kat["toy_row_sha256_d%d" % d][n]
        ref_norms = [float(l.split(" ")[1]) for l in gold.kat["toy_norms_d%d" % d].strip().split("\n")]
        mine = np.sqrt(ctx.sumsq(got).astype(np.float64) / d)
        assert np.allclose(mine, ref_norms, rtol=1e-5, atol=1e-12)


def test_many_tiny_samples_cross_the_dispatch_limit(ctx):
    """17.6 M samples of 2 hashes, d = 64: more workgroups than one dispatch can hold (2^32 work-items), so
    the launcher must cut the unit list into slabs.  Expected sketches from a torch restatement of splitmix64
    (int64 arithmetic wraps like uint64; logical shifts emulated)."""
    import torch
    n = 17_600_000
    g = torch.Generator(device="cuda")
    g.manual_seed(9)
    h = torch.randint(0, synth.MAX_HASH, (n, 2), dtype=torch.int64, device="cuda", generator=g)
    offsets = np.arange(n + 1, dtype=np.int64) * 2
    out = torch.empty((n, 64), dtype=torch.int32, device="cuda")
    ctx.project_csr(h.view(-1), offsets, 64, out=out)
    ctx.synchronize()

    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)

    def c(v):   # uint64 constant as a wrapped int64
        return v - (1 << 64) if v >= (1 << 63) else v

    z = h + c(0x9e3779b97f4a7c15)
    z = (z ^ lsr(z, 30)) * c(0xbf58476d1ce4e5b9)
    z = (z ^ lsr(z, 27)) * c(0x94d049bb133111eb)
    z = z ^ lsr(z, 31)
    for k in (0, 1, 31, 32, 62, 63):
        bits = (lsr(z, k) & 1) if k else (z & 1)
        want = (2 - 2 * bits.sum(dim=1)).to(torch.int32)
        assert torch.equal(out[:, k], want), k


def test_host_input_goes_through_the_upload_pipeline(ctx):
    """host hash lists beyond one 32 MiB staging piece are uploaded piecewise through two pinned buffers while the
    samples already complete are projected (several launches): same sketches as from device-resident input, also with
    a sample that spans pieces and is cut into several units"""
    import torch
    sizes = np.full(600, 60_000, dtype=np.int64)
    sizes[7] = 0
    sizes[300] = 2_500_000                       # spans the first piece border region and several 65 536-hash units
    offsets = np.zeros(len(sizes) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(sizes)
    rng = np.random.default_rng(77)
    hashes = rng.integers(0, synth.MAX_HASH, size=int(offsets[-1]), dtype=np.uint64)
    assert hashes.nbytes > 2 * (128 << 20)
    got = ctx.project_csr(hashes, offsets, 1024)                      # host in, host out
    dev = torch.from_numpy(hashes.view(np.int64)).to("cuda")
    want = ctx.project_csr(dev, offsets, 1024)                        # device in, host out (single launch)
    assert np.array_equal(got, want)
    for s in (0, 7, 300, 599):
        assert np.array_equal(got[s], orc.project(hashes[offsets[s]:offsets[s + 1]], 1024))


@pytest.mark.parametrize("d", [2048, 512, 300])
def test_shared_round_variants_and_their_carry_hazard(ctx, d):
    """project_variant 14 / 12 compute the part of the first splitmix64 round that consecutive blocks share once per
    hash; that is only valid while adding 64*b does not carry out of bit 29 of h + golden + 64*b0, so batches holding
    such a hash (bits 8..29 all ones: one in 4 million) take the general path.  Hashes built to sit on that edge for
    every block group, mixed into ordinary ones: every variant must give the oracle's sketch."""
    rng = np.random.default_rng(5)
    golden, nblk = 0x9e3779b97f4a7c15, (d + 63) // 64
    edge = []
    for b0 in range(0, nblk):
        for low8 in (0, 63, 64, 65, 128, 200, 255):
            target = 0x3fffff00 | low8 | (int(rng.integers(0, 4)) << 30)
            lo = (target - ((golden + 64 * b0) & 0xffffffff)) & 0xffffffff
            edge.append((int(rng.integers(0, 2**31)) << 32) | lo)
    sizes = [len(edge) + 3000, 700, 0, 66000, 5]
    offsets = np.zeros(len(sizes) + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(sizes)
    hashes = rng.integers(0, 2**63, size=int(offsets[-1]), dtype=np.uint64)
    pos = rng.permutation(sizes[0])[:len(edge)]
    hashes[pos] = np.array(edge, dtype=np.uint64)
    hashes[offsets[3] + 1000:offsets[3] + 1000 + len(edge)] = np.array(edge, dtype=np.uint64)
    want = orc.project_csr(hashes, offsets, d, threads=8, fast=True)
    old = ctx.get_option("project_variant")
    try:
        for v in (14, 12, 2, 1, 0):
            ctx.set_option("project_variant", v)
            assert np.array_equal(ctx.project_csr(hashes, offsets, d), want), v
    finally:
        ctx.set_option("project_variant", old)


def test_norm_text_round_trip_on_the_device(ctx):
    """mvs_norms_sq_text == (strtod of the "%g" line of vector_norms.txt)^2, bit for bit: the oracle's
    format_norm / norm_sq_from_text (project_everything.cpp:328-330, pairwise_comp_optimized.cpp:893-901) on
    random sums of squares, on exact decimal ties, next to powers of ten and at the extremes."""
    import torch
    rng = np.random.default_rng(99)
    cases = []
    for d in (2048, 4096, 64, 1, 3, 1000):
        vals = [0, 1, 2, 3, d - 1, d, d + 1, 2**62, 2**63 - 1]
        vals += [int(v) for v in rng.integers(1, 2**40, size=3000)]
        vals += [int(v) for v in rng.integers(1, 2**62, size=2000)]
        vals += [int(v) for v in (rng.integers(1, 3000, size=1000).astype(np.int64) ** 2 * 50_000)]   # sketch-like sizes
        # sqrt(s / d) a short dyadic decimal: exact ties at the 7th digit (k / 16, k / 32, ... with 7 digits)
        for k in range(1, 400000, 997):
            for den in (2, 4, 8, 16, 32, 64):
                num = k * k * d
                if num % (den * den) == 0:
                    vals.append(num // (den * den))
        # next to powers of ten: sqrt(s / d) ~ 10^e
        for e in range(-1, 9):
            c = int(round(d * 10.0 ** (2 * e)))
            vals += [max(0, c + o) for o in (-2, -1, 0, 1, 2)]
        # 6-digit boundaries: x ~ 999999.5 * 10^q
        for q in range(-5, 4):
            x = 999999.5 * 10.0 ** q
            c = int(x * x * d)
            vals += [max(0, c + o) for o in (-1, 0, 1)]
        cases.append((d, np.array([v for v in vals if 0 <= v < 2**63], dtype=np.int64)))
    for d, s in cases:
        dev = torch.from_numpy(s).cuda()
        out = torch.empty(len(s), dtype=torch.float64, device="cuda")
        torch.cuda.synchronize()              # the upload ran on torch's stream, the kernel runs on the context's
        ctx.norms_sq_text(dev, d, out)
        ctx.synchronize()                     # asynchronous entry point
        got = out.cpu().numpy()
        want = np.array([orc.norm_sq_from_text(orc.format_norm(np.sqrt(float(v) / float(d)))) for v in s.tolist()])
        bad = np.nonzero(got != want)[0]
        assert len(bad) == 0, (d, s[bad[:5]], got[bad[:5]], want[bad[:5]])
    with pytest.raises(ValueError):
        ctx.norms_sq_text(np.zeros(4, dtype=np.int64), 2048, np.zeros(4))
