"""Synthetic FracMinHash-like inputs: unique-ish u64 hashes drawn uniformly from
[0, 2^64/1000) -- the toy set's max_hash -- with cluster structure so that the keep test fires.

Samples come in clusters of `cluster` members; each sample = `shared` fraction of hashes from its
cluster's pool + the rest private, so cluster mates have Jaccard ~ shared/(2-shared) (0.25 for 0.4).
"""
import numpy as np

MAX_HASH = 18446744073709552  # floor(2^64 / 1000), sourmash scaled=1000


def sample_sizes(n_samples, n_hashes, rng, lognormal_sigma=None):
    if lognormal_sigma is None:
        return np.full(n_samples, int(n_hashes), dtype=np.int64)
    s = rng.lognormal(np.log(n_hashes), lognormal_sigma, n_samples)
    return np.clip(s, 100, 2_000_000).astype(np.int64)


def make_csr_numpy(n_samples, n_hashes, seed, cluster=16, shared=0.4, lognormal_sigma=None):
    """-> (hashes uint64[sum n_i], offsets int64[n_samples+1]) on the host."""
    rng = np.random.default_rng(seed)
    sizes = sample_sizes(n_samples, n_hashes, rng, lognormal_sigma)
    offsets = np.zeros(n_samples + 1, dtype=np.int64)
    offsets[1:] = np.cumsum(sizes)
    hashes = np.empty(offsets[-1], dtype=np.uint64)
    pool = None
    for s in range(n_samples):
        if s % cluster == 0:
            pool = rng.integers(0, MAX_HASH, size=int(sizes[s:s + cluster].max()), dtype=np.uint64)
        n = int(sizes[s])
        k = int(round(shared * n))
        seg = hashes[offsets[s]:offsets[s + 1]]
        seg[:k] = pool[:k]
        seg[k:] = rng.integers(0, MAX_HASH, size=n - k, dtype=np.uint64)
    return hashes, offsets


def make_csr_torch(n_samples, n_hashes, seed, device, cluster=16, shared=0.4):
    """Fixed-size samples generated on the device: -> (hashes int64 tensor [n_samples*n_hashes] holding the
    u64 bit patterns, offsets numpy int64)."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = int(round(shared * n_hashes))
    n_clusters = (n_samples + cluster - 1) // cluster
    pools = torch.randint(0, MAX_HASH, (n_clusters, k), dtype=torch.int64, device=device, generator=g)
    out = torch.empty((n_samples, n_hashes), dtype=torch.int64, device=device)
    out[:, :k] = pools.repeat_interleave(cluster, dim=0)[:n_samples]
    # private part in slabs to bound the temporary
    slab = max(1, (1 << 28) // max(1, n_hashes - k))
    for s0 in range(0, n_samples, slab):
        s1 = min(n_samples, s0 + slab)
        out[s0:s1, k:] = torch.randint(0, MAX_HASH, (s1 - s0, n_hashes - k), dtype=torch.int64, device=device,
                                       generator=g)
    offsets = np.arange(n_samples + 1, dtype=np.int64) * n_hashes
    return out.reshape(-1), offsets


def make_csr_torch_ragged(n_samples, mean_hashes, sigma, seed, device, cluster=16, shared=0.4, lo=100, hi=2_000_000):
    """Ragged samples generated on the device: sizes ~ lognormal(ln mean_hashes, sigma) clipped to [lo, hi]
    (the spread of real FracMinHash sets); a sample shares the first 40 % of its hashes with its
    cluster's pool.  -> (hashes int64 tensor [sum n_i], offsets numpy int64)."""
    import torch
    rng = np.random.default_rng(seed)
    sizes = np.clip(rng.lognormal(np.log(mean_hashes), sigma, n_samples), lo, hi).astype(np.int64)
    offsets = np.zeros(n_samples + 1, dtype=np.int64)
    np.cumsum(sizes, out=offsets[1:])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.randint(0, MAX_HASH, (int(offsets[-1]),), dtype=torch.int64, device=device, generator=g)
    ks = (shared * sizes).astype(np.int64)
    for c0 in range(0, n_samples, cluster):
        c1 = min(n_samples, c0 + cluster)
        kmax = int(ks[c0:c1].max())
        if kmax == 0:
            continue
        pool = torch.randint(0, MAX_HASH, (kmax,), dtype=torch.int64, device=device, generator=g)
        for s in range(c0, c1):
            out[int(offsets[s]):int(offsets[s]) + int(ks[s])] = pool[:int(ks[s])]
    return out, offsets


def make_sketches_numpy(n_samples, d, n_hashes, seed, cluster=16, shared=0.4):
    """Synthesise sketches directly (pairwise-only configs): v = shared component + private component,
    each n - 2*Binomial(n, 1/2) per entry.  int32 [n_samples, d]."""
    rng = np.random.default_rng(seed)
    k = int(round(shared * n_hashes))
    out = np.empty((n_samples, d), dtype=np.int32)
    base = None
    for s in range(n_samples):
        if s % cluster == 0:
            base = k - 2 * rng.binomial(k, 0.5, size=d)
        out[s] = base + (n_hashes - k) - 2 * rng.binomial(n_hashes - k, 0.5, size=d)
    return out


def make_sketches_torch(n_samples, d, n_hashes, seed, device, cluster=16, shared=0.4):
    """Device-side synthesis with a normal approximation of the binomials (rounded to the parity the
    exact sketch would have): int32 [n_samples, d]."""
    import torch
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    k = int(round(shared * n_hashes))
    n_clusters = (n_samples + cluster - 1) // cluster

    def pm_binomial(shape, n):
        # n - 2*Binomial(n, 1/2) ~ N(0, n), same parity as n
        x = torch.randn(shape, device=device, generator=g) * (n ** 0.5)
        r = torch.round((x - (n & 1)) / 2) * 2 + (n & 1)
        return r.to(torch.int32)

    base = pm_binomial((n_clusters, d), k).repeat_interleave(cluster, dim=0)[:n_samples]
    out = base + pm_binomial((n_samples, d), n_hashes - k)
    return out.contiguous()


def make_sketches_torch_rows(n_total, d, n_hashes, seed, device, row_begin, row_end, cluster=16, shared=0.4, slab=65536):
    """Rows [row_begin, row_end) of an n_total-row synthetic sketch matrix that is the same whichever rank (and
    whatever row range) asks: the matrix is defined slab by slab (slab = a multiple of the cluster size), every slab
    from its own generator seeded with (seed, slab index), so a rank generates only the slabs its rows touch.
    Same distribution as make_sketches_torch.  int32 [row_end - row_begin, d]."""
    import torch
    assert slab % cluster == 0
    k = int(round(shared * n_hashes))
    out = torch.empty((max(0, row_end - row_begin), d), dtype=torch.int32, device=device)

    def pm_binomial(shape, n, g):
        x = torch.randn(shape, device=device, generator=g) * (n ** 0.5)
        return (torch.round((x - (n & 1)) / 2) * 2 + (n & 1)).to(torch.int32)

    for s0 in range(row_begin // slab * slab, row_end, slab):
        s1 = min(n_total, s0 + slab)
        g = torch.Generator(device=device)
        g.manual_seed(seed * 1_000_003 + s0 // slab)
        n_cl = (s1 - s0 + cluster - 1) // cluster
        base = pm_binomial((n_cl, d), k, g).repeat_interleave(cluster, dim=0)[:s1 - s0]
        rows = base + pm_binomial((s1 - s0, d), n_hashes - k, g)
        lo, hi = max(s0, row_begin), min(s1, row_end)
        if hi > lo:
            out[lo - row_begin:hi - row_begin] = rows[lo - s0:hi - s0]
    return out
