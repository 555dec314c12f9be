// mvs_project.hip -- projection kernels (gfx950 / CDNA4).
//
// Reference semantics: transform_set_into_vector(), src/random_projection.cpp:9-26
//   v[k] = sum_h (1 - 2*bit_{k%64}(SM(h + 64*(k/64))))  ==  n - 2*count_k
// where count_k = number of hashes whose bit k%64 of SM(h + 64*(k/64)) is set.
//
// MI355X design (see DESIGN.md "K1"):
//   * the +-1 matrix is implicit (hash generated), so there is nothing to stage from HBM; the kernel
//     is integer-VALU bound (19 VALU per splitmix64, 6 of them 32-bit multiplies);
//   * each LANE hashes its own stream of hashes (coalesced 8-byte loads, 512 B per wave load) and
//     counts set bits per position with BIT-SLICED counters: a Harley-Seal carry-save tree built from
//     v_bitop3_b32 full adders (xor3 / majority), ~4.6 VALU per (hash, 64-dim block) instead of 128
//     extract+add;
//   * one wave owns BPW 64-dim blocks of one (sample, hash-chunk) unit; at the end the 64 lanes'
//     bit-sliced counters are summed by a butterfly of bit-sliced ripple adders (ds_bpermute) and
//     lane k extracts count_k;
//   * samples are cut into units of <= 65536 hashes so that long samples spread over many
//     workgroups; units of one sample combine with int32 atomics (order independent => exact).
#include "mvs_internal.h"

namespace mvs {

namespace {

constexpr int kLV = 11;            // bit-sliced counter depth per lane: counts up to 2047
constexpr uint64_t kGolden = 0x9e3779b97f4a7c15ULL;

__device__ __forceinline__ uint32_t xor3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0x96);
}
__device__ __forceinline__ uint32_t maj3(uint32_t a, uint32_t b, uint32_t c) {
    return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8);
}

// src/random_projection.cpp:14-17 applied to z = hash + i + 0x9e37... (the adds of :13-14 are folded
// into one 64-bit add by the caller)
__device__ __forceinline__ uint64_t splitmix_tail(uint64_t z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

template <int BPW>
struct Acc {
    uint32_t lo[BPW][kLV];
    uint32_t hi[BPW][kLV];
};

// Harley-Seal: absorb 2^LEVEL inputs into the persistent bit-sliced digits s[0..LEVEL-1]; returns
// (in clo/chi) the carry word of weight 2^LEVEL.  2^LEVEL - 1 full adders per 2^LEVEL inputs.
template <int LEVEL, int BPW, class Gen>
__device__ __forceinline__ void absorb(Acc<BPW>& s, Gen& g, uint32_t (&clo)[BPW], uint32_t (&chi)[BPW]) {
    if constexpr (LEVEL == 0) {
        g.next(clo, chi);
    } else {
        uint32_t alo[BPW], ahi[BPW], blo[BPW], bhi[BPW];
        absorb<LEVEL - 1, BPW>(s, g, alo, ahi);
        absorb<LEVEL - 1, BPW>(s, g, blo, bhi);
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const uint32_t tl = s.lo[b][LEVEL - 1], th = s.hi[b][LEVEL - 1];
            clo[b] = maj3(tl, alo[b], blo[b]);
            chi[b] = maj3(th, ahi[b], bhi[b]);
            s.lo[b][LEVEL - 1] = xor3(tl, alo[b], blo[b]);
            s.hi[b][LEVEL - 1] = xor3(th, ahi[b], bhi[b]);
        }
    }
}

// add a carry word of weight 2^FROM into digits FROM..kLV-1 (half adders)
template <int FROM, int BPW>
__device__ __forceinline__ void ripple(Acc<BPW>& s, uint32_t (&clo)[BPW], uint32_t (&chi)[BPW]) {
#pragma unroll
    for (int l = FROM; l < kLV; ++l) {
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const uint32_t tl = s.lo[b][l] & clo[b], th = s.hi[b][l] & chi[b];
            s.lo[b][l] ^= clo[b];
            s.hi[b][l] ^= chi[b];
            clo[b] = tl;
            chi[b] = th;
        }
    }
}

// yields SM(h + 64*block) for the next hash of a register-resident batch, for each of the wave's BPW
// blocks.  The batch (8 hashes per lane = 512 per wave) was loaded one batch ahead of its use.
template <int BPW, bool MASKED>
struct BatchGen {
    const uint64_t (&h)[8];
    const uint64_t (&cb)[BPW];   // 64*block + golden
    int64_t remaining;           // MASKED: hashes left from this lane's first hash of the batch
    int j = 0;
    __device__ __forceinline__ BatchGen(const uint64_t (&h_)[8], const uint64_t (&cb_)[BPW], int64_t rem)
        : h(h_), cb(cb_), remaining(rem) {}
    __device__ __forceinline__ void next(uint32_t (&lo)[BPW], uint32_t (&hi)[BPW]) {
        const uint64_t hv = h[j];
        bool valid = true;
        if constexpr (MASKED) valid = (int64_t)j * 64 < remaining;
        ++j;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            uint64_t x = splitmix_tail(hv + cb[b]);
            if constexpr (MASKED) x = valid ? x : 0ULL;
            lo[b] = (uint32_t)x;
            hi[b] = (uint32_t)(x >> 32);
        }
    }
};

// The same values for a wave that owns BPW CONSECUTIVE blocks, with the part of the first splitmix64 round that the
// blocks share computed once per hash.  x_b = x_0 + 64 b differs from x_0 only below bit 30 unless the addition carries
// out of bit 29; without that carry  x_b >> 30 == x_0 >> 30 =: t  and the high word of x_b equals that of x_0, so
//     z_b = x_b ^ (x_b >> 30)  has  z_b.hi = x_0.hi ^ t.hi  (shared)  and  z_b.lo = (x_0.lo + 64 b) ^ t.lo ,
//     z_b * C1 mod 2^64 = z_b.lo * C1.lo  +  ((z_b.lo * C1.hi + z_b.hi * C1.lo) << 32)     with z_b.hi * C1.lo shared.
// Per block that saves the 64-bit add, the 64-bit shift, one xor and one of the four 32-bit multiplies of the round
// (9 quarter-rate instructions per hash at BPW = 4).  The carry case -- bits 8..29 of x_0 all ones, one hash in 4 million
// -- is detected per batch by hazard() and such a batch goes through BatchGen (the wave-uniform branch costs nothing
// when it is not taken), so every value is exact.
constexpr uint32_t kC1Lo = 0x1ce4e5b9u, kC1Hi = 0xbf58476du;

template <int BPW>
struct BatchGenShared {
    const uint64_t (&x)[8];      // hash + 64 * first block + golden (the sums hazard() has looked at)
    int j = 0;
    __device__ __forceinline__ explicit BatchGenShared(const uint64_t (&x_)[8]) : x(x_) {}
    __device__ __forceinline__ void next(uint32_t (&lo)[BPW], uint32_t (&hi)[BPW]) {
        const uint64_t x0 = x[j++];
        const uint64_t t = x0 >> 30;
        const uint32_t x0l = (uint32_t)x0, tl = (uint32_t)t;
        const uint32_t zh = (uint32_t)(x0 >> 32) ^ (uint32_t)(t >> 32);
        const uint32_t p = zh * kC1Lo;
#pragma unroll
        for (int b = 0; b < BPW; ++b) {
            const uint32_t zl = (x0l + 64u * (uint32_t)b) ^ tl;
            const uint64_t m = (uint64_t)zl * (uint64_t)kC1Lo;
            const uint32_t whi = (uint32_t)(m >> 32) + zl * kC1Hi + p;
            uint64_t w = ((uint64_t)whi << 32) | (uint64_t)(uint32_t)m;
            w = (w ^ (w >> 27)) * 0x94d049bb133111ebULL;
            w ^= w >> 31;
            lo[b] = (uint32_t)w;
            hi[b] = (uint32_t)(w >> 32);
        }
    }
};

// x[j] = h[j] + cb0 for the batch; true if some hash of it could carry out of bit 29 when 64 * b (b < 4) is added
__device__ __forceinline__ bool hazard(const uint64_t (&h)[8], uint64_t cb0, uint64_t (&x)[8]) {
    uint32_t least = 0xffffffffu;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        x[j] = h[j] + cb0;
        const uint32_t xl = (uint32_t)x[j];
        const uint32_t miss = ~xl & 0x3fffff00u;                 // zero <=> bits 8..29 are all ones
        least = miss < least ? miss : least;
    }
    return __any(least == 0u) != 0;
}

// load the batch that starts at hash index `pos` of the unit; indices are clamped into the unit so the
// prefetch of a batch that does not exist (or the tail of a partial one) stays in bounds
template <bool CLAMP>
__device__ __forceinline__ void load_batch(uint64_t (&h)[8], const uint64_t* base, int64_t pos, int lane,
                                           int64_t last) {
    if constexpr (CLAMP) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            int64_t idx = pos + j * 64 + lane;
            idx = idx < last ? idx : last;
            h[j] = base[idx];
        }
    } else {
        const uint64_t* p = base + pos + lane;   // whole batch known to be inside the unit
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = p[j * 64];
    }
}

// Sum the 64 lanes' bit-sliced counters and return, in lane k, the count for bit position k.  LV0 = number of
// counter digits that can be non-zero (every lane absorbed fewer than 2^LV0 hashes): short units skip the dead
// digits, which is most of this step's work (6*LV0 + 21 digit additions).
template <int LV0>
__device__ __forceinline__ int32_t reduce_counts(const uint32_t (&lo_in)[kLV], const uint32_t (&hi_in)[kLV],
                                                 int lane) {
    static_assert(LV0 >= 1 && LV0 <= kLV, "live digits");
    constexpr int kOut = LV0 + 6;
    uint32_t lo[kOut], hi[kOut];
#pragma unroll
    for (int l = 0; l < kOut; ++l) {
        lo[l] = l < LV0 ? lo_in[l] : 0u;
        hi[l] = l < LV0 ? hi_in[l] : 0u;
    }
#pragma unroll
    for (int s = 0; s < 6; ++s) {
        uint32_t cl = 0, ch = 0;
#pragma unroll
        for (int l = 0; l < LV0 + s + 1; ++l) {
            const uint32_t pl = (uint32_t)__shfl_xor((int)lo[l], 1 << s, 64);
            const uint32_t ph = (uint32_t)__shfl_xor((int)hi[l], 1 << s, 64);
            const uint32_t sl = xor3(lo[l], pl, cl), sh = xor3(hi[l], ph, ch);
            cl = maj3(lo[l], pl, cl);
            ch = maj3(hi[l], ph, ch);
            lo[l] = sl;
            hi[l] = sh;
        }
    }
    int32_t cnt = 0;
    const int sh = lane & 31;
#pragma unroll
    for (int l = 0; l < kOut; ++l) {
        const uint32_t w = lane < 32 ? lo[l] : hi[l];
        cnt += (int32_t)((w >> sh) & 1u) << l;
    }
    return cnt;
}

// 256 threads = 4 waves, each wave owns BPW consecutive 64-dim blocks and streams over all hashes of
// one unit; a unit needs ny = ceil(nblk / (4*BPW)) workgroups.  1-D grid, XCD aware: workgroups are
// dealt round-robin over the 8 XCDs, so the ny workgroups of one unit are given linear ids 8 apart
// (same XCD, dispatched together) and share the unit's hashes through that XCD's L2:
//   id = (unit/8) * 8*ny + y*8 + unit%8.   Placement only affects HBM traffic, never results.
// STATS: also accumulate each sample's exact sum of squares (int64 atomics, one per wave) and the largest |v|
// of the launch -- valid only when every sample is a single unit (the host checks), because a multi-unit
// sample's entries are only final once all its units have been added.
template <int BPW, bool STATS, bool SHARED = false>
__global__ __launch_bounds__(256) void k_project(const uint64_t* __restrict__ hashes,
                                                 const ProjUnit* __restrict__ units, long long n_units, int ny,
                                                 int d, int nblk, int32_t* __restrict__ out,
                                                 unsigned long long* __restrict__ sumsq,
                                                 unsigned long long* __restrict__ max_abs) {
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long long grp = (long long)blockIdx.x / (8 * ny);
    const int rem = (int)((long long)blockIdx.x % (8 * ny));
    const long long unit = grp * 8 + (rem & 7);
    const int y = rem >> 3;
    if (unit >= n_units) return;
    const int b0 = (y * 4 + wave) * BPW;
    if (b0 >= nblk) return;
    const ProjUnit u = units[unit];
    const uint64_t* base = hashes + u.begin;
    const int64_t count = u.count;

    Acc<BPW> s;
#pragma unroll
    for (int b = 0; b < BPW; ++b)
#pragma unroll
        for (int l = 0; l < kLV; ++l) s.lo[b][l] = s.hi[b][l] = 0u;

    uint64_t cb[BPW];
#pragma unroll
    for (int b = 0; b < BPW; ++b) cb[b] = (uint64_t)(b0 + b) * 64ULL + kGolden;

    // Batches of 8 hashes per lane (512 per wave); batch i+1 is loaded while batch i is hashed.
    if (count > 0) {
        const int64_t last = count - 1;
        const int64_t nfull = count >> 9;
        uint64_t hv[8], hn[8];
        load_batch<true>(hv, base, 0, lane, last);
        int64_t b = 0;
        // four batches = 32 hashes per lane: Harley-Seal tree of depth 5, then one ripple.  The loop
        // runs while the NEXT four batches are full too, so its prefetches need no bounds handling.
        for (; b + 4 < nfull; b += 4) {
            uint32_t c8lo[4][BPW], c8hi[4][BPW];
#pragma unroll
            for (int sb = 0; sb < 4; ++sb) {
                load_batch<false>(hn, base, (b + sb + 1) << 9, lane, last);
                uint64_t xv[8];
                if (SHARED && !hazard(hv, cb[0], xv)) {
                    BatchGenShared<BPW> g(xv);
                    absorb<3, BPW>(s, g, c8lo[sb], c8hi[sb]);
                } else {
                    BatchGen<BPW, false> g(hv, cb, 0);
                    absorb<3, BPW>(s, g, c8lo[sb], c8hi[sb]);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) hv[j] = hn[j];
            }
            uint32_t c16lo[2][BPW], c16hi[2][BPW], c32lo[BPW], c32hi[BPW];
#pragma unroll
            for (int h2 = 0; h2 < 2; ++h2)
#pragma unroll
                for (int q = 0; q < BPW; ++q) {
                    const uint32_t tl = s.lo[q][3], th = s.hi[q][3];
                    c16lo[h2][q] = maj3(tl, c8lo[2 * h2][q], c8lo[2 * h2 + 1][q]);
                    c16hi[h2][q] = maj3(th, c8hi[2 * h2][q], c8hi[2 * h2 + 1][q]);
                    s.lo[q][3] = xor3(tl, c8lo[2 * h2][q], c8lo[2 * h2 + 1][q]);
                    s.hi[q][3] = xor3(th, c8hi[2 * h2][q], c8hi[2 * h2 + 1][q]);
                }
#pragma unroll
            for (int q = 0; q < BPW; ++q) {
                const uint32_t tl = s.lo[q][4], th = s.hi[q][4];
                c32lo[q] = maj3(tl, c16lo[0][q], c16lo[1][q]);
                c32hi[q] = maj3(th, c16hi[0][q], c16hi[1][q]);
                s.lo[q][4] = xor3(tl, c16lo[0][q], c16lo[1][q]);
                s.hi[q][4] = xor3(th, c16hi[0][q], c16hi[1][q]);
            }
            ripple<5, BPW>(s, c32lo, c32hi);
        }
        // leftover full batches
        for (; b < nfull; ++b) {
            load_batch<true>(hn, base, (b + 1) << 9, lane, last);
            BatchGen<BPW, false> g(hv, cb, 0);
            uint32_t clo[BPW], chi[BPW];
            absorb<3, BPW>(s, g, clo, chi);
            ripple<3, BPW>(s, clo, chi);
#pragma unroll
            for (int j = 0; j < 8; ++j) hv[j] = hn[j];
        }
        // partial last batch, masked; only as many of its 8 hash slots are evaluated as hold anything (a sample of
        // 100 hashes needs 2 of them, not 8)
        if ((count & 511) != 0) {
            const int rem = (int)(count & 511);
            BatchGen<BPW, true> g(hv, cb, count - (nfull << 9) - lane);
            uint32_t clo[BPW], chi[BPW];
            if (rem <= 64) {
                absorb<0, BPW>(s, g, clo, chi);
                ripple<0, BPW>(s, clo, chi);
            } else if (rem <= 128) {
                absorb<1, BPW>(s, g, clo, chi);
                ripple<1, BPW>(s, clo, chi);
            } else if (rem <= 256) {
                absorb<2, BPW>(s, g, clo, chi);
                ripple<2, BPW>(s, clo, chi);
            } else {
                absorb<3, BPW>(s, g, clo, chi);
                ripple<3, BPW>(s, clo, chi);
            }
        }
    }

    long long ss = 0;
    unsigned int mx = 0;
#pragma unroll
    for (int b = 0; b < BPW; ++b) {
        if (b0 + b >= nblk) break;
        int32_t cnt;   // a lane absorbed at most ceil(count / 64) hashes
        if (count <= 64) cnt = reduce_counts<1>(s.lo[b], s.hi[b], lane);
        else if (count <= 64 * 15) cnt = reduce_counts<4>(s.lo[b], s.hi[b], lane);
        else if (count <= 64 * 127) cnt = reduce_counts<7>(s.lo[b], s.hi[b], lane);
        else if (count <= 64 * 511) cnt = reduce_counts<9>(s.lo[b], s.hi[b], lane);
        else cnt = reduce_counts<kLV>(s.lo[b], s.hi[b], lane);
        const int k = (b0 + b) * 64 + lane;
        if (k < d) {
            const int32_t v = (int32_t)count - 2 * cnt;
            int32_t* dst = out + (int64_t)u.sample * d + k;
            if (u.single)
                *dst = v;
            else
                atomicAdd(dst, v);
            if (STATS) {
                ss += (long long)v * v;
                const unsigned int av = (unsigned int)(v < 0 ? -v : v);
                mx = av > mx ? av : mx;
            }
        }
    }
    if (STATS) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ss += __shfl_xor(ss, o, 64);
            const unsigned int other = (unsigned int)__shfl_xor((int)mx, o, 64);
            mx = other > mx ? other : mx;
        }
        if (lane == 0) {
            atomicAdd(sumsq + u.sample, (unsigned long long)ss);
            // the running maximum only grows: look before bumping it (one atomic per wave on ONE address
            // serialises at the L2 -- 1.6e7 of them cost 160 ms on a million small samples)
            if (mx > *reinterpret_cast<volatile unsigned long long*>(max_abs)) atomicMax(max_abs, (unsigned long long)mx);
        }
    }
}

// one wave per sketch row: exact int64 sum of squares
__global__ __launch_bounds__(256) void k_sumsq(const int32_t* __restrict__ sk, int64_t n, int d,
                                               int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int32_t* p = sk + row * d;
    long long acc = 0;
    for (int k = lane; k < d; k += 64) {
        const long long v = p[k];
        acc += v * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if (lane == 0) out[row] = acc;
}

// one wave per sketch row: exact int64 sum of squares AND the largest |v| of the whole array (one pass)
__global__ __launch_bounds__(256) void k_stats(const int32_t* __restrict__ sk, int64_t n, int d,
                                               int64_t* __restrict__ sumsq, unsigned long long* __restrict__ max_abs) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n) return;
    const int32_t* p = sk + row * d;
    long long acc = 0;
    unsigned int m = 0;
    if ((d & 3) == 0 && ((uintptr_t)p & 15) == 0) {
        const int4* p4 = reinterpret_cast<const int4*>(p);
        for (int k = lane; k < d / 4; k += 64) {
            const int4 v = p4[k];
            const long long a = v.x, b = v.y, c = v.z, e = v.w;
            acc += a * a + b * b + c * c + e * e;
            const unsigned int ma = (unsigned int)(a < 0 ? -a : a), mb = (unsigned int)(b < 0 ? -b : b);
            const unsigned int mc = (unsigned int)(c < 0 ? -c : c), me = (unsigned int)(e < 0 ? -e : e);
            m = max(max(m, max(ma, mb)), max(mc, me));
        }
    } else {
        for (int k = lane; k < d; k += 64) {
            const long long v = p[k];
            acc += v * v;
            m = max(m, (unsigned int)(v < 0 ? -v : v));
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        acc += __shfl_xor(acc, o, 64);
        m = max(m, (unsigned int)__shfl_xor((int)m, o, 64));
    }
    if (lane == 0) {
        sumsq[row] = acc;
        if (m > *reinterpret_cast<volatile unsigned long long*>(max_abs)) atomicMax(max_abs, (unsigned long long)m);
    }
}

__global__ __launch_bounds__(256) void k_saturate_i16(const int32_t* __restrict__ in, int64_t n,
                                                      int16_t* __restrict__ out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int32_t v = in[i];
        out[i] = (int16_t)(v > 32767 ? 32767 : (v < -32768 ? -32768 : v));
    }
}

}  // namespace

// variant: 1 / 2 = that many 64-dim blocks per wave, every block hashed on its own; 12 / 14 = two / four blocks per wave
// with the shared first splitmix64 round (BatchGenShared)
template <int BPW, bool SHARED>
static void launch_project_as(hipStream_t stream, unsigned grid, bool stats, const uint64_t* d_hashes, const ProjUnit* d_units,
                              long long nu, int ny, int d, int nblk, int32_t* d_out, unsigned long long* d_sumsq,
                              unsigned long long* d_max_abs) {
    if (stats)
        hipLaunchKernelGGL((k_project<BPW, true, SHARED>), dim3(grid), dim3(256), 0, stream, d_hashes, d_units, nu, ny, d, nblk,
                           d_out, d_sumsq, d_max_abs);
    else
        hipLaunchKernelGGL((k_project<BPW, false, SHARED>), dim3(grid), dim3(256), 0, stream, d_hashes, d_units, nu, ny, d, nblk,
                           d_out, d_sumsq, d_max_abs);
}

int launch_project(hipStream_t stream, const uint64_t* d_hashes, const ProjUnit* d_units, int64_t n_units,
                   int d, int32_t* d_out, int variant, unsigned long long* d_sumsq, unsigned long long* d_max_abs) {
    if (n_units == 0) return 0;
    const int nblk = (d + 63) / 64;
    const int bpw = variant % 10;
    const int ny = (nblk + 4 * bpw - 1) / (4 * bpw);
    // a dispatch holds at most 2^32 work-items per dimension: slabs of at most ~2^32/256 workgroups
    const int64_t kMaxUnits = ((int64_t)(0xffffffffLL / 256) / (8 * ny) - 1) * 8;
    for (int64_t u0 = 0; u0 < n_units; u0 += kMaxUnits) {
        const int64_t nu = n_units - u0 < kMaxUnits ? n_units - u0 : kMaxUnits;
        const unsigned grid = (unsigned)(((nu + 7) / 8) * 8 * ny);
        const bool stats = d_sumsq != nullptr;
        switch (variant) {
            case 1: launch_project_as<1, false>(stream, grid, stats, d_hashes, d_units + u0, (long long)nu, ny, d, nblk, d_out, d_sumsq, d_max_abs); break;
            case 12: launch_project_as<2, true>(stream, grid, stats, d_hashes, d_units + u0, (long long)nu, ny, d, nblk, d_out, d_sumsq, d_max_abs); break;
            case 14: launch_project_as<4, true>(stream, grid, stats, d_hashes, d_units + u0, (long long)nu, ny, d, nblk, d_out, d_sumsq, d_max_abs); break;
            default: launch_project_as<2, false>(stream, grid, stats, d_hashes, d_units + u0, (long long)nu, ny, d, nblk, d_out, d_sumsq, d_max_abs); break;
        }
    }
    return 0;
}

// one wave per row, 4 rows per workgroup; slabs of 2^22 workgroups keep each dispatch below 2^32 work-items
constexpr int64_t kRowSlab = (int64_t)4 << 22;

int launch_sumsq(hipStream_t stream, const int32_t* d_sk, int64_t n, int d, int64_t* d_out) {
    for (int64_t r0 = 0; r0 < n; r0 += kRowSlab) {
        const int64_t m = n - r0 < kRowSlab ? n - r0 : kRowSlab;
        hipLaunchKernelGGL(k_sumsq, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, d_sk + r0 * d, m, d, d_out + r0);
    }
    return 0;
}

int launch_stats(hipStream_t stream, const int32_t* d_sk, int64_t n, int d, int64_t* d_sumsq,
                 unsigned long long* d_max_abs) {
    for (int64_t r0 = 0; r0 < n; r0 += kRowSlab) {
        const int64_t m = n - r0 < kRowSlab ? n - r0 : kRowSlab;
        hipLaunchKernelGGL(k_stats, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, stream, d_sk + r0 * d, m, d, d_sumsq + r0,
                           d_max_abs);
    }
    return 0;
}

int launch_saturate_i16(hipStream_t stream, const int32_t* d_in, int64_t n, int16_t* d_out) {
    if (n == 0) return 0;
    int64_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(k_saturate_i16, dim3((unsigned)blocks), dim3(256), 0, stream, d_in, n, d_out);
    return 0;
}

}  // namespace mvs
