// mvs_recode.hip -- sketches -> what the comparison reads: largest |v|, signed base-256 limb planes, the filter's coarse plane
// (row-major and fragment-major) with its row statistics, all of them in one pass (k_recode_rows), the limb planes rebuilt from
// the low-limb wire format, per-row filter constants and thresholds.  load_matrix_block's re-coding
// (src/pairwise_comp_optimized.cpp:33-54) for the MI355X comparison kernels (mvs_pairwise.hip).
#include "mvs_internal.h"
#include "mvs_encode.h"
#include "mvs_pairwise_dev.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

namespace mvs {

namespace {

// ---------------------------------------------------------------------------------------------------
// helpers: max |v|, limb split, candidate thresholds
// ---------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void k_max_abs(const T* __restrict__ v, int64_t n,
                                                 unsigned long long* __restrict__ out) {
    unsigned long long m = 0;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const long long x = (long long)v[i];
        const unsigned long long ax = (unsigned long long)(x < 0 ? -x : x);
        m = ax > m ? ax : m;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned long long other = __shfl_xor(m, o, 64);
        m = other > m ? other : m;
    }
    if ((threadIdx.x & 63) == 0 && m) atomicMax(out, m);
}

// signed base-256 digits: v = l0 + 256*l1 + ... (mod 2^32), every digit in [-128, 127]
template <typename T>
__global__ __launch_bounds__(256) void k_limb_split(const T* __restrict__ sk, int64_t n_rows, int d, int limbs,
                                                    int8_t* __restrict__ planes, int d_pad, int64_t row_offset) {
    const int words = d_pad / 4;
    const int64_t total = n_rows * words;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t row = idx / words;
        const int k = (int)(idx % words) * 4;
        if (k >= d) continue;   // pad words stay zero
        int32_t v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = (k + e < d) ? (int32_t)sk[row * d + k + e] : 0;
        const int np = planes_of(limbs);
        int8_t* dst = planes + (row_offset + row) * np * (int64_t)d_pad + k;
        if (is_k3(limbs)) {
            // signed base-128 digits l0, l1 in [-64, 63] (|v| <= 8127) and their sum, which fits int8
            uint32_t p0 = 0, p1 = 0, p2 = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int32_t l0 = ((v[e] + 64) & 127) - 64;
                const int32_t l1 = (v[e] - l0) >> 7;
                p0 |= (uint32_t)(uint8_t)(int8_t)l0 << (8 * e);
                p1 |= (uint32_t)(uint8_t)(int8_t)l1 << (8 * e);
                p2 |= (uint32_t)(uint8_t)(int8_t)(l0 + l1) << (8 * e);
            }
            *reinterpret_cast<uint32_t*>(dst) = p0;
            *reinterpret_cast<uint32_t*>(dst + (int64_t)d_pad) = p1;
            *reinterpret_cast<uint32_t*>(dst + 2 * (int64_t)d_pad) = p2;
            continue;
        }
        for (int l = 0; l < limbs; ++l) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int32_t digit = (int32_t)(int8_t)(v[e] & 0xff);
                packed |= (uint32_t)(uint8_t)digit << (8 * e);
                // v - digit is a multiple of 256; unsigned subtract so that the one wrapping case
                // (v near INT32_MAX, 4 limbs) stays defined and congruent mod 2^32
                v[e] = (int32_t)((uint32_t)v[e] - (uint32_t)digit) >> 8;
            }
            *reinterpret_cast<uint32_t*>(dst + (int64_t)l * d_pad) = packed;
        }
    }
}

// conservative integer part of the keep threshold: keep(i,j) implies P >= thr[i] + thr[j]
__global__ __launch_bounds__(256) void k_cand_thr(const double* __restrict__ n2, int64_t n, int64_t n_alloc,
                                                  int d, double coeff, int32_t* __restrict__ thr) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alloc) return;
    int32_t t = (1 << 30) - 1;   // padding rows: never a candidate
    if (i < n) {
        const double x = n2[i];
        t = -1;
        if (x >= 0.0) {
            const double f = floor(coeff * (double)d * x * (1.0 - 1.0 / 1048576.0)) - 1.0;
            t = f >= 1073741823.0 ? (1 << 30) - 1 : (f < -1.0 ? -1 : (int32_t)f);
        }
    }
    thr[i] = t;
}

// ---------------------------------------------------------------------------------------------------
// Two-stage comparison ("filter"), for sets of two base-256 limbs.
//
// Every row also gets ONE int8 plane c = round(v / m) with its own radix m = ceil(max|v| / 127), and
// r = v - m c is only known through its norm.  With A = m_i m_j <c_i,c_j>, Cauchy-Schwarz gives
//     | <v_i,v_j> - A |  <=  m_i |c_i| |r_j| + |r_i| m_j |c_j| + |r_i| |r_j|  =: B .
// Both keep tests imply  P > d * coeff * (n2_i + n2_j) =: tau_i + tau_j  where P is the int32 dot.  P is the
// true dot unless it wraps, and it can only wrap if |v_i| |v_j| >= 2^31, i.e. if one of the two rows has a sum
// of squares >= 2^31: such "big" rows get s = -inf and pair with everything as candidates (the re-check
// reproduces the wrapped value exactly).  For all other pairs a kept pair satisfies  A + B > tau_i + tau_j,
// i.e. after dividing by m_i m_j, with a = |c|, p = |r| / m, s = tau / m, w = 1 / m:
//     <c_i,c_j>  >  s_i w_j + s_j w_i - a_i p_j - p_i (a_j + p_j) .
// The one-pass MFMA filter evaluates exactly that per cell in fp32; s is deflated and a, p are inflated by
// 2^-12, which dominates every rounding error of the evaluation (4 fused operations, 2^-22 relative to
// the sum of magnitudes) and of the int -> float conversion of the dot, so no kept pair is ever dropped.
// (The ping-pong kernel's epilogue forms  (float)dot - rhs  as ONE fma chain on the matrix pipe, the converted dot
// being the addend: five roundings of 2^-24 relative to |dot| + the sum of magnitudes.  The sign of the result can
// only be in doubt where |dot| is about rhs, i.e. at most that sum, so the error is below 2^-20 of it against a
// margin of 2^-12.)
// Pairs that pass go to a candidate list; k_exact_pairs recomputes their dots exactly from the limb
// planes and applies the reference's keep test and quantisation.  On typical sketches (d = 2048) B is
// about a fifth of the threshold and ~1e-4 of the unrelated pairs pass.
// ---------------------------------------------------------------------------------------------------
// The high limb on the wire (multi-rank steps: mvs_sketch_set_planes_from_wire).  A rank that holds a row's coarse plane c,
// its radix m and its LOW limb l0 can rebuild the high limb: v is the one value congruent to l0 mod 256 near m c --
//   |c| < 127 :  |v - m c| <= ceil(m / 2) <= 126 for m <= 252, so v = t + wrap8(l0 - t) with t = m c;
//   |c| = 127 :  v lies beyond: s v in [L, max|v|] with L = 127 m - ceil(m / 2), s = sign(c) -- one value mod 256 as long as
//                max|v| <= L + 254, and then v = t' + wrap8(l0 - t') with t' = s (L + 127)
// (wrap8 = the representative in [-128, 127]; checked exhaustively for every m <= 252 and every v the rule admits:
// tests/test_oracle_golden.py).  The radix search therefore only tries radices with max|v| <= L + 254 -- the radix that
// just avoids clamping, ceil(max|v| / 127), always qualifies -- and the exchange carries 2 bytes per entry instead of 3.
__device__ __forceinline__ bool radix_keeps_high_limb(int mc, int mx) { return mx <= 127 * mc - (mc + 1) / 2 + 254; }
static_assert(MVS_WIRE_RADIX_MAX == 252 && MVS_WIRE_MAX_ABS == 127 * 252, "the bounds the header states");

// One entry of a radix trial: the squared residual of v under radix mc (ic = 1.0f / mc).  Two-limb values only:
// |v| <= 32896 = 128 * 256 + 128, so the radix that just avoids clamping is m <= 260, a trial radix is mc >= m - 15 * step
// with step <= 8, and |r| is at most mc / 2 where the coarse value is not clamped and |v| - 127 mc <= 127 (m - mc) <= 15240
// where it is: every factor fits the 24-bit multipliers (full rate; the 32-bit multiply and the 64-bit
// multiply-add are quarter rate, and sixteen trials over every entry are what the kernels around this spend their time on),
// and sixteen squares fit 32 bits (16 * 15240^2 = 3.72e9).
constexpr int kTrialResidualMax = 15240;
static_assert(16ull * kTrialResidualMax * kTrialResidualMax < (1ull << 32), "sixteen squared residuals per 32-bit partial sum");
// (Written as instructions: left to itself the compiler turns the sum of squares into a chain of v_mad_u64_u32.)
__device__ __forceinline__ int mad24(int a, int b, int c) {                // a * b + c, a and b within 24 bits
    int o;
    asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(o) : "v"(a), "v"(b), "v"(c));
    return o;
}
// part += r^2 for r = v - mc * round(v / mc) clamped; neg_mc = -mc
__device__ __forceinline__ unsigned trial_residual_acc(unsigned part, int v, float vf, int neg_mc, float ic) {
    int c = (int)__builtin_rintf(vf * ic);
    c = c > 127 ? 127 : (c < -127 ? -127 : c);
    const int r = mad24(neg_mc, c, v);
    return (unsigned)mad24(r, r, (int)part);
}

__global__ __launch_bounds__(256) void k_coarse_build(const int8_t* __restrict__ planes, int64_t n, int64_t n_alloc,
                                                      int d_pad, int8_t* __restrict__ coarse,
                                                      CoarseRow* __restrict__ rows, int radix_mode) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= n_alloc) return;
    const int chunks = d_pad / 16;   // 16-byte chunks per plane row (d_pad is a multiple of 128)
    v4i* out = reinterpret_cast<v4i*>(coarse + row * (int64_t)d_pad);
    if (row >= n) {   // padding rows
        for (int k = lane; k < chunks; k += 64) out[k] = v4i{0, 0, 0, 0};
        if (lane == 0) rows[row] = CoarseRow{1, 0, 0, 0};
        return;
    }
    const v4i* lo = reinterpret_cast<const v4i*>(planes + row * 2 * (int64_t)d_pad);
    const v4i* hi = lo + chunks;
    int mx = 0;
    unsigned long long ss = 0;
    for (int k = lane; k < chunks; k += 64) {
        const v4i l4 = lo[k], h4 = hi[k];
#pragma unroll
        for (int w = 0; w < 4; ++w) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = (int)(int8_t)((uint32_t)l4[w] >> (8 * e)) + 256 * (int)(int8_t)((uint32_t)h4[w] >> (8 * e));
                const int av = v < 0 ? -v : v;
                mx = av > mx ? av : mx;
                ss += (unsigned)__mul24(v, v);   // |v| <= 32896: the square fits 32 bits
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
        ss += __shfl_xor(ss, o, 64);
    }
    int m = mx <= 127 ? 1 : (mx + 126) / 127;
    if (radix_mode == 1 && m > 1) {
        // The filter's bound grows with |r| (r = v - m c, c clamped to +-127): the radix that just avoids clamping is
        // not the one with the smallest residual -- sketch entries are bell shaped, a slightly smaller radix halves
        // the rounding error of ALL entries and clamps a handful of them.  Try 16 radices from ceil(max|v| / 127)
        // downwards and keep the one with the smallest sum of squared residuals (exact integers).
        const int step = m >= 64 ? m / 32 : 1;
        unsigned long long best = ~0ULL;
        int best_m = m;
        for (int t = 0; t < 16; ++t) {
            const int mc = m - t * step;
            if (mc < 1 || !radix_keeps_high_limb(mc, mx)) break;
            const float ic = 1.0f / (float)mc;
            unsigned long long r2c = 0;
            for (int k = lane; k < chunks; k += 64) {
                const v4i l4 = lo[k], h4 = hi[k];
                unsigned part = 0;                             // 16 squares of |r| <= kTrialResidualMax: fits (see there)
#pragma unroll
                for (int w = 0; w < 4; ++w)
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const int v = (int)(int8_t)((uint32_t)l4[w] >> (8 * e)) + 256 * (int)(int8_t)((uint32_t)h4[w] >> (8 * e));
                        part = trial_residual_acc(part, v, (float)v, -mc, ic);
                    }
                r2c += part;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r2c += __shfl_xor(r2c, o, 64);
            if (r2c < best) {
                best = r2c;
                best_m = mc;
            }
        }
        m = best_m;
    }
    const float inv = 1.0f / (float)m;
    unsigned c2 = 0, r2 = 0;   // <= 129^2 * 32768 per row: fits
    for (int k = lane; k < chunks; k += 64) {
        const v4i l4 = lo[k], h4 = hi[k];
        v4i o4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int v = (int)(int8_t)((uint32_t)l4[w] >> (8 * e)) + 256 * (int)(int8_t)((uint32_t)h4[w] >> (8 * e));
                int c = (int)rintf((float)v * inv);
                c = c > 127 ? 127 : (c < -127 ? -127 : c);
                const int r = mad24(-m, c, v);   // exact, whatever the rounding above did (24-bit factors: see mad24)
                c2 = (unsigned)mad24(c, c, (int)c2);
                r2 = (unsigned)mad24(r, r, (int)r2);
                packed |= (uint32_t)(uint8_t)(int8_t)c << (8 * e);
            }
            o4[w] = (int)packed;
        }
        out[k] = o4;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c2 += __shfl_xor(c2, o, 64);
        r2 += __shfl_xor(r2, o, 64);
    }
    if (lane == 0) {
        rows[row] = CoarseRow{m, (int32_t)c2, (int32_t)r2, ss >= (1ULL << 31) ? 1 : 0};
    }
}

// ---------------------------------------------------------------------------------------------------
// k_recode_rows<T, CH>: sketches -> two-limb planes + fragment-major coarse plane + row statistics in ONE pass (block plans:
// a rank re-codes its own rows every step; the three kernels this replaces -- k_limb_split, k_coarse_build,
// k_coarse_fm -- read or write every row five times).  One wave per row, a lane keeps CH chunks of 16 entries in registers
// (d_pad <= CH * 1024): the radix trials of k_coarse_build run on registers, the limb digits and the coarse bytes leave as
// 16-byte stores -- the coarse bytes straight into their place in the fragment-major plane (lane's 16 k values of row r at
// [(r / 16 * nk + k / 64) * 1024 + ((k / 16 % 4) * 16 + r % 16) * 16]; a workgroup is one group of 16 rows, so the sixteen
// 16-byte pieces of every 256-byte run arrive together).  Same digits, same coarse values, same statistics as the three
// kernels produce (the statistics are taken from the value the two limbs hold, as k_coarse_build reads it back).
// Rows [n_rows, count) of the range are written as zero rows (their planes are zero already: never written).
// ---------------------------------------------------------------------------------------------------
template <typename T, int CH, int RW>
__global__ __launch_bounds__(RW * 64) void k_recode_rows(const T* __restrict__ sk, int64_t n_rows, int64_t count, int d, int d_pad,
                                                      int8_t* __restrict__ planes, int8_t* __restrict__ coarse_fm,
                                                      CoarseRow* __restrict__ rows, int radix_mode) {
    const int lane = threadIdx.x & 63;
    // relative to the range's first row (a multiple of 16); RW rows per workgroup (8 where a lane holds 64 entries: the
    // 128 registers a 1024-thread workgroup leaves per lane spill there)
    const int64_t row = (int64_t)blockIdx.x * RW + (threadIdx.x >> 6);
    if (row >= count) return;
    const int nk = d_pad / 64;
    int v[CH][16];
    int mx = 0;
    unsigned long long ss = 0;
    const bool real = row < n_rows;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int k0 = (lane + 64 * c) * 16;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[c][e] = 0;
        if (real && k0 < d) {
            const T* src = sk + row * (int64_t)d + k0;
            if (k0 + 16 <= d && ((reinterpret_cast<uintptr_t>(src) & 15) == 0)) {
                if constexpr (sizeof(T) == 4) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const v4i x = *reinterpret_cast<const v4i*>(src + 4 * q);
                        v[c][4 * q] = x[0]; v[c][4 * q + 1] = x[1]; v[c][4 * q + 2] = x[2]; v[c][4 * q + 3] = x[3];
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < 2; ++q) {
                        const v4i x = *reinterpret_cast<const v4i*>(src + 8 * q);
#pragma unroll
                        for (int w = 0; w < 4; ++w) {
                            v[c][8 * q + 2 * w] = (int)(int16_t)((uint32_t)x[w] & 0xffffu);
                            v[c][8 * q + 2 * w + 1] = (int)(int16_t)((uint32_t)x[w] >> 16);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (k0 + e < d) v[c][e] = (int)src[e];
            }
        }
        // two signed base-256 digits, as k_limb_split takes them; from here on v is what those two digits hold
        if (k0 < d_pad) {
            v4i lo4, hi4;
#pragma unroll
            for (int w = 0; w < 4; ++w) {
                uint32_t pl = 0, ph = 0;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int x = v[c][4 * w + e];
                    const int l0 = (int)(int8_t)(x & 0xff);
                    const int l1 = (int)(int8_t)(((int32_t)((uint32_t)x - (uint32_t)l0) >> 8) & 0xff);
                    pl |= (uint32_t)(uint8_t)l0 << (8 * e);
                    ph |= (uint32_t)(uint8_t)l1 << (8 * e);
                    const int y = l0 + 256 * l1;
                    v[c][4 * w + e] = y;
                    const int ay = y < 0 ? -y : y;
                    mx = ay > mx ? ay : mx;
                    ss += (unsigned)__mul24(y, y);
                }
                lo4[w] = (int)pl;
                hi4[w] = (int)ph;
            }
            if (real) {
                *reinterpret_cast<v4i*>(planes + row * 2 * (int64_t)d_pad + k0) = lo4;
                *reinterpret_cast<v4i*>(planes + (row * 2 + 1) * (int64_t)d_pad + k0) = hi4;
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const int other = __shfl_xor(mx, o, 64);
        mx = other > mx ? other : mx;
        ss += __shfl_xor(ss, o, 64);
    }
    int m = mx <= 127 ? 1 : (mx + 126) / 127;
    if (radix_mode == 1 && m > 1) {          // k_coarse_build's search: the radix with the smallest residual among 16
        const int step = m >= 64 ? m / 32 : 1;
        unsigned long long best = ~0ULL;
        int best_m = m;
        for (int t = 0; t < 16; ++t) {
            const int mc = m - t * step;
            if (mc < 1 || !radix_keeps_high_limb(mc, mx)) break;
            const float ic = 1.0f / (float)mc;
            unsigned long long r2c = 0;
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                unsigned part = 0;
#pragma unroll
                for (int e = 0; e < 16; ++e) part = trial_residual_acc(part, v[c][e], (float)v[c][e], -mc, ic);
                r2c += part;
            }
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) r2c += __shfl_xor(r2c, o, 64);
            if (r2c < best) {
                best = r2c;
                best_m = mc;
            }
        }
        m = best_m;
    }
    const float inv = 1.0f / (float)m;
    unsigned c2 = 0, r2 = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        const int kc = lane + 64 * c;
        if (kc * 16 >= d_pad) continue;
        v4i o4;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            uint32_t packed = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int x = v[c][4 * w + e];
                int cc = (int)rintf((float)x * inv);
                cc = cc > 127 ? 127 : (cc < -127 ? -127 : cc);
                const int r = mad24(-m, cc, x);
                c2 = (unsigned)mad24(cc, cc, (int)c2);
                r2 = (unsigned)mad24(r, r, (int)r2);
                packed |= (uint32_t)(uint8_t)(int8_t)cc << (8 * e);
            }
            o4[w] = (int)packed;
        }
        *reinterpret_cast<v4i*>(coarse_fm + ((row >> 4) * nk + (kc >> 2)) * 1024 + (((kc & 3) << 4) + (row & 15)) * 16) = o4;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c2 += __shfl_xor(c2, o, 64);
        r2 += __shfl_xor(r2, o, 64);
    }
    if (lane == 0) rows[row] = real ? CoarseRow{m, (int32_t)c2, (int32_t)r2, ss >= (1ULL << 31) ? 1 : 0} : CoarseRow{1, 0, 0, 0};
}

__global__ __launch_bounds__(256) void k_rows_needed(const PairwiseArgs a, int n_tr, int n_tc, long long f0, long long f1, long long n_rows,
                                                     unsigned char* __restrict__ need) {
    unsigned long long n_cand = *reinterpret_cast<volatile const unsigned long long*>(a.cand_counter);
    n_cand = n_cand < a.cand_capacity ? n_cand : a.cand_capacity;
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n_cand; i += stride) {
        const long long col = a.cand[i].y & 0x7fffffff;
        if ((col < f0 || col >= f1) && col < n_rows) need[col] = 1;
    }
    // flagged tiles (a workgroup per tile it finds flagged: one thread per column)
    for (int t = blockIdx.x; t < n_tr * n_tc; t += gridDim.x) {
        if (a.tile_flag[t] == 0u) continue;
        const long long col = a.col_begin + (long long)(t % n_tc) * 256 + threadIdx.x;
        if ((col < f0 || col >= f1) && col < n_rows) need[col] = 1;
    }
}

// k_planes_from_wire: limb planes of rows whose LOW limb arrived in a wire buffer (lo[row * d_pad + k]) and whose coarse
// plane and statistics are in place (fragment-major, as the filter reads them): both limb rows are written -- the rule is
// at radix_keeps_high_limb.  A workgroup takes 16 rows (one KiB of the fragment-major plane holds 16 rows x 64 k), a wave one.
// need != NULL: only the rows marked there (k_rows_needed: what a plan's re-check and flagged tiles will read).
// Groups [skip0, skip1) (units of 16 rows) are left alone: a plan's own frame, so that ONE launch covers the rows on both sides.
// [r6] Measured on one rank's step of an 8-way split of 100k x 2048 (about a quarter of the 75 000 foreign rows wanted; a 128-byte
// line of the coarse plane holds 8 rows, so nearly all 154 MB of it are read; 256 MB in all by the counters): 16 rows x 4 chunks
// per wave, one launch per side of the frame and per 256 k: 4.6 + 83.3 us; 4 x 16 or 8 x 8 per wave in one launch: 86; the same with
// a thread walking its whole row, all loads in flight: 77-79; a wave per row (below): 66 us = 3.9 TB/s.
__device__ __forceinline__ v4i high_limbs_from_wire(const v4i c4, const v4i l4, int m) {
    const int h = (m + 1) >> 1, edge = 127 * m - h + 127;
    v4i h4;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
        uint32_t ph = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int c = (int)(int8_t)((uint32_t)c4[w] >> (8 * e));
            const int l0 = (int)(int8_t)((uint32_t)l4[w] >> (8 * e));
            const int t = c == 127 ? edge : (c == -127 ? -edge : m * c);
            // v = t + wrap8(l0 - t) is the value congruent to l0 near t, and (v - l0) / 256 = floor((t - l0 + 127) / 256)
            ph |= (uint32_t)(uint8_t)((t - l0 + 127) >> 8) << (8 * e);
        }
        h4[w] = (int)ph;
    }
    return h4;
}

// A WAVE PER ROW (16 waves = the 16 rows of a group per workgroup), a lane per 16-k chunk, KJ rounds of 64 chunks with all their
// loads in flight: a wave either has its row wanted -- then all 64 lanes work -- or leaves at once.  With several rows per wave
// (8 x 8, 4 x 16, 16 x 4 were all tried) three lanes of four idle through the ~12 instructions per entry of the rule whenever
// one row of the wave is wanted (a quarter to a half of the rows are), and the counters showed the kernel busy, not waiting
// (SQ_WAIT_ANY 0.27 of the wave cycles, 256 MB at 3.4 TB/s).  The coarse pieces of one row are 16 bytes in every 256: the
// wanted rows of a group run on ONE CU at the same time and meet in its L1 / the XCD's L2 on the lines they share.
template <int KJ>
__global__ __launch_bounds__(1024) void k_planes_from_wire(const int8_t* __restrict__ lo_wire, const int8_t* __restrict__ coarse_fm,
                                                           const CoarseRow* __restrict__ rows, int64_t count, int d_pad,
                                                           int8_t* __restrict__ planes, const unsigned char* __restrict__ need,
                                                           int64_t skip0, int64_t skip1) {
    const int nk = d_pad / 64;
    int64_t grp = blockIdx.x;                              // 16 rows
    if (grp >= skip0) grp += skip1 - skip0;
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int64_t row = grp * 16 + r;
    if (row >= count || (need && need[row] == 0)) return;
    const int m = rows[row].radix;
    const int chunks = d_pad / 16;
    const int8_t* cbase = coarse_fm + grp * nk * 1024 + r * 16;
    const int8_t* lbase = lo_wire + row * (int64_t)d_pad;
    int8_t* p0 = planes + row * 2 * (int64_t)d_pad;
    int8_t* p1 = p0 + d_pad;
    for (int kc0 = lane; kc0 < chunks; kc0 += 64 * KJ) {
        v4i c4[KJ], l4[KJ];
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int kc = kc0 + 64 * j;
            if (kc < chunks) {
                c4[j] = *reinterpret_cast<const v4i*>(cbase + (kc >> 2) * 1024 + ((kc & 3) << 8));
                l4[j] = *reinterpret_cast<const v4i*>(lbase + kc * 16);
            }
        }
#pragma unroll
        for (int j = 0; j < KJ; ++j) {
            const int kc = kc0 + 64 * j;
            if (kc < chunks) {
                *reinterpret_cast<v4i*>(p0 + kc * 16) = l4[j];
                *reinterpret_cast<v4i*>(p1 + kc * 16) = high_limbs_from_wire(c4[j], l4[j], m);
            }
        }
    }
}

// per-call filter constants {s, w, a, p} (see above); padding rows never pass (s = +inf)
__global__ __launch_bounds__(256) void k_filter_meta(const CoarseRow* __restrict__ rows, const double* __restrict__ n2,
                                                     int64_t n, int64_t n_alloc, int d, double coeff,
                                                     float4* __restrict__ meta) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_alloc) return;
    float4 o = make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    if (i < n) {
        const CoarseRow st = rows[i];
        const double m = (double)st.radix;
        const double eps = 1.0 / 4096.0;
        const double tau = coeff * (double)d * n2[i] / m;          // NaN stays NaN: such a row is never kept
        // big rows: dots may wrap, always re-check.  Negative squared norms (never produced by the reference's
        // loader, but callers pass arbitrary doubles) too: with a negative threshold the truncating keep test no
        // longer implies P > d * threshold.
        o.x = (st.big || n2[i] < 0.0) ? -__builtin_inff() : (float)(tau - fabs(tau) * eps);
        o.y = (float)(1.0 / m);
        o.z = (float)(sqrt((double)st.c2) * (1.0 + eps));
        o.w = (float)(sqrt((double)st.r2) / m * (1.0 + eps));
    }
    meta[i] = o;
}

}  // namespace

int launch_max_abs(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_elems,
                   unsigned long long* d_out) {
    if (n_elems == 0) return 0;
    int64_t blocks = (n_elems + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (elem_bytes == 4)
        hipLaunchKernelGGL(k_max_abs<int32_t>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const int32_t*)d_sk, n_elems, d_out);
    else
        hipLaunchKernelGGL(k_max_abs<int16_t>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const int16_t*)d_sk, n_elems, d_out);
    return 0;
}

int launch_limb_split(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_rows, int d, int limbs,
                      int8_t* d_planes, int d_pad, int64_t row_offset) {
    if (n_rows == 0) return 0;
    const int64_t total = n_rows * (d_pad / 4);
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (elem_bytes == 4)
        hipLaunchKernelGGL(k_limb_split<int32_t>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const int32_t*)d_sk, n_rows, d, limbs, d_planes, d_pad, row_offset);
    else
        hipLaunchKernelGGL(k_limb_split<int16_t>, dim3((unsigned)blocks), dim3(256), 0, stream,
                           (const int16_t*)d_sk, n_rows, d, limbs, d_planes, d_pad, row_offset);
    return 0;
}

int launch_cand_thr(hipStream_t stream, const double* d_norms_sq, int64_t n, int64_t n_alloc, int d,
                    double coeff, int32_t* d_thr) {
    hipLaunchKernelGGL(k_cand_thr, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, stream, d_norms_sq, n,
                       n_alloc, d, coeff, d_thr);
    return 0;
}

int launch_coarse_build(hipStream_t stream, const int8_t* d_planes, int64_t n, int64_t n_alloc, int d_pad,
                        int8_t* d_coarse, CoarseRow* d_rows, int radix_mode) {
    if (n_alloc <= 0) return 0;
    hipLaunchKernelGGL(k_coarse_build, dim3((unsigned)((n_alloc + 3) / 4)), dim3(256), 0, stream, d_planes, n, n_alloc,
                       d_pad, d_coarse, d_rows, radix_mode);
    return 0;
}

// row-major plane(s) -> fragment-major (PairwiseArgs::coarse_fm, planes_fm): one wave per KiB, written as whole lines.
// Source row of (sample, limb) = (sample * limbs + limb) * d_pad; chunk ch = ((sample / 16) * limbs + limb) * nk + k / 64
__global__ __launch_bounds__(256) void k_coarse_fm(const int8_t* __restrict__ coarse, long long chunks, int nk, int d_pad, int limbs,
                                                   int8_t* __restrict__ fm) {
    const int lane = threadIdx.x & 63;
    const int fr = lane & 15, fq = lane >> 4;
    for (long long ch = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); ch < chunks; ch += (long long)gridDim.x * 4) {
        const long long bl = ch / nk;                              // (sample block, limb)
        const int ks = (int)(ch - bl * nk);
        const long long blk = bl / limbs;
        const int limb = (int)(bl - blk * limbs);
        const v4i v = *reinterpret_cast<const v4i*>(coarse + ((blk * 16 + fr) * limbs + limb) * (long long)d_pad + ks * 64 + fq * 16);
        *reinterpret_cast<v4i*>(fm + ch * 1024 + lane * 16) = v;
    }
}

int launch_coarse_fm(hipStream_t stream, const int8_t* d_coarse, int64_t n_alloc, int d_pad, int8_t* d_fm, int limbs) {
    if (n_alloc <= 0) return 0;
    const int nk = d_pad / 64;
    const long long chunks = (long long)(n_alloc / 16) * limbs * nk;
    const unsigned grid = (unsigned)std::min<long long>((chunks + 3) / 4, 65536);
    hipLaunchKernelGGL(k_coarse_fm, dim3(grid), dim3(256), 0, stream, d_coarse, chunks, nk, d_pad, limbs, d_fm);
    return 0;
}

// rows [0, count) (count a multiple of 16; all pointers at the range's first row, which is a multiple of 16) except rows
// [skip_first, skip_first + skip_count) (multiples of 16 too; 0, 0: none)
int launch_planes_from_wire(hipStream_t stream, const int8_t* d_lo_wire, const int8_t* d_coarse_fm, const CoarseRow* d_rows,
                            int64_t count, int d_pad, int8_t* d_planes, const unsigned char* d_need, int64_t skip_first,
                            int64_t skip_count) {
    const int64_t groups = count / 16 - skip_count / 16;
    if (groups <= 0) return 0;
    hipLaunchKernelGGL(k_planes_from_wire<2>, dim3((unsigned)groups), dim3(1024), 0, stream, d_lo_wire, d_coarse_fm, d_rows, count, d_pad,
                       d_planes, d_need, skip_first / 16, (skip_first + skip_count) / 16);
    return 0;
}

// need[row] = 1 for the storage rows OUTSIDE the frame [f0, f1) that a plan's second half reads: the columns of its candidates
// (count on the device, as the filter launches left it) and the 256 columns of every flagged tile; need is zero on entry
int launch_rows_needed(hipStream_t stream, const PairwiseArgs& a, int n_tr, int n_tc, int64_t f0, int64_t f1, int64_t n_rows,
                       unsigned char* d_need) {
    hipLaunchKernelGGL(k_rows_needed, dim3(512), dim3(256), 0, stream, a, n_tr, n_tc, (long long)f0, (long long)f1, (long long)n_rows, d_need);
    return 0;
}

// sketches (n_rows x d, device) -> limb planes, fragment-major coarse plane and statistics of `count` rows (a multiple of 16;
// rows beyond n_rows: zero rows), all pointers at the range's first row.  false: this sketch length has no fused kernel
bool launch_recode_rows(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_rows, int64_t count, int d, int d_pad,
                        int8_t* d_planes, int8_t* d_coarse_fm, CoarseRow* d_rows, int radix_mode, int rows_per_wg) {
    if (count <= 0) return true;
    if (d_pad > 4096 || (count & 15)) return false;
    const int ch = d_pad <= 1024 ? 1 : (d_pad <= 2048 ? 2 : 4);
    // 8 rows per workgroup: two workgroups share a CU and are in different phases (loads / radix trials / stores), and the
    // eight 16-byte pieces of a fragment-major run still fill a 128-byte line; 16 rows (one 1024-thread workgroup per CU:
    // all of its waves load, compute and store in step) only where the option asks for it and the registers allow
    const bool wide = rows_per_wg == 16 && ch < 4;
#define MVS_RECODE(T, CH, RW) hipLaunchKernelGGL((k_recode_rows<T, CH, RW>), dim3((unsigned)(count / RW)), dim3(RW * 64), 0, stream, (const T*)d_sk, n_rows, count, d, d_pad, d_planes, d_coarse_fm, d_rows, radix_mode)
#define MVS_RECODE_T(T)                                                     \
    do {                                                                    \
        if (ch == 4) MVS_RECODE(T, 4, 8);                                   \
        else if (ch == 2) { if (wide) MVS_RECODE(T, 2, 16); else MVS_RECODE(T, 2, 8); } \
        else { if (wide) MVS_RECODE(T, 1, 16); else MVS_RECODE(T, 1, 8); }  \
    } while (0)
    if (elem_bytes == 4) MVS_RECODE_T(int32_t);
    else MVS_RECODE_T(int16_t);
#undef MVS_RECODE_T
#undef MVS_RECODE
    return true;
}

int launch_filter_meta(hipStream_t stream, const CoarseRow* d_rows, const double* d_norms_sq, int64_t n,
                       int64_t n_alloc, int d, double coeff, float4* d_meta) {
    hipLaunchKernelGGL(k_filter_meta, dim3((unsigned)((n_alloc + 255) / 256)), dim3(256), 0, stream, d_rows,
                       d_norms_sq, n, n_alloc, d, coeff, d_meta);
    return 0;
}

}  // namespace mvs
