// mvs_encode.hip -- the shard codec on the device: CSR rows (ascending columns + quantised Jaccard) -> the byte records
// the shard writer appends to matrix.bin.
//
// Reference: write_sparse_results_jaccard_wo_sort (src/pairwise_comp_optimized.cpp:718-736) stores per row a
// bits::compact_vector of the q values and, when the row holds more than one cell, a bits::rice_sequence of the column
// deltas.  The `bits` submodule is absent from the reference tree, so the byte layout is this build's own
// (csrc/host/mvs_codec.hpp documents it); the kernels below produce exactly the bytes mvs_codec::compact_vector::save and
// mvs_codec::rice_sequence::save produce for the same values (tests compare files byte for byte):
//   compact_vector : [size][width][n_words][words...]            value i at bit i*width, width = bits of the largest value
//   rice_sequence  : [size][k] [low: compact_vector of width k, absent when k == 0]
//                    [n_high_bits][n_words][high words...]       unary quotients: q zeros then a one
//                    [n_samples][sample...]                      bit position before every 64th element
//                    k = floor(log2(floor(mean))) for a mean above 1, else 0
// All fields are little-endian u64 words, so a record is a whole number of words and 64 consecutive values of width w fill
// exactly w words: one wave packs a row in chunks of 64 values without ever sharing a word between chunks.  Only the unary
// part is irregular: a chunk's bits are collected in LDS and leave as whole words, the word two chunks share is carried over
// (a chunk spanning more words than the stage holds -- one enormous column gap in a row of tiny ones -- falls back to atomic
// ORs into the words the caller has zeroed).
//
// Why on the device: for a dense result the host encoder bounds the executable (1e9 cells: 1.3 s on 16 cores against
// 0.1 s of comparison + download), and the encoded rows are 1.4 bytes per cell on the link instead of 5.
#include "mvs_encode.h"

#include <algorithm>

#include <rocprim/device/device_scan.hpp>

#include "../../include/mvs_hip.h"

namespace mvs {

namespace {

using u64 = unsigned long long;

__device__ __forceinline__ unsigned bit_width_u64(u64 v) { return v ? 64u - (unsigned)__builtin_clzll(v) : 0u; }

// One wave per row (workgroup = one wave).
template <typename Q>
__global__ __launch_bounds__(64) void k_enc_size(const long long* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                 const Q* __restrict__ q, u64* __restrict__ size, unsigned* __restrict__ jac,
                                                 unsigned* __restrict__ first_col, EncRow* __restrict__ par) {
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = row_ptr[r], e = row_ptr[r + 1];
    const u64 n = (u64)(e - b);
    if (n == 0) {
        if (lane == 0) {
            size[r] = 0;
            jac[r] = 0;
            first_col[r] = 0;
            par[r] = EncRow{0, 0, 0};
        }
        return;
    }
    unsigned mx = 0;
    for (long long i = b + lane; i < e; i += 64) {
        const unsigned v = (unsigned)q[i];
        mx = v > mx ? v : mx;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned other = (unsigned)__shfl_xor((int)mx, o, 64);
        mx = other > mx ? other : mx;
    }
    const unsigned wq = mx ? bit_width_u64(mx) : 1u;                     // compact_vector::build: width of the largest, at least 1
    const u64 jac_bytes = 8 * (3 + (n * wq + 63) / 64);
    u64 total = jac_bytes, high = 0;
    unsigned k = 0;
    if (n > 1) {
        const u64 nr = n - 1;
        const u64 sum = (u64)(col[e - 1] - col[b]);                      // the deltas telescope
        const u64 mean = sum / nr;
        k = mean > 1 ? bit_width_u64(mean) - 1 : 0;
        u64 s = 0;
        for (long long j = b + lane; j < e - 1; j += 64) s += (u64)(unsigned)(col[j + 1] - col[j]) >> k;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) s += (u64)__shfl_xor((long long)s, o, 64);
        high = nr + s;
        total += 8 * (5 + (high + 63) / 64 + (nr + 63) / 64) + (k ? 8 * (3 + (nr * k + 63) / 64) : 0);
    }
    if (lane == 0) {
        size[r] = total;
        jac[r] = (unsigned)jac_bytes;
        first_col[r] = (unsigned)col[b];
        par[r] = EncRow{high, wq, k};
    }
}

// The workgroup IS one wave, and a wave's LDS operations execute in the order it issued them: between a phase that writes a
// stage and one that reads it nothing has to be waited for, the compiler just must not move LDS accesses across the line.
// (__syncthreads() would also do, but hipcc puts `s_waitcnt vmcnt(0)` in front of a barrier: the global loads the fill loop
// keeps in flight for LATER chunks would be drained at every chunk, which is exactly the latency the loop is built to hide.)
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// value v (already masked to `width` bits) of slot `lane` into a zeroed stage of `width` words: bytes and shorts are plain
// stores (a lane owns its byte), other widths OR their bits in
__device__ __forceinline__ void stage_put(u64* stage, int lane, unsigned width, u64 v) {
    if (width == 8) {
        reinterpret_cast<unsigned char*>(stage)[lane] = (unsigned char)v;
    } else if (width == 16) {
        reinterpret_cast<unsigned short*>(stage)[lane] = (unsigned short)v;
    } else {
        const unsigned p = (unsigned)lane * width, w = p >> 6, off = p & 63u;
        atomicOr(&stage[w], v << off);
        if (off + width > 64) atomicOr(&stage[w + 1], v >> (64 - off));
    }
}

// ---- the common row: 8-bit q values, Rice parameter at most 16, at least two cells ----
// 256 cells per iteration, FOUR consecutive cells per lane: a quarter of the prefix sums, shuffles, stage hand-overs and
// write-outs per cell of the general loop below (which spends ~390 instructions per chunk of 64 cells, whatever its LDS
// atomics cost), the q values leave as one dword per lane without touching LDS, a lane's four low-bit fields (4 k bits) and
// its four unary codes (a window of at most 64 bits when the lane's codes are that short) go into the stage words with one
// or two ORs each.  The record layout is the one described at the top; the general loop handles everything else.
struct FastRowIn {
    int c[5];             // columns of cells i0 .. i0 + 3 and of cell i0 + 4 (indices clamped to the row)
    unsigned q;           // the four q bytes (cells beyond the row: the last cell's byte, masked out when used)
};

template <typename Q>
__device__ __forceinline__ void enc_row_fast(const int32_t* __restrict__ col, const Q* __restrict__ q, long long b, unsigned n,
                                             const EncRow pr, u64* __restrict__ w, int lane, u64 (&st_l)[2][64], u64 (&st_h)[2][64]) {
    const unsigned k = pr.k, nr = n - 1;
    const u64 wq_words = ((u64)n * 8 + 63) / 64;
    u64* qdst = w + 3;
    u64* z = qdst + wq_words;
    const u64 low_words = k ? ((u64)nr * k + 63) / 64 : 0;
    const u64 idx = k ? 5 + low_words : 2;
    u64* ldst = z + 5;
    const u64 hw = (pr.high_bits + 63) / 64, ns = ((u64)nr + 63) / 64;
    u64* high = z + idx + 2;
    u64* samples = high + hw + 1;
    if (lane == 0) {
        w[0] = n;
        w[1] = 8;
        w[2] = wq_words;
        z[0] = nr;
        z[1] = k;
        if (k) {
            z[2] = nr;
            z[3] = k;
            z[4] = low_words;
        }
        z[idx] = pr.high_bits;
        z[idx + 1] = hw;
        high[hw] = ns;
    }
    const unsigned lmask = (1u << k) - 1u;
    const int32_t* cb = col + b;
    const Q* qb = q + b;
    auto load = [&](unsigned c0, FastRowIn& in) {
        const unsigned i0 = c0 + 4u * (unsigned)lane;
#pragma unroll
        for (int j = 0; j < 5; ++j) {
            const unsigned i = i0 + (unsigned)j;
            in.c[j] = cb[i < n ? i : nr];
        }
        unsigned qq = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const unsigned i = i0 + (unsigned)j;
            qq |= (unsigned)qb[i < n ? i : nr] << (8 * j);
        }
        in.q = qq;
    };
    u64 base = 0, carry = 0;
    bool direct = false;
    unsigned it = 0;
    auto iteration = [&](unsigned c0, const FastRowIn& in, FastRowIn& nxt) __attribute__((always_inline)) {
        load(c0 + 256, nxt);                                   // in flight while this iteration is packed (clamped: harmless past the row)
        u64* sl = st_l[it & 1];
        u64* sh = st_h[it & 1];
        const unsigned i0 = c0 + 4u * (unsigned)lane;
        // deltas, quotients, code lengths of the lane's four cells
        unsigned quot[4], low[4], len[4];
        unsigned L = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool has = i0 + (unsigned)j < nr;
            const unsigned d = has ? (unsigned)(in.c[j + 1] - in.c[j]) : 0u;
            quot[j] = d >> k;
            low[j] = d & lmask;
            len[j] = has ? quot[j] + 1u : 0u;
            L += len[j];
        }
        unsigned incl = L;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        const unsigned T = (unsigned)__shfl((int)incl, 63, 64);
        const unsigned excl = incl - L;
        const unsigned b63 = (unsigned)(base & 63);
        const u64 w0 = base >> 6;
        const unsigned nw = (b63 + T + 63u) >> 6;               // words this iteration's unary codes touch
        const bool has_d = c0 < nr;                            // (the last iteration of a row may hold its last cell only)
        const bool staged = has_d && !direct && nw <= 64u;
        const unsigned lw = 4u * k;                            // low words of a full iteration (256 k bits)
        if ((unsigned)lane < lw) sl[lane] = 0;
        if (staged && (unsigned)lane < nw) sh[lane] = lane == 0 ? carry : 0;
        wave_sync();
        // q values: one dword per lane, straight to the record (the caller has zeroed it: the half word past the row stays 0)
        if (i0 < n) {
            const unsigned valid = n - i0 >= 4u ? 0xffffffffu : (1u << (8u * (n - i0))) - 1u;
            reinterpret_cast<unsigned*>(qdst)[i0 >> 2] = in.q & valid;
        }
        if (has_d) {
            if (k && i0 < nr) {                                // the lane's four low-bit fields: 4 k <= 64 bits at bit 4 k lane
                u64 f = 0;
#pragma unroll
                for (int j = 3; j >= 0; --j) f = (f << k) | (u64)low[j];
                const unsigned p = (unsigned)lane * lw, sw = p >> 6, so = p & 63u;
                atomicOr(&sl[sw], f << so);
                if (so + lw > 64u) atomicOr(&sl[sw + 1], f >> (64u - so));
            }
            if ((i0 & 63u) == 0u && i0 < nr) samples[i0 >> 6] = base + excl;     // bit position before every 64th element
            if (staged) {
                if (L) {
                    const unsigned s0 = b63 + excl;             // the lane's first code starts here (relative to word w0)
                    if (L <= 64u) {
                        u64 win = 0;
                        unsigned at = 0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (len[j]) win |= 1ULL << (at + quot[j]);
                            at += len[j];
                        }
                        const unsigned sw = s0 >> 6, so = s0 & 63u;
                        atomicOr(&sh[sw], win << so);
                        if (so && (win >> (64u - so))) atomicOr(&sh[sw + 1], win >> (64u - so));
                    } else {
                        unsigned at = s0;
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            if (len[j]) {
                                const unsigned pos = at + quot[j];
                                atomicOr(&sh[pos >> 6], 1ULL << (pos & 63u));
                            }
                            at += len[j];
                        }
                    }
                }
            } else {
                if (!direct) {
                    if (lane == 0 && carry) atomicOr(&high[w0], carry);
                    carry = 0;
                    direct = true;
                }
                u64 at = base + excl;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (len[j]) {
                        const u64 pos = at + quot[j];
                        atomicOr(&high[pos >> 6], 1ULL << (pos & 63));
                    }
                    at += len[j];
                }
            }
        }
        wave_sync();
        if (k && has_d) {
            const unsigned remd = nr - c0 < 256u ? nr - c0 : 256u;
            if ((unsigned)lane < (remd * k + 63u) / 64u) ldst[(u64)(c0 >> 6) * k + (u64)lane] = sl[lane];
        }
        if (staged) {
            const bool ends_on_border = ((b63 + T) & 63u) == 0u;
            const unsigned full = ends_on_border ? nw : nw - 1u;  // words that no later iteration adds to
            if ((unsigned)lane < full) high[w0 + (u64)lane] = sh[lane];
            carry = ends_on_border ? 0 : sh[nw - 1u];
        }
        base += T;
        ++it;
    };
    FastRowIn A, B;
    load(0, A);
    for (unsigned c0 = 0; c0 < n; c0 += 512) {
        iteration(c0, A, B);
        if (c0 + 256 >= n) break;
        iteration(c0 + 256, B, A);
    }
    if (lane == 0 && carry) high[base >> 6] = carry;
}

// Pass 2.  ONE loop over the row's chunks of 64 cells does all three containers of the record -- the q values, the low
// bits of the column deltas, their unary quotients -- so the columns are read once, and the loads run ahead of their use
// (columns two chunks, q one chunk): a wave's chunks depend on each other only through the bit position of the unary part,
// and with the loads inside that chain the kernel was bound by their latency (12 ms for 1e9 cells, 157 chunks per row one
// after the other; 3 passes over the row).  The LDS stages alternate between two sets (a chunk zeroes the set the previous one does not read).
template <typename Q>
__global__ __launch_bounds__(64) void k_enc_fill(const long long* __restrict__ row_ptr, const int32_t* __restrict__ col,
                                                 const Q* __restrict__ q, const u64* __restrict__ offset,
                                                 const EncRow* __restrict__ par, unsigned char* __restrict__ out, unsigned stage_words,
                                                 int fast_rows, u64 cap_cells, u64 cap_bytes) {
    __shared__ u64 st_q[2][64], st_l[2][64], st_h[2][64];
    const long long r = blockIdx.x;
    const int lane = threadIdx.x;
    const long long b = row_ptr[r], e = row_ptr[r + 1];
    const u64 n = (u64)(e - b);
    if (n == 0) return;
    // buffers sized from the previous row block (no read-back of this block's totals in front of the launch): a row whose cells or
    // record lie beyond them is skipped -- the caller reads the totals afterwards and does the block again
    if ((u64)e > cap_cells || offset[r + 1] > cap_bytes) return;
    const EncRow pr = par[r];
    const unsigned wq = pr.wq, k = pr.k;
    u64* w = reinterpret_cast<u64*>(out + offset[r]);
    if (sizeof(Q) == 1 && fast_rows && wq == 8 && k <= 16 && n >= 2 && n < (1ULL << 31)) {       // row-uniform
        enc_row_fast<Q>(col, q, b, (unsigned)n, pr, w, lane, st_l, st_h);
        return;
    }
    const u64 wq_words = (n * wq + 63) / 64;
    const u64 nr = n - 1;                                                // deltas; :732 a single-entry row has no delta sequence
    u64* qdst = w + 3;
    u64* z = qdst + wq_words;                                            // the rice_sequence (n >= 2)
    const u64 low_words = k ? (nr * k + 63) / 64 : 0;
    const u64 idx = k ? 5 + low_words : 2;
    u64* ldst = z + 5;
    const u64 hw = (pr.high_bits + 63) / 64, ns = (nr + 63) / 64;
    u64* high = z + idx + 2;
    u64* samples = high + hw + 1;
    if (lane == 0) {
        w[0] = n;                                                        // compact_vector header
        w[1] = wq;
        w[2] = wq_words;
        if (n >= 2) {
            z[0] = nr;
            z[1] = k;
            if (k) {
                z[2] = nr;
                z[3] = k;
                z[4] = low_words;
            }
            z[idx] = pr.high_bits;
            z[idx + 1] = hw;
            high[hw] = ns;
        }
    }
    const u64 qmask = wq >= 64 ? ~0ULL : ((1ULL << wq) - 1ULL), lmask = (1ULL << k) - 1ULL;
    // loads beyond the row are clamped into it (their values are never used): a conditional load is a branch, and hipcc
    // waits for the load at the end of the branch -- no load would stay in flight across the chunk
    auto ld_col = [&](u64 i) -> int { return col[b + (long long)(i < n ? i : nr)]; };
    auto ld_q = [&](u64 i) -> unsigned { return (unsigned)q[b + (long long)(i < n ? i : nr)]; };
    // The loads of chunk t + 2 are issued while chunk t is packed.  Three register sets take turns (the loop is unrolled by
    // three with the roles rotated) -- handing the values down at the end of an iteration would be register moves that wait
    // for the loads.
    int cA = ld_col((u64)lane), cB = ld_col(64 + (u64)lane), cC = 0;
    unsigned qA = ld_q((u64)lane), qB = ld_q(64 + (u64)lane), qC = 0;
    // unary part: element j sits at bit (sum over i < j of (quotient_i + 1)) + quotient_j; a wave prefix sum per chunk.  A
    // chunk's bits are collected in LDS and leave as whole words; the word a chunk ends in (shared with the next chunk
    // unless it ends on a word border) is carried over instead of written.  (One global atomic OR per element, the first
    // version, cost 60 ms on 1e9 cells: atomics execute at the memory side, 64 bytes of traffic each.)  A chunk whose
    // quotients are so large that it spans more words than the stage holds ORs its bits into memory (zeroed by the caller),
    // and so does every later chunk of the row: the word they start in may already hold bits there.
    u64 base = 0, carry = 0;                                             // carry: the bits of word base >> 6 set so far
    bool direct = false;
    auto chunk = [&](u64 c0, int c_cur, int c_nxt, int& c_ld, unsigned q_cur, unsigned& q_ld) __attribute__((always_inline)) {
        c_ld = ld_col(c0 + 128 + (u64)lane);                             // in flight while this chunk and the next are packed
        q_ld = ld_q(c0 + 128 + (u64)lane);
        const unsigned it = (unsigned)(c0 >> 6);
        u64* sq = st_q[it & 1];
        u64* sl = st_l[it & 1];
        u64* sh = st_h[it & 1];
        const u64 i = c0 + (u64)lane;
        int succ = __shfl_down(c_cur, 1, 64);
        const int edge = __shfl(c_nxt, 0, 64);
        if (lane == 63) succ = edge;
        const bool has_d = i < nr;
        const u64 dlt = has_d ? (u64)(unsigned)(succ - c_cur) : 0;
        const u64 quot = dlt >> k;
        const u64 len = has_d ? quot + 1 : 0;
        // 32-bit prefix sum: the quotients of a ROW sum to less than 2 n (k = floor(log2(mean delta))), so a chunk's to less
        // than 2^32 -- half the cross-lane traffic of a 64-bit scan
        unsigned incl32 = (unsigned)len;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl32, o, 64);
            if (lane >= o) incl32 += up;
        }
        const u64 incl = incl32;
        const u64 total = (u64)(unsigned)__shfl((int)incl32, 63, 64);
        const bool chunk_has_d = c0 < nr;
        const u64 w0 = base >> 6, nw = ((base & 63) + total + 63) / 64;   // the words this chunk's unary codes touch
        const bool staged = chunk_has_d && !direct && nw <= (u64)stage_words;
        if ((unsigned)lane < wq) sq[lane] = 0;
        if ((unsigned)lane < k) sl[lane] = 0;
        if (staged && (u64)lane < nw) sh[lane] = lane == 0 ? carry : 0;
        wave_sync();
        if (i < n) stage_put(sq, lane, wq, (u64)q_cur & qmask);
        if (k && has_d) stage_put(sl, lane, k, dlt & lmask);
        if (chunk_has_d) {
            if (lane == 0) samples[c0 / 64] = base;
            if (staged) {
                if (has_d) {
                    const u64 pos = (base & 63) + (incl - len) + quot;   // relative to word w0
                    atomicOr(&sh[pos >> 6], 1ULL << (pos & 63));
                }
            } else {
                if (!direct) {
                    if (lane == 0 && carry) atomicOr(&high[w0], carry);
                    carry = 0;
                    direct = true;
                }
                if (has_d) {
                    const u64 pos = base + (incl - len) + quot;
                    atomicOr(&high[pos >> 6], 1ULL << (pos & 63));
                }
            }
        }
        wave_sync();
        const u64 rem = n - c0 < 64 ? n - c0 : 64;
        if ((u64)lane < (rem * wq + 63) / 64) qdst[(c0 / 64) * wq + (u64)lane] = sq[lane];
        if (k && chunk_has_d) {
            const u64 remd = nr - c0 < 64 ? nr - c0 : 64;
            if ((u64)lane < (remd * k + 63) / 64) ldst[(c0 / 64) * k + (u64)lane] = sl[lane];
        }
        if (staged) {
            const bool ends_on_border = ((base + total) & 63) == 0;
            const u64 full = ends_on_border ? nw : nw - 1;               // words that no later chunk adds to
            if ((u64)lane < full) high[w0 + (u64)lane] = sh[lane];
            carry = ends_on_border ? 0 : sh[nw - 1];
        }
        base += total;
    };
    for (u64 c0 = 0; c0 < n; c0 += 192) {
        chunk(c0, cA, cB, cC, qA, qC);
        if (c0 + 64 >= n) break;
        chunk(c0 + 64, cB, cC, cA, qB, qA);
        if (c0 + 128 >= n) break;
        chunk(c0 + 128, cC, cA, cB, qC, qB);
    }
    if (lane == 0 && carry) high[base >> 6] = carry;
}


// ---- a sorted cell list -> the CSR arrays the encoder reads (mvs_cells_stream[_encoded]) ----
// cells are ordered by (row, col); rows [row0, row0 + rows) of them leave.  abs_ptr[r] = index of the first cell whose row is
// >= row0 + r (r = 0 .. rows): one binary search per row -- rows without cells cost nothing else, and no per-row counts are
// needed from whoever produced the list.
__global__ void k_cells_rowptr(const mvs_cell* __restrict__ cells, long long n, int row0, long long rows, long long* __restrict__ abs_ptr) {
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r > rows) return;
    const long long want = (long long)row0 + r;
    long long lo = 0, hi = n;
    while (lo < hi) {
        const long long mid = (lo + hi) >> 1;
        if ((long long)cells[mid].row < want) lo = mid + 1;
        else hi = mid;
    }
    abs_ptr[r] = lo;
}

// columns and q of the cells [abs_ptr[0], abs_ptr[rows]) into arrays that start at 0, the row index rebased likewise; *wide is
// set when some q does not fit a byte (then the caller runs the pass again with q16)
template <typename Q>
__global__ void k_cells_split(const mvs_cell* __restrict__ cells, const long long* __restrict__ abs_ptr, long long rows,
                              long long* __restrict__ rel_ptr, int32_t* __restrict__ col, Q* __restrict__ q, unsigned int* __restrict__ wide) {
    const long long base = abs_ptr[0], end = abs_ptr[rows];
    const long long tid = (long long)blockIdx.x * blockDim.x + threadIdx.x, step = (long long)gridDim.x * blockDim.x;
    if (rel_ptr)
        for (long long r = tid; r <= rows; r += step) rel_ptr[r] = abs_ptr[r] - base;
    bool odd = false;
    for (long long i = base + tid; i < end; i += step) {
        const mvs_cell c = cells[i];
        col[i - base] = c.col;
        q[i - base] = (Q)c.q;
        odd |= sizeof(Q) == 1 && ((unsigned)c.q > 255u);
    }
    if (odd && wide) atomicOr(wide, 1u);
}
}  // namespace

int launch_encode_sizes(hipStream_t stream, const long long* d_row_ptr, const int32_t* d_col, const void* d_q, int q_bytes,
                        int64_t rows, unsigned long long* d_size, unsigned int* d_jac, unsigned int* d_first_col, EncRow* d_par) {
    if (rows <= 0) return 0;
    if (q_bytes == 2)
        hipLaunchKernelGGL(k_enc_size<uint16_t>, dim3((unsigned)rows), dim3(64), 0, stream, d_row_ptr, d_col, (const uint16_t*)d_q,
                           d_size, d_jac, d_first_col, d_par);
    else
        hipLaunchKernelGGL(k_enc_size<uint8_t>, dim3((unsigned)rows), dim3(64), 0, stream, d_row_ptr, d_col, (const uint8_t*)d_q,
                           d_size, d_jac, d_first_col, d_par);
    return 0;
}

int encode_offsets(hipStream_t stream, unsigned long long* d_size, unsigned long long* d_offset, int64_t rows, void* d_scratch,
                   size_t scratch_bytes, size_t* scratch_needed) {
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, d_size, d_offset, 0ULL, (size_t)rows + 1, rocprim::plus<unsigned long long>(), stream);
    if (e != hipSuccess) return MVS_E_HIP;
    if (scratch_needed) *scratch_needed = need;
    if (d_scratch == nullptr) return 0;
    if (scratch_bytes < need) return MVS_E_CAPACITY;
    e = rocprim::exclusive_scan(d_scratch, need, d_size, d_offset, 0ULL, (size_t)rows + 1, rocprim::plus<unsigned long long>(), stream);
    return e == hipSuccess ? 0 : MVS_E_HIP;
}

int launch_encode_fill(hipStream_t stream, const long long* d_row_ptr, const int32_t* d_col, const void* d_q, int q_bytes,
                       int64_t rows, const unsigned long long* d_offset, const EncRow* d_par, unsigned char* d_out, int stage_words,
                       unsigned long long cap_cells, unsigned long long cap_bytes) {
    if (rows <= 0) return 0;
    const unsigned sw = stage_words < 1 ? 1u : (stage_words > 64 ? 64u : (unsigned)stage_words);
    // stage_words = 64 (the default): common rows (8-bit q, Rice parameter <= 16) take the four-cells-per-lane loop; a lower
    // value (tests) keeps every row on the general loop, whose fallbacks the value exists to exercise; 65+ = general loop, full stage
    const int fast = stage_words == 64 ? 1 : 0;
    if (q_bytes == 2)
        hipLaunchKernelGGL(k_enc_fill<uint16_t>, dim3((unsigned)rows), dim3(64), 0, stream, d_row_ptr, d_col, (const uint16_t*)d_q,
                           d_offset, d_par, d_out, sw, fast, cap_cells, cap_bytes);
    else
        hipLaunchKernelGGL(k_enc_fill<uint8_t>, dim3((unsigned)rows), dim3(64), 0, stream, d_row_ptr, d_col, (const uint8_t*)d_q,
                           d_offset, d_par, d_out, sw, fast, cap_cells, cap_bytes);
    return 0;
}

int launch_cells_rowptr(hipStream_t stream, const mvs_cell* d_cells, int64_t n, int64_t row0, int64_t rows, long long* d_abs_ptr) {
    hipLaunchKernelGGL(k_cells_rowptr, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, stream, d_cells, (long long)n, (int)row0,
                       (long long)rows, d_abs_ptr);
    return 0;
}

int launch_cells_split(hipStream_t stream, const mvs_cell* d_cells, const long long* d_abs_ptr, int64_t rows, int64_t n_upper,
                       long long* d_rel_ptr, int32_t* d_col, void* d_q, int q_bytes, unsigned int* d_wide) {
    const unsigned blocks = (unsigned)std::min<int64_t>(2048, std::max<int64_t>(1, (std::max(n_upper, rows + 1) + 255) / 256));
    if (q_bytes == 2)
        hipLaunchKernelGGL(k_cells_split<uint16_t>, dim3(blocks), dim3(256), 0, stream, d_cells, d_abs_ptr, (long long)rows, d_rel_ptr,
                           d_col, (uint16_t*)d_q, d_wide);
    else
        hipLaunchKernelGGL(k_cells_split<uint8_t>, dim3(blocks), dim3(256), 0, stream, d_cells, d_abs_ptr, (long long)rows, d_rel_ptr,
                           d_col, (uint8_t*)d_q, d_wide);
    return 0;
}

}  // namespace mvs
