// mvs_cells.hip -- kept cells after the comparison kernels: packed lists and the dense byte matrix -> CSR (streamed output),
// a block plan's cells routed into a rank's shard and ordered by (row, col) (row buckets, merge / radix sorts).  The reference
// keeps `all_results` on the host and groups it by row in its writer (src/pairwise_comp_optimized.cpp:974-990, :700-722).
#include "mvs_internal.h"
#include "mvs_encode.h"
#include "mvs_pairwise_dev.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

namespace mvs {

namespace {

// ---- streamed output: sorted packed cells -> CSR (row_ptr, col, q) ----
// row_ptr[r] = index of the first cell whose row is >= r, r in [0, rows]; keys sorted ascending, row = key >> shift.
// One thread per row, a binary search each: no slow case whether rows are empty or hold millions of cells.
__global__ __launch_bounds__(256) void k_packed_row_ptr(const unsigned long long* __restrict__ keys, unsigned long long n,
                                                        int shift, long long rows, long long* __restrict__ row_ptr) {
    const long long r = (long long)blockIdx.x * 256 + threadIdx.x;
    if (r > rows) return;
    unsigned long long lo = 0, hi = n;                       // first index in [0, n] whose row is >= r
    while (lo < hi) {
        const unsigned long long mid = lo + ((hi - lo) >> 1);
        if ((long long)(keys[mid] >> shift) < r) lo = mid + 1;
        else hi = mid;
    }
    row_ptr[r] = (long long)lo;
}

// col / q of every cell; *wide is set when some q does not fit 8 bits (only a norms file that does not belong to the
// vectors does that: a negative Jaccard estimate casts to a 16-bit value) -- the caller then takes
// the 16-bit array instead
__global__ __launch_bounds__(256) void k_packed_unpack(const unsigned long long* __restrict__ keys, unsigned long long n,
                                                       unsigned long long col_mask, int32_t* __restrict__ col,
                                                       uint8_t* __restrict__ q8, uint16_t* __restrict__ q16,
                                                       unsigned int* __restrict__ wide) {
    const unsigned long long i = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    if (i >= n) return;
    const unsigned long long k = keys[i];
    const unsigned q = (unsigned)(k & 0xffffu);
    col[i] = (int32_t)((k >> 16) & col_mask);
    if (q16) q16[i] = (uint16_t)q;
    else {
        q8[i] = (uint8_t)q;
        if (q > 255u) *wide = 1u;                 // benign race: every writer stores the same value
    }
}

// ---- dense byte matrix (see epilogue_exact16) -> CSR ----
// nonzero bytes of 16: one bit per byte
__device__ __forceinline__ unsigned nz_mask16(const v4i w) {
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const unsigned x = (unsigned)w[i];
        m |= ((x & 0xffu) ? 1u : 0u) << (4 * i) | ((x & 0xff00u) ? 2u : 0u) << (4 * i) | ((x & 0xff0000u) ? 4u : 0u) << (4 * i) |
             ((x & 0xff000000u) ? 8u : 0u) << (4 * i);
    }
    return m;
}

// can tile (tr, tc) of the matrix hold a kept cell?  (flag / touch arrays: a few hundred KB, L2 resident)
__device__ __forceinline__ bool tile_active(const DenseActive& A, int tr, int tc) {
    if (A.flags == nullptr) return true;
    const size_t t = (size_t)tr * A.n_tc + tc;
    if (A.flags[t] != 0u || A.touch[t] != 0u) return true;
    const int mr = tc - A.o, mc = tr + A.o;                       // the tile whose mirror image this one is
    return A.sym && mr >= 0 && mr < A.n_tr && mc < A.n_tc && A.flags[(size_t)mr * A.n_tc + mc] != 0u;
}

// The tile columns of one tile row that can hold a kept cell, in ascending order: list[t * ld_list + 0 ..) and count[t]
// for tile row tr0 + t.  The row passes below walk these lists -- at 10 % density a row of 391 tiles has 43 active ones,
// and looking the flags up tile by tile (three dependent loads in front of every 16 bytes of the row, 25 steps per row)
// was what the passes' time went into, not the bytes.
__global__ __launch_bounds__(256) void k_active_tiles(const DenseActive A, int tr0, int n_tc, int* __restrict__ list,
                                                      int* __restrict__ count) {
    __shared__ int part[4];
    __shared__ int run;
    const int tr = tr0 + (int)blockIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (threadIdx.x == 0) run = 0;
    __syncthreads();
    int* out = list + (size_t)blockIdx.x * (size_t)n_tc;
    for (int t0 = 0; t0 < n_tc; t0 += 256) {
        const int t = t0 + (int)threadIdx.x;
        const bool f = t < n_tc && tile_active(A, tr, t);
        const unsigned long long m = __ballot(f);
        if (lane == 0) part[w] = __popcll(m);
        __syncthreads();
        int pos = run + __popcll(m & ((1ULL << lane) - 1ULL));
        for (int i = 0; i < w; ++i) pos += part[i];
        if (f) out[pos] = t;
        __syncthreads();
        if (threadIdx.x == 0) run += part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
    if (threadIdx.x == 0) count[blockIdx.x] = run;
}

// counts[r] = kept cells of row r, ends[r] = {first, last} kept column (when counts[r] > 0): one workgroup per row, 16 bytes
// of an active tile per thread and step
__global__ __launch_bounds__(256) void k_dense_count(const uint8_t* __restrict__ dense, long long ld, long long n_cols,
                                                     long long* __restrict__ counts, int2* __restrict__ ends, long long row_rel0,
                                                     int tr0, int n_tc, const int* __restrict__ list, const int* __restrict__ list_n) {
    __shared__ unsigned part[4];
    __shared__ int part_lo[4], part_hi[4];
    const uint8_t* row = dense + (long long)blockIdx.x * ld;
    const int t = (int)((row_rel0 + blockIdx.x) >> 8) - tr0;
    const int* tl = list + (size_t)t * (size_t)n_tc;
    const int pieces = list_n[t] * 16;
    unsigned c = 0;
    int lo = 0x7fffffff, hi = -1;
    for (int p = (int)threadIdx.x; p < pieces; p += 256) {
        const long long k = (long long)tl[p >> 4] * 256 + (p & 15) * 16;
        if (k >= n_cols) continue;
        unsigned m = nz_mask16(*reinterpret_cast<const v4i*>(row + k));
        if (k + 16 > n_cols) m &= (1u << (n_cols - k)) - 1u;                  // columns beyond the last sample
        if (m) {
            c += (unsigned)__popc(m);
            const int a = (int)k + (__ffs((int)m) - 1), b = (int)k + (31 - __clz((int)m));
            lo = a < lo ? a : lo;
            hi = b > hi ? b : hi;
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        c += __shfl_xor(c, o, 64);
        const int l2 = __shfl_xor(lo, o, 64), h2 = __shfl_xor(hi, o, 64);
        lo = l2 < lo ? l2 : lo;
        hi = h2 > hi ? h2 : hi;
    }
    if ((threadIdx.x & 63) == 0) {
        part[threadIdx.x >> 6] = c;
        part_lo[threadIdx.x >> 6] = lo;
        part_hi[threadIdx.x >> 6] = hi;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        counts[blockIdx.x] = (long long)(part[0] + part[1] + part[2] + part[3]);
        int l = part_lo[0], h = part_hi[0];
        for (int i = 1; i < 4; ++i) {
            l = part_lo[i] < l ? part_lo[i] : l;
            h = part_hi[i] > h ? part_hi[i] : h;
        }
        ends[blockIdx.x] = make_int2(l, h);
    }
}

// what the shard encoder's size pass (k_enc_size, mvs_encode.hip) leaves per row; filled here when SIZES
struct DenseSizes {
    unsigned long long* size;
    unsigned int* jac;
    unsigned int* first_col;
    EncRow* par;
};

// col / q of row r at row_ptr[r]: one workgroup per row, 4 KiB of the row's ACTIVE tiles per step, positions by a block-wide
// prefix sum.  The step's kept cells are gathered in LDS and leave as contiguous runs (thread t writes entries t, t + 256,
// ...): written straight from the lanes -- every lane a short run of its own, a store instruction touching 64 scattered
// words -- the kernel wrote 3.3 x its bytes to memory (WRITE_SIZE 1.03 GB per 6250-row block for 0.31 GB of col + q,
// profiles/r03_c2d_pmc_summary.txt before this change).
// SIZES: the row's record size for the shard codec comes out of the same pass (what k_enc_size computes from the CSR arrays
// this kernel has just written: width of the largest q; Rice parameter from the mean column delta, which is known before the
// pass -- first and last kept column from k_dense_count --; sum of the deltas' quotients).
template <bool SIZES>
__global__ __launch_bounds__(256) void k_dense_fill(const uint8_t* __restrict__ dense, long long ld, long long n_cols,
                                                    const long long* __restrict__ row_ptr, int32_t* __restrict__ col,
                                                    uint8_t* __restrict__ q, long long row_rel0, int tr0, int n_tc,
                                                    const int* __restrict__ list, const int* __restrict__ list_n,
                                                    const int2* __restrict__ ends, const DenseSizes out, long long capacity) {
    // capacity: entries col / q hold (a caller that sized them from the PREVIOUS row block, without reading this block's count
    // back first; stores beyond it are dropped and the caller, who learns the count afterwards, does the block again)
    // A lane scatters a RUN of up to 16 entries starting at its prefix position; in a dense row those positions are 16 apart,
    // i.e. 16 words (columns) or 4 words (q bytes) apart: a 16-way / 4-way bank conflict on every store of the loop
    // (SQ_LDS_BANK_CONFLICT 2.8e7 cycles per launch, round 4).  One pad word per 16 entries (columns: index i lives at
    // i + i / 16, a stride of 17 words; q: byte i at i + 4 * (i / 16), a stride of 5 words -- both odd) spreads the lanes of a
    // store over all 64 banks; the contiguous read-out below stays conflict free.
    __shared__ unsigned wsum[2][4];
    __shared__ int32_t s_col[256 * 16 + 256];
    __shared__ uint8_t s_q[256 * 16 + 4 * 256];
    auto ci = [](unsigned i) { return i + (i >> 4); };
    auto qi = [](unsigned i) { return i + ((i >> 4) << 2); };
    __shared__ unsigned long long red_s[4];
    __shared__ unsigned red_q[4];
    const uint8_t* row = dense + (long long)blockIdx.x * ld;
    long long base = row_ptr[blockIdx.x];
    const long long row_total = row_ptr[blockIdx.x + 1] - base;
    if (row_total == 0) {                                         // block-uniform
        if (SIZES && threadIdx.x == 0) {
            out.size[blockIdx.x] = 0;
            out.jac[blockIdx.x] = 0;
            out.first_col[blockIdx.x] = 0;
            out.par[blockIdx.x] = EncRow{0, 0, 0};
        }
        return;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int t = (int)((row_rel0 + blockIdx.x) >> 8) - tr0;
    const int* tl = list + (size_t)t * (size_t)n_tc;
    const int pieces = list_n[t] * 16;
    unsigned rice_k = 0;
    int2 fl = make_int2(0, 0);
    if (SIZES) {
        fl = ends[blockIdx.x];
        if (row_total > 1) {
            const unsigned long long mean = (unsigned long long)(fl.y - fl.x) / (unsigned long long)(row_total - 1);   // the deltas telescope
            rice_k = mean > 1 ? 63u - (unsigned)__builtin_clzll(mean) : 0u;
        }
    }
    unsigned long long quot = 0;                                  // this thread's share of the sum of (delta >> k)
    unsigned qmax = 0;
    int prev_last = 0;                                            // last kept column of the steps so far
    bool have_prev = false;
    unsigned step = 0;
    for (int p0 = 0; p0 < pieces; p0 += 256, ++step) {
        const int p = p0 + (int)threadIdx.x;
        v4i wv = v4i{0, 0, 0, 0};
        unsigned m = 0;
        long long k = 0;
        if (p < pieces) {
            k = (long long)tl[p >> 4] * 256 + (p & 15) * 16;
            if (k < n_cols) {
                wv = *reinterpret_cast<const v4i*>(row + k);
                m = nz_mask16(wv);
                if (k + 16 > n_cols) m &= (1u << (n_cols - k)) - 1u;   // columns beyond the last sample
            }
        }
        const unsigned mine = (unsigned)__popc(m);
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        unsigned* ws = wsum[step & 1];                            // two sets: no barrier between a step's reads and the next step's writes
        if (lane == 63) ws[w] = incl;
        __syncthreads();                                          // (also: the previous step's write-out has read s_col / s_q)
        unsigned before = 0, total = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            before += i < w ? ws[i] : 0u;
            total += ws[i];
        }
        unsigned at = before + (incl - mine);
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            s_col[ci(at)] = (int32_t)(k + b);
            s_q[qi(at)] = (uint8_t)((unsigned)wv[b >> 2] >> (8 * (b & 3)));
            ++at;
        }
        __syncthreads();
        for (unsigned i = threadIdx.x; i < total; i += 256) {
            const int32_t cv = s_col[ci(i)];
            const unsigned qv = s_q[qi(i)];
            if (base + i < capacity) {
                col[base + i] = cv;
                q[base + i] = (uint8_t)qv;
            }
            if (SIZES) {
                qmax = qv > qmax ? qv : qmax;
                if (i > 0) quot += (unsigned long long)(unsigned)(cv - s_col[ci(i - 1)]) >> rice_k;
                else if (have_prev) quot += (unsigned long long)(unsigned)(cv - prev_last) >> rice_k;
            }
        }
        if (SIZES && total) {
            prev_last = s_col[ci(total - 1)];
            have_prev = true;
        }
        base += total;
    }
    if (SIZES) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            quot += (unsigned long long)__shfl_xor((long long)quot, o, 64);
            const unsigned other = (unsigned)__shfl_xor((int)qmax, o, 64);
            qmax = other > qmax ? other : qmax;
        }
        __syncthreads();
        if (lane == 0) {
            red_s[w] = quot;
            red_q[w] = qmax;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long s = red_s[0] + red_s[1] + red_s[2] + red_s[3];
            unsigned mx = red_q[0];
            for (int i = 1; i < 4; ++i) mx = red_q[i] > mx ? red_q[i] : mx;
            const unsigned long long n = (unsigned long long)row_total;
            const unsigned wq = mx ? 32u - (unsigned)__clz((int)mx) : 1u;         // compact_vector::build: width of the largest, at least 1
            const unsigned long long jac_bytes = 8 * (3 + (n * wq + 63) / 64);
            unsigned long long total_bytes = jac_bytes, high = 0;
            if (n > 1) {
                const unsigned long long nr = n - 1;
                high = nr + s;
                total_bytes += 8 * (5 + (high + 63) / 64 + (nr + 63) / 64) + (rice_k ? 8 * (3 + (nr * rice_k + 63) / 64) : 0);
            }
            out.size[blockIdx.x] = total_bytes;
            out.jac[blockIdx.x] = (unsigned)jac_bytes;
            out.first_col[blockIdx.x] = (unsigned)fl.x;
            out.par[blockIdx.x] = EncRow{high, wq, n > 1 ? rice_k : 0u};
        }
    }
}

// ---- the re-check's kept cells (packed words) -> dense byte matrix ----
// pass 1: mark the tile of every cell; a tile marked for the first time goes on the list of tiles to clear
__global__ __launch_bounds__(256) void k_packed_touch(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ n_ptr,
                                                      int shift, unsigned long long col_mask, unsigned int* __restrict__ touch, int n_tc,
                                                      int* __restrict__ new_list, unsigned int* __restrict__ new_count) {
    const unsigned long long n = *n_ptr;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long key = keys[i];
        const long long row = (long long)(key >> shift), col = (long long)((key >> 16) & col_mask);
        const size_t t = (size_t)(row >> 8) * n_tc + (size_t)(col >> 8);
        if (*reinterpret_cast<volatile unsigned int*>(touch + t) != 0u) continue;
        if (atomicExch(touch + t, 1u) == 0u) new_list[atomicAdd(new_count, 1u)] = (int)t;
    }
}

// pass 2: clear the newly touched tiles (256 rows x 256 bytes each, inside the matrix)
__global__ __launch_bounds__(256) void k_clear_tiles(const int* __restrict__ list, const unsigned int* __restrict__ count, uint8_t* __restrict__ dense,
                                                     long long ld, long long matrix_rows, int n_tc) {
    const unsigned n = *count;
    for (unsigned e = blockIdx.x; e < n; e += gridDim.x) {
        const int t = list[e];
        const long long r0 = (long long)(t / n_tc) * 256, c0 = (long long)(t % n_tc) * 256;
        for (int x = threadIdx.x; x < 256 * 16; x += 256) {
            const long long r = r0 + (x >> 4), cc = c0 + (x & 15) * 16;
            if (r < matrix_rows && cc < ld) *reinterpret_cast<v4i*>(dense + r * ld + cc) = v4i{0, 0, 0, 0};
        }
    }
}

// pass 3: the bytes
__global__ __launch_bounds__(256) void k_packed_scatter(const unsigned long long* __restrict__ keys, const unsigned long long* __restrict__ n_ptr,
                                                        int shift, unsigned long long col_mask, uint8_t* __restrict__ dense, long long ld,
                                                        unsigned int* __restrict__ odd) {
    const unsigned long long n = *n_ptr;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * 256) {
        const unsigned long long key = keys[i];
        const long long row = (long long)(key >> shift), col = (long long)((key >> 16) & col_mask);
        const unsigned q = (unsigned)(key & 0xffffULL);
        if (q == 0u || q > 255u) *odd = 1u;
        dense[row * ld + col] = (uint8_t)q;
    }
}

// ---- kept cells of a block plan -> the rank's shard (mvs_cells_route / mvs_cells_collect) ----
// header of a send buffer: [0] foreign cells appended (may exceed the capacity), [1] status, [2] max |v|, [3] raw cells,
// [4] raw capacity; 64 bytes, then the cells
struct RouteArgs {
    const mvs_cell* raw;
    const unsigned long long* n_raw;
    unsigned long long raw_capacity;
    long long block_pad, block_rows, n_total;   // storage row s -> sample (s / block_pad) * block_rows + s % block_pad
    int own_begin, own_end;
    mvs_cell* own;
    unsigned long long own_capacity;
    unsigned long long* own_count;               // the shard's state block: [0] cells of own rows, [1] low word: most cells in one
                                                 // row (k_rows_max), then one uint32 per own row: its cells (+ one zero entry)
    unsigned long long* send;                    // header (8 words) + cells, or NULL
    unsigned long long foreign_capacity;
    long long status, max_abs;
};

// room for `mine` cells per lane behind *counter with ONE atomic per wave (wave_reserve above), for the route / collect kernels;
// row_cells (the shard's own cells only): per-row counts for the row-bucket sort, indexed by row - row0
__device__ __forceinline__ void append_cells(mvs_cell* out, unsigned long long cap, unsigned long long* counter, const mvs_cell* c,
                                             unsigned want_mask, int lane, unsigned* row_cells = nullptr, int row0 = 0) {
    const unsigned mine = (unsigned)__popc(want_mask);
    if (__ballot(mine != 0) == 0ULL) return;
    unsigned long long slot = wave_reserve(counter, mine, lane);
#pragma unroll
    for (int k = 0; k < 8; ++k)
        if (want_mask & (1u << k)) {
            if (slot < cap) out[slot] = c[k];
            if (row_cells) atomicAdd(row_cells + (c[k].row - row0), 1u);
            ++slot;
        }
}

// a wave takes 512 consecutive cells per round (8 per lane, each load instruction 1 KiB contiguous) and reserves room for all
// it keeps with one atomic: 1.6 M cells are 3 200 atomics on the counter instead of 25 000 (one per 64 cells: 0.3 ms, the
// counter's line going back and forth)
__global__ __launch_bounds__(256) void k_cells_route(const RouteArgs r) {
    const int lane = threadIdx.x & 63;
    const unsigned long long total = *r.n_raw;
    const unsigned long long n = total < r.raw_capacity ? total : r.raw_capacity;
    if (blockIdx.x == 0 && threadIdx.x == 0 && r.send) {
        r.send[1] = (unsigned long long)r.status;
        r.send[2] = (unsigned long long)r.max_abs;
        r.send[3] = total;
        r.send[4] = r.raw_capacity;
    }
    mvs_cell* foreign = r.send ? reinterpret_cast<mvs_cell*>(r.send + 8) : nullptr;
    const unsigned long long waves = (unsigned long long)gridDim.x * 4;
    for (unsigned long long base = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 512; base < n; base += waves * 512) {
        mvs_cell c[8];
        unsigned mine = 0, other = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long i = base + (unsigned long long)k * 64 + lane;
            c[k] = mvs_cell{0, 0, 0, 0};
            if (i < n) {
                c[k] = r.raw[i];
                const long long br = c[k].row / r.block_pad, orow = c[k].row - br * r.block_pad;
                const long long bc = c[k].col / r.block_pad, ocol = c[k].col - bc * r.block_pad;
                const long long row = br * r.block_rows + orow, col = bc * r.block_rows + ocol;
                const bool valid = orow < r.block_rows && ocol < r.block_rows && row < r.n_total && col < r.n_total;
                c[k].row = (int32_t)row;
                c[k].col = (int32_t)col;
                const bool own = valid && row >= r.own_begin && row < r.own_end;
                mine |= own ? 1u << k : 0u;
                other |= (valid && !own) ? 1u << k : 0u;
            }
        }
        append_cells(r.own, r.own_capacity, r.own_count, c, mine, lane, reinterpret_cast<unsigned*>(r.own_count + 2), r.own_begin);
        if (r.send) append_cells(foreign, r.foreign_capacity, r.send, c, other, lane);
    }
}

// recv: `world` send buffers of hdr + capacity cells each; blockIdx.y = peer
__global__ __launch_bounds__(256) void k_cells_collect(const unsigned long long* __restrict__ recv, int rank, unsigned long long capacity,
                                                       int own_begin, int own_end, mvs_cell* __restrict__ own, unsigned long long own_capacity,
                                                       unsigned long long* __restrict__ own_count) {
    const int peer = blockIdx.y;
    if (peer == rank) return;
    const int lane = threadIdx.x & 63;
    const unsigned long long* buf = recv + (size_t)peer * (8 + capacity * 2);
    const unsigned long long total = buf[0];
    const unsigned long long n = total < capacity ? total : capacity;
    const mvs_cell* cells = reinterpret_cast<const mvs_cell*>(buf + 8);
    const unsigned long long waves = (unsigned long long)gridDim.x * 4;
    for (unsigned long long base = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 512; base < n; base += waves * 512) {
        mvs_cell c[8];
        unsigned mine = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long i = base + (unsigned long long)k * 64 + lane;
            c[k] = mvs_cell{0, 0, 0, 0};
            if (i < n) {
                c[k] = cells[i];
                mine |= (c[k].row >= own_begin && c[k].row < own_end) ? 1u << k : 0u;
            }
        }
        append_cells(own, own_capacity, own_count, c, mine, lane, reinterpret_cast<unsigned*>(own_count + 2), own_begin);
    }
}

// ---- the shard's cells in (row, col) order by ROW BUCKETS (mvs_cells_sort_rows): the route / collect kernels have counted the
// cells of every row; an exclusive scan of the counts gives every row its segment, the cells are scattered into their rows'
// segments (order inside a row: arbitrary), and one wave per row orders its <= 64 cells by column with a bitonic network over
// the lanes.  Four short kernels instead of a general sort of 16-byte records (1.6 M cells: 0.28 ms; 2 x 10^5: 0.15 ms) --
// a shard has ~16 cells per row.  A shard with a row of more than 64 cells takes the general sort (the caller knows the
// largest row from the report it reads anyway).
__global__ __launch_bounds__(1024) void k_rows_max(unsigned long long* __restrict__ state, int rows) {
    __shared__ unsigned part[16];
    const unsigned* cnt = reinterpret_cast<const unsigned*>(state + 2);
    unsigned m = 0;
    for (int i = threadIdx.x; i < rows; i += 1024) m = cnt[i] > m ? cnt[i] : m;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned x = (unsigned)__shfl_xor((int)m, o, 64);
        m = x > m ? x : m;
    }
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = part[w] > m ? part[w] : m;
        state[1] = m;
    }
}

// The scan of the row counts, the copy that becomes the scatter's cursors and the widest row in ONE workgroup (rows + 1 <=
// kRowsScanMax: a rank's shard of a split; 12 500 rows of an 8-way split took rocprim's two kernels + a device copy + k_rows_max
// 25 us of a 1.7 ms step): thread t owns a contiguous run of <= 16 counts, all loaded before the first is used, the 1024 sums
// are scanned over the lanes and through the LDS.  counts holds rows + 1 entries (the last one zero), row_ptr / cursor likewise.
constexpr int kRowsScanPer = 16, kRowsScanMax = kRowsScanPer * 1024;
__global__ __launch_bounds__(1024) void k_rows_scan(unsigned long long* __restrict__ state, int rows, unsigned* __restrict__ row_ptr,
                                                    unsigned* __restrict__ cursor) {
    __shared__ unsigned wsum[16], wmax[16];
    const unsigned* cnt = reinterpret_cast<const unsigned*>(state + 2);
    const int n = rows + 1, t = threadIdx.x, lane = t & 63, w = t >> 6;
    const int per = (n + 1023) / 1024;                     // <= kRowsScanPer
    const int b = t * per;
    unsigned v[kRowsScanPer];
#pragma unroll
    for (int k = 0; k < kRowsScanPer; ++k) v[k] = (k < per && b + k < n) ? cnt[b + k] : 0u;
    unsigned sum = 0, m = 0;
#pragma unroll
    for (int k = 0; k < kRowsScanPer; ++k) {
        sum += v[k];
        m = v[k] > m ? v[k] : m;
    }
    unsigned incl = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned x = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += x;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned x = (unsigned)__shfl_xor((int)m, o, 64);
        m = x > m ? x : m;
    }
    if (lane == 63) wsum[w] = incl;
    if (lane == 0) wmax[w] = m;
    __syncthreads();
    unsigned base = incl - sum;
    for (int k = 0; k < w; ++k) base += wsum[k];
#pragma unroll
    for (int k = 0; k < kRowsScanPer; ++k) {
        if (k < per && b + k < n) {
            row_ptr[b + k] = base;
            cursor[b + k] = base;
        }
        base += v[k];
    }
    if (t == 0) {
        for (int k = 1; k < 16; ++k) m = wmax[k] > m ? wmax[k] : m;
        state[1] = m;
    }
}

// up to 8 device ranges cleared by one launch (a plan's counters, candidate headers, tile flags, row marks: six memsets became
// nine fill kernels of 5 us each in front of every plan); bytes are multiples of 4
struct ZeroRanges {
    unsigned* p[8];
    unsigned long long words[8];
    int n;
};
__global__ __launch_bounds__(256) void k_zero_ranges(const ZeroRanges z) {
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (int r = 0; r < z.n; ++r) {
        unsigned* __restrict__ p = z.p[r];
        const unsigned long long nw = z.words[r];
        if ((reinterpret_cast<unsigned long long>(p) & 15) == 0) {           // whole 16-byte stores, then the tail
            const unsigned long long n4 = nw >> 2;
            uint4* p4 = reinterpret_cast<uint4*>(p);
            for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) p4[i] = uint4{0, 0, 0, 0};
            for (unsigned long long i = (n4 << 2) + (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nw; i += stride) p[i] = 0u;
        } else {
            for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < nw; i += stride) p[i] = 0u;
        }
    }
}

// d_count != NULL: the number of cells is read there (a sort queued in front of the read-back that would have told the host) and
// bounded by in_cap; positions beyond out_cap are not written (the row counts include cells a full buffer dropped)
__global__ __launch_bounds__(256) void k_rows_scatter(const mvs_cell* __restrict__ in, unsigned long long n,
                                                      const unsigned long long* __restrict__ d_count, unsigned long long in_cap, int row0,
                                                      unsigned* __restrict__ cursor, mvs_cell* __restrict__ out, unsigned long long out_cap) {
    if (d_count) {
        n = *d_count;
        n = n < in_cap ? n : in_cap;
    }
    const unsigned long long stride = (unsigned long long)gridDim.x * 256;
    for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) {
        const mvs_cell c = in[i];
        const unsigned pos = atomicAdd(cursor + (c.row - row0), 1u);
        if (pos < out_cap) out[pos] = c;
    }
}

// bitonic network over W lanes (W = 16: four rows per wave, W = 64: one), ascending by column; lanes without a cell hold INT_MAX
template <int W>
__device__ __forceinline__ void bitonic_by_col(mvs_cell& c, int lane) {
#pragma unroll
    for (int k = 2; k <= W; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1) {
            mvs_cell o;
            o.row = __shfl_xor(c.row, j, 64);
            o.col = __shfl_xor(c.col, j, 64);
            o.dot = __shfl_xor(c.dot, j, 64);
            o.q = __shfl_xor(c.q, j, 64);
            const bool up = (lane & k) == 0 || k == W;        // this k-block sorts ascending (the last merge always does)
            const bool low = (lane & j) == 0;                 // the lower lane of a pair keeps the smaller key when ascending
            const bool take_min = up == low;
            if (take_min ? o.col < c.col : o.col > c.col) c = o;
        }
}

__global__ __launch_bounds__(256) void k_rows_sort(mvs_cell* __restrict__ cells, const unsigned* __restrict__ row_ptr, int rows,
                                                   unsigned long long out_cap) {
    const int lane = threadIdx.x & 63;
    const int waves = gridDim.x * 4;
    // a wave takes four consecutive rows: when none of them holds more than 16 cells (the usual shard: clusters of 16) each
    // quarter of the wave sorts one row, 10 exchange steps instead of 21 on a quarter of the lanes; otherwise row by row
    for (int r0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4; r0 < rows; r0 += waves * 4) {
        const int rq = r0 + (lane >> 4);
        const unsigned bq = rq < rows ? row_ptr[rq] : 0u;
        const unsigned cq = rq < rows && row_ptr[rq + 1] <= out_cap ? row_ptr[rq + 1] - bq : 0u;      // (rows beyond the buffer: not there)
        if (__ballot(cq > 16u) == 0ULL) {
            mvs_cell c{0, 0x7fffffff, 0, 0};
            const unsigned l16 = (unsigned)lane & 15u;
            if (l16 < cq) c = cells[bq + l16];
            if (__ballot(cq > 1u) != 0ULL) bitonic_by_col<16>(c, lane);
            if (l16 < cq && cq > 1u) cells[bq + l16] = c;
            continue;
        }
        for (int r = r0; r < r0 + 4 && r < rows; ++r) {
            const unsigned b = row_ptr[r], cnt = row_ptr[r + 1] - b;
            if (cnt < 2 || row_ptr[r + 1] > out_cap) continue;
            mvs_cell c{0, 0x7fffffff, 0, 0};
            if ((unsigned)lane < cnt) c = cells[b + lane];
            bitonic_by_col<64>(c, lane);
            if ((unsigned)lane < cnt) cells[b + lane] = c;
        }
    }
}

struct CellLess {
    __host__ __device__ bool operator()(const mvs_cell& x, const mvs_cell& y) const {
        return x.row < y.row || (x.row == y.row && x.col < y.col);
    }
};


}  // namespace

int launch_cells_route(hipStream_t stream, const mvs_cell* d_raw, const unsigned long long* d_n_raw, unsigned long long raw_capacity,
                       long long block_pad, long long block_rows, long long n_total, int own_begin, int own_end, mvs_cell* d_own,
                       unsigned long long own_capacity, unsigned long long* d_own_count, unsigned long long* d_send,
                       unsigned long long foreign_capacity, long long status, long long max_abs) {
    RouteArgs r{d_raw, d_n_raw, raw_capacity, block_pad, block_rows, n_total, own_begin, own_end, d_own, own_capacity, d_own_count,
                d_send, foreign_capacity, status, max_abs};
    const unsigned long long blocks = std::min<unsigned long long>(1024ULL, std::max<unsigned long long>(1ULL, (raw_capacity + 255) / 256));
    hipLaunchKernelGGL(k_cells_route, dim3((unsigned)blocks), dim3(256), 0, stream, r);
    return 0;
}

int launch_cells_collect(hipStream_t stream, const unsigned long long* d_recv, int world, int rank, unsigned long long capacity,
                         int own_begin, int own_end, mvs_cell* d_own, unsigned long long own_capacity, unsigned long long* d_own_count) {
    if (world <= 1) return 0;
    const unsigned long long blocks = std::min<unsigned long long>(256ULL, std::max<unsigned long long>(1ULL, (capacity + 255) / 256));
    hipLaunchKernelGGL(k_cells_collect, dim3((unsigned)blocks, (unsigned)world), dim3(256), 0, stream, d_recv, rank, capacity, own_begin,
                       own_end, d_own, own_capacity, d_own_count);
    return 0;
}

// the row-bucket sort (see k_rows_scatter): d_state = the shard's state block the route / collect kernels filled; d_scratch holds
// 2 x (rows + 1) uint32 (row_ptr, cursor) + the scan's own scratch
int launch_rows_max(hipStream_t stream, unsigned long long* d_state, int rows) {
    hipLaunchKernelGGL(k_rows_max, dim3(1), dim3(1024), 0, stream, d_state, rows);
    return 0;
}

int launch_zero_ranges(hipStream_t stream, void* const* ptrs, const size_t* bytes, int n) {
    ZeroRanges z{};
    unsigned long long total = 0;
    for (int k = 0; k < n; ++k) {
        if (!ptrs[k] || bytes[k] == 0) continue;
        if (z.n == 8 || (bytes[k] & 3) != 0 || (reinterpret_cast<unsigned long long>(ptrs[k]) & 3) != 0) return MVS_E_INVALID;
        z.p[z.n] = static_cast<unsigned*>(ptrs[k]);
        z.words[z.n] = bytes[k] / 4;
        total += z.words[z.n];
        ++z.n;
    }
    if (z.n == 0) return 0;
    const unsigned blocks = (unsigned)std::min<unsigned long long>(1024, std::max<unsigned long long>(1, (total / 4 + 255) / 256));
    hipLaunchKernelGGL(k_zero_ranges, dim3(blocks), dim3(256), 0, stream, z);
    return 0;
}

int sort_cells_rows(hipStream_t stream, const mvs_cell* d_in, mvs_cell* d_out, int64_t n, int row0, int rows,
                    const unsigned long long* d_state, void* d_scratch, size_t scratch_bytes, size_t* scratch_needed,
                    int64_t in_cap, int64_t out_cap) {
    const unsigned* counts = reinterpret_cast<const unsigned*>(d_state + 2);
    const size_t tab = ((size_t)rows + 1) * sizeof(unsigned);
    const size_t tab_al = (tab + 255) / 256 * 256;
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, counts, (unsigned*)nullptr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), stream);
    if (e != hipSuccess) return MVS_E_HIP;
    if (scratch_needed) *scratch_needed = 2 * tab_al + need;
    if (d_scratch == nullptr) return 0;
    if (scratch_bytes < 2 * tab_al + need) return MVS_E_CAPACITY;
    unsigned* row_ptr = reinterpret_cast<unsigned*>(d_scratch);
    unsigned* cursor = reinterpret_cast<unsigned*>(static_cast<char*>(d_scratch) + tab_al);
    void* scan_tmp = static_cast<char*>(d_scratch) + 2 * tab_al;
    if (rows < kRowsScanMax) {      // scan, cursors and the widest row (state[1], what k_rows_max would write) in one launch
        hipLaunchKernelGGL(k_rows_scan, dim3(1), dim3(1024), 0, stream, const_cast<unsigned long long*>(d_state), rows, row_ptr, cursor);
    } else {
        e = rocprim::exclusive_scan(scan_tmp, need, counts, row_ptr, 0u, (size_t)rows + 1, rocprim::plus<unsigned>(), stream);
        if (e != hipSuccess) return MVS_E_HIP;
        if (hipMemcpyAsync(cursor, row_ptr, tab, hipMemcpyDeviceToDevice, stream) != hipSuccess) return MVS_E_HIP;
        hipLaunchKernelGGL(k_rows_max, dim3(1), dim3(1024), 0, stream, const_cast<unsigned long long*>(d_state), rows);
    }
    // in_cap >= 0: the count is d_state[0] on the device, at most in_cap cells are there; the grid is sized for the buffer
    const bool ahead = in_cap >= 0;
    const int64_t size_for = ahead ? in_cap : n;
    const unsigned blocks = (unsigned)std::min<int64_t>(ahead ? 1024 : 2048, std::max<int64_t>(1, (size_for + 255) / 256));
    hipLaunchKernelGGL(k_rows_scatter, dim3(blocks), dim3(256), 0, stream, d_in, (unsigned long long)(ahead ? 0 : n), ahead ? d_state : nullptr,
                       (unsigned long long)(ahead ? in_cap : 0), row0, cursor, d_out, out_cap >= 0 ? (unsigned long long)out_cap : ~0ULL);
    // (a shard without rows -- a rank behind the last sample -- still gets a valid grid)
    hipLaunchKernelGGL(k_rows_sort, dim3((unsigned)std::max(1, std::min(4096, (rows + 15) / 16))), dim3(256), 0, stream, d_out, row_ptr, rows,
                       out_cap >= 0 ? (unsigned long long)out_cap : ~0ULL);
    return 0;
}

int sort_packed(hipStream_t stream, unsigned long long* d_in, unsigned long long* d_out, int64_t n, int begin_bit, int end_bit,
                void* d_scratch, size_t scratch_bytes, size_t* scratch_needed) {
    size_t need = 0;
    hipError_t e = rocprim::radix_sort_keys(nullptr, need, d_in, d_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, stream);
    if (e != hipSuccess) return MVS_E_HIP;
    if (scratch_needed) *scratch_needed = need;
    if (d_scratch == nullptr) return 0;
    if (scratch_bytes < need) return MVS_E_CAPACITY;
    e = rocprim::radix_sort_keys(d_scratch, need, d_in, d_out, (size_t)n, (unsigned)begin_bit, (unsigned)end_bit, stream);
    return e == hipSuccess ? 0 : MVS_E_HIP;
}

int launch_packed_csr(hipStream_t stream, const unsigned long long* d_keys, int64_t n, int shift, int64_t rows,
                      unsigned long long col_mask, long long* d_row_ptr, int32_t* d_col, uint8_t* d_q8, uint16_t* d_q16,
                      unsigned int* d_wide) {
    if (d_row_ptr)
        hipLaunchKernelGGL(k_packed_row_ptr, dim3((unsigned)((rows + 1 + 255) / 256)), dim3(256), 0, stream, d_keys,
                           (unsigned long long)n, shift, (long long)rows, d_row_ptr);
    if (n > 0 && d_col)
        hipLaunchKernelGGL(k_packed_unpack, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, d_keys, (unsigned long long)n,
                           col_mask, d_col, d_q8, d_q16, d_wide);
    return 0;
}

// rows [0, rows) of a dense byte matrix -> the active tiles of their tile rows (d_list: n_trows x n_tc ints, d_list_n:
// n_trows), counts + first / last kept column, row_ptr (exclusive scan, row_ptr[rows] = total), then col / q
void dense_tile_rows(const DenseActive& active, int64_t rows, int64_t n_cols, int* tr0, int* n_trows, int* n_tc) {
    *tr0 = (int)(active.row_rel0 >> 8);
    *n_trows = rows > 0 ? (int)((active.row_rel0 + rows - 1) >> 8) - *tr0 + 1 : 0;
    *n_tc = (int)((n_cols + 255) / 256);
}

int launch_dense_count(hipStream_t stream, const uint8_t* d_dense, int64_t ld, int64_t n_cols, int64_t rows, long long* d_counts,
                       int2* d_ends, const DenseActive& active, int* d_list, int* d_list_n) {
    if (rows <= 0) return 0;
    int tr0, n_trows, n_tc;
    dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    hipLaunchKernelGGL(k_active_tiles, dim3((unsigned)n_trows), dim3(256), 0, stream, active, tr0, n_tc, d_list, d_list_n);
    hipLaunchKernelGGL(k_dense_count, dim3((unsigned)rows), dim3(256), 0, stream, d_dense, (long long)ld, (long long)n_cols, d_counts,
                       d_ends, (long long)active.row_rel0, tr0, n_tc, (const int*)d_list, (const int*)d_list_n);
    return 0;
}

int launch_packed_to_dense(hipStream_t stream, const unsigned long long* d_keys, const unsigned long long* d_n, int shift,
                           unsigned long long col_mask, uint8_t* d_dense, int64_t ld, int64_t matrix_rows, unsigned int* d_touch,
                           int n_tc, int* d_new, unsigned int* d_new_count, unsigned int* d_odd) {
    hipLaunchKernelGGL(k_packed_touch, dim3(512), dim3(256), 0, stream, d_keys, d_n, shift, col_mask, d_touch, n_tc, d_new, d_new_count);
    hipLaunchKernelGGL(k_clear_tiles, dim3(2048), dim3(256), 0, stream, (const int*)d_new, (const unsigned int*)d_new_count, d_dense,
                       (long long)ld, (long long)matrix_rows, n_tc);
    hipLaunchKernelGGL(k_packed_scatter, dim3(512), dim3(256), 0, stream, d_keys, d_n, shift, col_mask, d_dense, (long long)ld, d_odd);
    return 0;
}

int dense_row_ptr(hipStream_t stream, long long* d_counts, long long* d_row_ptr, int64_t rows, void* d_scratch, size_t scratch_bytes,
                  size_t* scratch_needed) {
    // counts has rows + 1 entries, the last one 0: the exclusive scan of all of them ends with the total
    size_t need = 0;
    hipError_t e = rocprim::exclusive_scan(nullptr, need, d_counts, d_row_ptr, 0LL, (size_t)rows + 1, rocprim::plus<long long>(), stream);
    if (e != hipSuccess) return MVS_E_HIP;
    if (scratch_needed) *scratch_needed = need;
    if (d_scratch == nullptr) return 0;
    if (scratch_bytes < need) return MVS_E_CAPACITY;
    e = rocprim::exclusive_scan(d_scratch, need, d_counts, d_row_ptr, 0LL, (size_t)rows + 1, rocprim::plus<long long>(), stream);
    return e == hipSuccess ? 0 : MVS_E_HIP;
}

// d_size non-NULL: the shard encoder's per-row sizes as well (size / jac / first_col / par of launch_encode_sizes)
int launch_dense_fill(hipStream_t stream, const uint8_t* d_dense, int64_t ld, int64_t n_cols, int64_t rows, const long long* d_row_ptr,
                      int32_t* d_col, uint8_t* d_q, const DenseActive& active, const int* d_list, const int* d_list_n,
                      const int2* d_ends, unsigned long long* d_size, unsigned int* d_jac, unsigned int* d_first_col, EncRow* d_par,
                      long long capacity) {
    if (rows <= 0) return 0;
    int tr0, n_trows, n_tc;
    dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    const DenseSizes out{d_size, d_jac, d_first_col, d_par};
    if (d_size)
        hipLaunchKernelGGL(k_dense_fill<true>, dim3((unsigned)rows), dim3(256), 0, stream, d_dense, (long long)ld, (long long)n_cols,
                           d_row_ptr, d_col, d_q, (long long)active.row_rel0, tr0, n_tc, d_list, d_list_n, d_ends, out, capacity);
    else
        hipLaunchKernelGGL(k_dense_fill<false>, dim3((unsigned)rows), dim3(256), 0, stream, d_dense, (long long)ld, (long long)n_cols,
                           d_row_ptr, d_col, d_q, (long long)active.row_rel0, tr0, n_tc, d_list, d_list_n, d_ends, out, capacity);
    return 0;
}

// (row, col) as one 64-bit radix key, row most significant
struct CellKey {
    __host__ __device__ ::rocprim::tuple<int32_t&, int32_t&> operator()(mvs_cell& c) const {
        return ::rocprim::tuple<int32_t&, int32_t&>{c.row, c.col};
    }
};

int sort_cells(hipStream_t stream, mvs_cell* d_cells, mvs_cell* d_tmp, int64_t n, void* d_scratch,
               size_t scratch_bytes, size_t* scratch_needed, const Options& opt) {
    size_t need = 0;
    // radix sort on the 64-bit (row, col) key for long lists (2.5e6 cells: 0.4 ms faster than the merge sort),
    // merge sort for short ones (1.6e5 cells: 0.03 ms faster); opt.sort = 1 (merge) / 2 (radix) forces one
    if (opt.sort == 2 || (opt.sort == 0 && n >= (1 << 19))) {
        hipError_t e = rocprim::radix_sort_keys(nullptr, need, d_cells, d_tmp, (size_t)n, CellKey(), 0u, 64u, stream);
        if (e != hipSuccess) return MVS_E_HIP;
        if (scratch_needed) *scratch_needed = need;
        if (d_scratch == nullptr) return 0;
        if (scratch_bytes < need) return MVS_E_CAPACITY;
        e = rocprim::radix_sort_keys(d_scratch, need, d_cells, d_tmp, (size_t)n, CellKey(), 0u, 64u, stream);
        return e == hipSuccess ? 0 : MVS_E_HIP;
    }
    hipError_t e = rocprim::merge_sort(nullptr, need, d_cells, d_tmp, (size_t)n, CellLess(), stream);
    if (e != hipSuccess) return MVS_E_HIP;
    if (scratch_needed) *scratch_needed = need;
    if (d_scratch == nullptr) return 0;
    if (scratch_bytes < need) return MVS_E_CAPACITY;
    e = rocprim::merge_sort(d_scratch, need, d_cells, d_tmp, (size_t)n, CellLess(), stream);
    return e == hipSuccess ? 0 : MVS_E_HIP;
}

}  // namespace mvs
