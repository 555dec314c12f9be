// mvs_pairwise_dev.h -- device helpers shared by the kernel units of the comparison (mvs_pairwise.hip, mvs_cells.hip):
// tile map, keep test / quantiser, cell stores, wave-level reservations.  Not installed.
#ifndef MVS_PAIRWISE_DEV_H
#define MVS_PAIRWISE_DEV_H

#include "mvs_internal.h"

namespace mvs {
namespace {

using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;


__host__ __device__ constexpr int num_acc_sets(int L) { return L == 1 ? 1 : (L == 2 ? 3 : 4); }

struct TileCoord {
    int tr, tc;
    bool valid;
};

// (blockIdx.x, blockIdx.y) -> tile.  The tile grid is cut into 16x16-tile super-patches: blockIdx.y is the
// patch row, blockIdx.x / 256 the patch column; inside a patch the XCD label (blockIdx.x % 8 -- gridDim.x is
// a multiple of 256, so this is also the linear workgroup id % 8), rotated by the patch row, picks a 4-row x 8-col sub-patch and
// (blockIdx.x / 8) % 32 walks it.  Placement only affects speed.  (A 2-D grid because a dispatch holds at
// most 2^32 work-items per dimension: one dimension would cap the matrix at ~370k samples.)
__device__ __forceinline__ TileCoord map_tile(unsigned b, unsigned patch_row, int n_tr, int n_tc, int map_mode = 0) {
    // The sub-patch an XCD takes rotates with the patch row.  With a fixed assignment the symmetric schedule is
    // lopsided: in a patch on the diagonal the sub-patches hold 32, 26, 10 or 0 tiles above the diagonal, in the
    // last patch column only the left sub-patches exist -- measured at 100k samples (per-workgroup time stamps):
    // 9072 .. 10150 tiles per XCD, the fullest XCD finishing 4 % after the average one.
    const unsigned x = (b + patch_row) & 7u;
    const unsigned q = b >> 3;
    const unsigned ql = q & 31u;
    const int spr = (int)patch_row, spc = (int)(q >> 5);
    TileCoord t;
    if (map_mode == 1) {          // 8-row x 4-col sub-patches
        t.tr = spr * 16 + (int)(x >> 2) * 8 + (int)(ql >> 2);
        t.tc = spc * 16 + (int)(x & 3u) * 4 + (int)(ql & 3u);
    } else if (map_mode == 2) {   // 2-row x 16-col sub-patches
        t.tr = spr * 16 + (int)x * 2 + (int)(ql >> 4);
        t.tc = spc * 16 + (int)(ql & 15u);
    } else if (map_mode == 3) {   // 16-row x 2-col sub-patches: grids of ONE patch row (a few query rows against a whole
        t.tr = spr * 16 + (int)(ql & 15u);          // database) -- with sub-patches that split the rows, a grid of <= 4 tile
        t.tc = spc * 16 + (int)x * 2 + (int)(ql >> 4);   // rows keeps 2 of the 8 XCDs busy (launchers: skinny_map, < 16 tile rows)
    } else {
        t.tr = spr * 16 + (int)(x >> 1) * 4 + (int)(ql >> 3);
        t.tc = spc * 16 + (int)(x & 1u) * 8 + (int)(ql & 7u);
    }
    t.valid = t.tr < n_tr && t.tc < n_tc;
    return t;
}

// exact per-cell decision + quantisation, identical operation order to the reference
__device__ __forceinline__ bool keep_cell(int32_t P, int d, double n2r, double n2c, int keep_mode, double coeff) {
    const double threshold = coeff * (n2r + n2c);                      // :139 (coeff = 0.05)
    if (keep_mode == MVS_KEEP_INT32) {
        const long long q = (long long)P / (long long)d;               // :140-141 truncating
        return (double)q > threshold;
    }
    return (double)P / (double)d > threshold;                          // _16bits.cpp:218
}

__device__ __forceinline__ int32_t quantize_cell(int32_t P, int d, double n2r, double n2c) {
    const double inter = (double)P / (double)d;                        // :661
    double jac = inter / (n2r + n2c - inter);                          // :662
    if (jac > 1) jac = 1;                                              // :663
    const double r = round(jac * 255.0);                               // :664
    if (!(r == r)) return 0;
    return (int32_t)(uint16_t)(long long)r;
}

// One kept cell into the output at `slot`: the 16-byte mvs_cell of the C ABI, or -- when the caller streams its results
// out (mvs_pairwise_stream) -- ONE 64-bit word  (row - pack_row0) << pack_shift | col << 16 | q  that sorts by (row, col)
// as an integer and carries everything the shard writer needs (src/pairwise_comp_optimized.cpp:718-736 uses the column
// deltas and q only); half the bytes to write, sort and move.
__device__ __forceinline__ void store_cell(const PairwiseArgs& a, unsigned long long slot, int32_t row, int32_t col, int32_t P,
                                           int32_t q) {
    if (slot >= a.capacity) return;
    if (a.packed) {
        a.packed[slot] = ((unsigned long long)(unsigned)(row - (int32_t)a.pack_row0) << a.pack_shift) |
                         ((unsigned long long)(unsigned)col << 16) | (unsigned long long)(unsigned)(q & 0xffff);
    } else {
        mvs_cell c;
        c.row = row;
        c.col = col;
        c.dot = P;
        c.q = q;
        a.cells[slot] = c;
    }
}

// Append the kept cells of one wave (one atomic per wave).  mirror: the cell (col, row) is appended too --
// dot, keep test and quantised Jaccard are symmetric in (row, col) bit for bit (fp add commutes).
__device__ __forceinline__ void emit_cell(const PairwiseArgs& a, bool keep, bool mirror, int32_t row, int32_t col,
                                          int32_t P, int lane) {
    if (a.dense) {   // dense byte matrix (the tile-granular comparison's re-check beside flagged tiles): scatter q, no list
        if (keep) {
            const int32_t q = quantize_cell(P, a.d, a.norms_sq[row], a.norms_sq[col]);
            if (q <= 0 || q > 255) *a.dense_flag = 1u;
            a.dense[((int64_t)row - a.dense_row0) * a.dense_ld + col] = (uint8_t)q;
            if (mirror) a.dense[((int64_t)col - a.dense_row0) * a.dense_ld + row] = (uint8_t)q;
        }
        return;
    }
    const unsigned long long mask = __ballot(keep);
    if (mask == 0ULL) return;
    const unsigned long long mmask = __ballot(keep && mirror);
    unsigned long long base = 0;
    const int leader = __ffsll((long long)mask) - 1;
    if (lane == leader) base = atomicAdd(a.counter, (unsigned long long)(__popcll(mask) + __popcll(mmask)));
    base = __shfl(base, leader, 64);
    if (keep) {
        const unsigned long long below = (1ULL << lane) - 1ULL;
        const unsigned long long slot = base + (unsigned long long)(__popcll(mask & below) + __popcll(mmask & below));
        const int32_t q = quantize_cell(P, a.d, a.norms_sq[row], a.norms_sq[col]);
        store_cell(a, slot, row, col, P, q);
        if (mirror) store_cell(a, slot + 1, col, row, P, q);
    }
}

// Room for `mine` entries per lane behind *counter with ONE atomic per wave: returns the lane's first slot.
// (A counter bumped once per 32 x 32 block serialises at the L2 as soon as many blocks hold something.)
__device__ __forceinline__ unsigned long long wave_reserve(unsigned long long* counter, unsigned mine, int lane) {
    unsigned incl = mine;   // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    unsigned long long base = 0;
    if (lane == 63) base = atomicAdd(counter, (unsigned long long)incl);
    base = __shfl(base, 63, 64);
    return base + (incl - mine);
}

// one kept cell (and its mirror image) at `slot`, which advances
__device__ __forceinline__ void write_cell(const PairwiseArgs& a, unsigned long long& slot, bool mirror, int32_t row,
                                           int32_t col, int32_t P) {
    const int32_t q = quantize_cell(P, a.d, a.norms_sq[row], a.norms_sq[col]);
    store_cell(a, slot, row, col, P, q);
    ++slot;
    if (mirror) {
        store_cell(a, slot, col, row, P, q);
        ++slot;
    }
}

}  // namespace
}  // namespace mvs

#endif
