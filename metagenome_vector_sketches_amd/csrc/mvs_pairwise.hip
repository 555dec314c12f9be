// mvs_pairwise.hip -- all-vs-all sketch comparison kernels (gfx950 / CDNA4).
//
// Reference semantics (src/pairwise_comp_optimized.cpp):
//   :135      P = block_i^T * block_j   as int32 (wraps mod 2^32)
//   :139-141  keep (i,j) iff int64(P)/d (truncating) > 0.05*(n2_i + n2_j)        [int32 path]
//             (_16bits.cpp:211-218: keep iff double(P)/d > 0.05*(n2_i + n2_j))   [int16 path]
//   :658-665  J = (P/d) / (n2_r + n2_c - P/d); J = min(J,1); q = uint16(round(J*255))
//
// MI355X design (DESIGN.md "K2"):
//   * sketches live in HBM as signed base-256 int8 limb planes, planes[(row*L + limb)*d_pad + k];
//     v == sum_a limb_a*256^a (mod 2^32), so P == sum_{a+b<=3} 256^(a+b) * <limb_a(i), limb_b(j)> (mod 2^32);
//   * each limb-pair product runs on the int8 matrix cores (v_mfma_i32_16x16x64_i8 / 32x32x32, exact int32
//     accumulation: |sum| <= 2 * 128*128*d_pad < 2^31 for d_pad <= 32768);
//   * workgroup = 512 threads = 8 waves, tile = 128 x 128 (or 256 x 256) samples, k-slices of 64 bytes are
//     copied HBM/L2 -> LDS with global_load_lds (16 B per lane, no VGPR round trip) into a 4-stage ring;
//     the LDS image is [limb][sample][64 B] with the 16-byte chunks XOR-swizzled on the SOURCE address side
//     so that the ds_read_b128 fragment reads are bank-conflict free;
//   * TWO-STAGE comparison (default for two limbs): the same MFMA kernel first runs ONE pass on a coarse int8
//     plane c = round(v / m_row) and drops every pair a proven bound rules out (MODE 2, "filter" below); the
//     surviving candidate pairs get their exact int32 dot from the limb planes in k_exact_pairs, followed
//     by the reference's fp64 keep test and Jaccard.  The exact kernel on every cell (MODE 0) remains for
//     other limb counts, dense results and MVS_PAIRWISE_FILTER=0: its epilogue recombines the limb products,
//     rejects almost every cell with one integer compare against a conservative per-sample threshold sum and
//     runs the fp64 test only on the survivors;
//   * kept cells / candidates are appended with one atomic per wave and tile (masks parked in LDS, wave
//     prefix sum), then sorted by (row, col);
//   * workgroup -> tile mapping walks 16 x 16-tile super-patches, each of the 8 XCDs (blockIdx % 8)
//     taking a 4 x 8 sub-patch, so that one XCD's L2 serves 12 operand panels to 32 tiles.
#include "mvs_internal.h"
#include "mvs_encode.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

#include <rocprim/device/device_merge_sort.hpp>
#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>

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

// ---------------------------------------------------------------------------------------------------
// MFMA kernel.  L = limbs (1 or 2).  MODE 0: comparison, 1: dense dots, 2: filter on the coarse plane.
// NST = LDS ring depth.
// LDS: NST stages x [A region | B region], region = [L][128 samples][64 B] (one 64-byte k-slice).
// The ring keeps NST-1 slices in flight: the kernel is bound by the latency of the HBM/L2 -> LDS
// copies (about 1-2 us under load), so what matters is the number of bytes in flight per CU.
// ---------------------------------------------------------------------------------------------------
constexpr int kSK = 64;   // bytes (= int8 k values) per ring stage

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    static_assert(N >= 0 && N <= 24, "unsupported vmcnt");
#define MVS_VMCNT_CASE(n) else if constexpr (N == n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    MVS_VMCNT_CASE(1) MVS_VMCNT_CASE(2) MVS_VMCNT_CASE(3) MVS_VMCNT_CASE(4) MVS_VMCNT_CASE(5) MVS_VMCNT_CASE(6)
    MVS_VMCNT_CASE(7) MVS_VMCNT_CASE(8) MVS_VMCNT_CASE(9) MVS_VMCNT_CASE(10) MVS_VMCNT_CASE(11) MVS_VMCNT_CASE(12)
    MVS_VMCNT_CASE(13) MVS_VMCNT_CASE(14) MVS_VMCNT_CASE(15) MVS_VMCNT_CASE(16) MVS_VMCNT_CASE(17) MVS_VMCNT_CASE(18)
    MVS_VMCNT_CASE(19) MVS_VMCNT_CASE(20) MVS_VMCNT_CASE(21) MVS_VMCNT_CASE(22) MVS_VMCNT_CASE(23) MVS_VMCNT_CASE(24)
#undef MVS_VMCNT_CASE
}

// Geometry (template parameters):
//   WM x WN waves; every wave owns AT*32 rows (AT 32-row MFMA tiles, 2 unless stated) x BT*32 columns of the
//   tile, so the workgroup tile is TM = WM*AT*32 rows x TN = WN*BT*32 columns.  DBUF: fragments double buffered in
//   registers (BT = 1); with BT = 2 the wave's 16 MFMAs per k-step are long enough for the SIMD's other
//   wave to hide the fragment reads, and the 192 accumulator registers leave no room for a second buffer.
// KARA: the L = 3 planes are (l0, l1, l0+l1) of base-128 digits; only the three "diagonal" products
//       X = <l0,l0'>, Z = <l1,l1'>, Y = <l0+l1, l0'+l1'> are formed and P = X + 128(Y-X-Z) + 16384 Z.
// ABL: profiling ablations of the k-loop (1 no MFMA, 2 no HBM/L2 -> LDS copies, 3 no fragment reads); the
// second __launch_bounds__ argument is the number of waves per SIMD the register allocation leaves room for
template <int L, bool KARA, int MODE, int NST, int WM, int WN, int BT, int AT = 2, bool DBUF = (BT == 1), int ABL = 0>
__global__ __launch_bounds__(WM * WN * 64, (AT * BT > 2 ? 1 : 2)) void k_pairwise_mfma(const PairwiseArgs a, int n_tr,
                                                                                      int n_tc, int n_spc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int kWavesT = WM * WN;
    constexpr int TM = WM * AT * 32;             // rows (A samples) per tile
    constexpr int TN = WN * BT * 32;             // columns (B samples) per tile
    constexpr int NB = DBUF ? 2 : 1;             // fragment register buffers
    constexpr int kRegion = L * TM * kSK;        // bytes of the A operand region
    constexpr int kRegionB = L * TN * kSK;       // bytes of the B operand region
    constexpr int kStage = kRegion + kRegionB;   // bytes of one stage
    constexpr int kPieces = kStage / 1024;       // 1 KiB pieces per stage
    constexpr int kPPW = kPieces / kWavesT;      // pieces per wave per stage
    constexpr int NS = KARA ? 3 : num_acc_sets(L);
    static_assert(!KARA || L == 3, "Karatsuba scheme has three planes");
    static_assert(kPieces % kWavesT == 0, "stage must split evenly over the waves");
    static_assert(TM % TN == 0 || TN % TM == 0, "tile edges must nest");

    const TileCoord tc = map_tile(blockIdx.x, blockIdx.y, n_tr, n_tc, a.map_mode);
    (void)n_spc;
    if (!tc.valid) return;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN;   // AT*32-row slice of the tile
    const int wn = wave % WN;   // BT*32-column slice of the tile

    const int64_t i0 = a.row_begin + (int64_t)tc.tr * TM;   // first A sample of the tile
    const int64_t j0 = a.col_begin + (int64_t)tc.tc * TN;   // first B sample of the tile

    // Symmetric schedule (row_begin % TM == 0, col_begin == 0, TN | TM): inside the square
    // [row_begin,row_end)^2 a tile strictly below the diagonal is skipped; its cells come from the tile
    // strictly above the diagonal that holds their transposes (DESIGN.md section 4 has the covering argument).
    bool mirror_tile = false;
    if (MODE != 1 && a.symmetric) {
        if (j0 >= a.sym_begin && j0 + TN <= i0) return;
        mirror_tile = j0 >= i0 + TM && j0 < a.sym_end;
    }
    if constexpr (MODE == 2) {   // the filter is not paying on this block (see cand_limit): stop wasting time
        if (*reinterpret_cast<volatile const unsigned int*>(a.cand_stop) != 0u) return;
    }

    // ---- per-lane source pointers of this wave's pieces (k0 = 0).  One piece = 16 LDS rows of 64 B;
    //      lane -> row piece*16 + lane/4, 16-byte slot lane%4 holding logical chunk slot ^ ((s>>2)&3).
    const int8_t* src[kPPW];
#pragma unroll
    for (int p = 0; p < kPPW; ++p) {
        const int piece = wave * kPPW + p;
        const int row = piece * 16 + (lane >> 2);
        const bool is_b = row >= L * TM;
        const int rr = is_b ? row - L * TM : row;
        const int limb = is_b ? rr / TN : rr / TM;
        const int s = is_b ? rr % TN : rr % TM;
        const int c = (lane & 3) ^ ((s >> 2) & 3);
        const int64_t sample = (is_b ? j0 : i0) + s;
        src[p] = (MODE == 2 ? a.coarse : a.planes) + (sample * L + limb) * (int64_t)a.d_pad + c * 16;
    }
    auto stage_copy = [&](int slot, int k0) {
        if (ABL == 2 && k0 != 0) return;   // ablation: no HBM/L2 -> LDS copies after the first slice
#pragma unroll
        for (int p = 0; p < kPPW; ++p) {
            const int piece = wave * kPPW + p;
            char* dst = smem + slot * kStage + piece * 1024;   // wave-uniform; lane data lands at +lane*16
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[p] + k0), (lds_ptr_t)dst, 16, 0, 0);
        }
    };

    // ---- fragment addressing ----
    const int fr = lane & 31;          // row (A) / col (B) inside a 32x32 MFMA tile
    const int fh = lane >> 5;          // k half
    const int key = (fr >> 2) & 3;     // swizzle key (tile bases are multiples of 16 samples)
    const int a_row0 = (wm * AT * 32 + fr) * kSK;                    // + t*32*kSK + limb*TM*kSK
    const int b_row0 = kRegion + (wn * BT * 32 + fr) * kSK;          // + u*32*kSK + limb*TN*kSK

    v16i acc[AT][BT][NS];
#pragma unroll
    for (int t = 0; t < AT; ++t)
#pragma unroll
        for (int u = 0; u < BT; ++u)
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][u][s][r] = 0;

    v4i fa[NB][AT][L], fb[NB][BT][L];
    auto load_frags = [&](int buf, const char* sb, int kk) {
        if (ABL == 3 && sb != smem) return;   // ablation: fragments are read from LDS only once
        const int coff = (((kk * 2 + fh) ^ key) << 4);
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
            for (int u = 0; u < BT; ++u)
                fb[buf][u][l] = *reinterpret_cast<const v4i*>(sb + b_row0 + u * 32 * kSK + l * TN * kSK + coff);
#pragma unroll
            for (int t = 0; t < AT; ++t)
                fa[buf][t][l] = *reinterpret_cast<const v4i*>(sb + a_row0 + t * 32 * kSK + l * TM * kSK + coff);
        }
    };
    // part 0: the group's first MFMA, part 1: the rest, part 2: all.  Splitting lets the loop put the
    // fragment reads of the NEXT k-step between them: hipcc waits lgkmcnt(0) in front of the first MFMA
    // after the loop back-edge, and that wait must not cover reads issued for the next step.
    auto mfma_group = [&](int buf, int part) {
#pragma unroll
        for (int t = 0; t < AT; ++t)
#pragma unroll
            for (int u = 0; u < BT; ++u)
#pragma unroll
                for (int la = 0; la < L; ++la)
#pragma unroll
                    for (int lb = 0; lb < L; ++lb) {
                        if (KARA ? (la != lb) : (la + lb > 3)) continue;   // 256^4 == 0 (mod 2^32)
                        const bool first = t == 0 && u == 0 && la == 0 && lb == 0;
                        if ((part == 0 && !first) || (part == 1 && first)) continue;
                        const int s = KARA ? la : la + lb;
                        if (ABL == 1) continue;   // ablation: no matrix-core work
                        acc[t][u][s] = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa[buf][t][la], fb[buf][u][lb],
                                                                            acc[t][u][s], 0, 0, 0);
                    }
    };

    // ---- main loop over 64-byte k-slices ----
    // Completion of the LDS-DMA copies is tracked by hand with counted s_waitcnt vmcnt(N) (each wave has
    // kPPW copies per slice in flight, oldest first) followed by a raw s_barrier; hipcc's own waitcnt
    // insertion is not relied on for global_load_lds (it was seen to drop the wait, and a plain
    // __syncthreads() would drain the whole ring).
#ifdef MVS_ABLATIONS
    const int nk = (a.debug_flags & 1) ? 0 : a.d_pad / kSK;
#else
    static_assert(ABL == 0, "k-loop ablations need a -DMVS_ABLATIONS build");
    const int nk = a.d_pad / kSK;
#endif
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nk) stage_copy(st, st * kSK);
    // slice 0 landed <=> at most (slices issued after it) * kPPW copies outstanding
    if (nk >= NST - 1) wait_vmcnt<(NST - 2) * kPPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    if (DBUF) load_frags(0, smem, 0);

    int slot = 0;
    for (int kt = 0; kt < nk; ++kt) {
        const char* sb = smem + slot * kStage;
        // refill the slot that slice kt-1 used (its readers passed the previous barrier)
        {
            const int nslot = slot == 0 ? NST - 1 : slot - 1;
            if (kt + NST - 1 < nk) stage_copy(nslot, (kt + NST - 1) * kSK);
        }
        if (DBUF) {
            mfma_group(0, 0);                    // needs only fragments read before the back-edge
            __builtin_amdgcn_sched_barrier(0);
            load_frags(1, sb, 1);                // in flight during the rest of this group
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 1);
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load_frags(0, sb, 0);
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(0, 2);
            __builtin_amdgcn_sched_barrier(0);
            load_frags(0, sb, 1);                // last reads of this slice
            __builtin_amdgcn_sched_barrier(0);
        }
        // slice kt+1 must have landed: allow only the copies of younger slices to be outstanding
        {
            const int younger = nk - kt - 2;   // slices issued after kt+1
            if (younger >= NST - 2) wait_vmcnt<(NST - 2) * kPPW>();
            else if (NST >= 4 && younger == NST - 3) wait_vmcnt<(NST >= 4 ? (NST - 3) : 0) * kPPW>();
            else if (NST >= 5 && younger == NST - 4) wait_vmcnt<(NST >= 5 ? (NST - 4) : 0) * kPPW>();
            else wait_vmcnt<0>();
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's reads of slice kt are done
        __builtin_amdgcn_s_barrier();
        slot = slot == NST - 1 ? 0 : slot + 1;
        if (DBUF) {
            load_frags(0, smem + slot * kStage, 0);   // unconditional: after the last slice this reads a
                                                     // stale (in-bounds) slot and the values are never used
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(1, 2);
        } else {
            mfma_group(0, 2);
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();

    // ---- epilogue ----
#ifdef MVS_ABLATIONS
    if (a.debug_flags & 2) {   // ablation: keep the accumulators alive, skip the epilogue
        int x = 0;
#pragma unroll
        for (int t = 0; t < AT; ++t)
#pragma unroll
            for (int u = 0; u < BT; ++u)
#pragma unroll
                for (int sset = 0; sset < NS; ++sset) x ^= acc[t][u][sset][0] ^ acc[t][u][sset][15];
        if (x == 0x7fffffff) a.counter[1] = 1;
        return;
    }
#endif
    if constexpr (MODE == 2) {
        // filter: a pair can only be kept if  <c_i,c_j>  >  s_i w_j + s_j w_i - a_i p_j - p_i (a_j + p_j)
        // (derivation at k_filter_meta); everything else is dropped without ever forming the exact dot.
        // The test is evaluated for two rows at a time (packed fp32 FMAs); rows come in adjacent pairs in the
        // accumulator layout, so the row constants sit in LDS as {s0 s1 w0 w1 | a0 a1 p0 p1} per row pair.
        // Rows / columns outside the call's ranges get s = +inf there and never pass.
        static_assert(L == 1 && !KARA, "the filter runs on the single coarse plane");
        using v2f = __attribute__((ext_vector_type(2))) float;
        using v4f = __attribute__((ext_vector_type(4))) float;
        float* frow = reinterpret_cast<float*>(smem);                      // TM/2 pairs x 8 floats
        float4* fcol = reinterpret_cast<float4*>(smem + TM * 16);          // TN entries
        for (int x = tid; x < TM + TN; x += kWavesT * 64) {
            const int64_t g = (x < TM ? i0 : j0 - TM) + x;
            float4 m = a.fmeta[g];
            if (g >= (x < TM ? a.row_end : a.col_end)) m = make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
            if (x < TM) {
                float* q = frow + (x >> 1) * 8 + (x & 1);
                q[0] = m.x;
                q[2] = m.y;
                q[4] = m.z;
                q[6] = m.w;
            } else {
                fcol[x - TM] = m;
            }
        }
        __syncthreads();
        // symmetric schedule: inside the square of the row range only the upper triangle is re-checked and its
        // cells are mirrored (the exact kernel computes whole diagonal tiles instead; the filter would hand
        // both (i,j) and (j,i) to the re-check).  Only tiles that touch the diagonal need the per-cell test.
        const bool straddle = a.symmetric && j0 < i0 + TM && j0 + TN > i0;
        const int delta = (int)(j0 - i0);                                  // col - row = col_l - row_l + delta
        // Pass 1 (sweep): one 16-bit mask of passing rows per lane and 32 x 32 block.  Pass 2: ONE atomic per wave
        // reserves room for all of the wave's candidates (a counter bumped once per block of a dense region
        // serialises at the L2: 5e6 bumps cost 35 ms), a wave-level prefix sum gives every lane its range.
        // (the masks wait in LDS, which is idle by now, rather than in registers next to the accumulators)
        unsigned* masks = reinterpret_cast<unsigned*>(smem + (TM + TN) * 16) + wave * (BT * AT * 64) + lane;
        unsigned mine = 0;
        auto sweep = [&](auto tri) {   // tri: the tile touches the diagonal of the symmetric square
            constexpr bool TRI = decltype(tri)::value;
#pragma unroll
            for (int u = 0; u < BT; ++u) {
                const int col_l = (wn * BT + u) * 32 + fr;
                const int64_t col = j0 + col_l;
                const float4 mj = fcol[col_l];
                const float bj = mj.z + mj.w;
                const v2f wj = {mj.y, mj.y}, sj = {mj.x, mj.x}, npj = {-mj.w, -mj.w}, nbj = {-bj, -bj};
                const bool in_square = a.symmetric && col >= a.sym_begin && col < a.sym_end;
                // rows above this one fail the triangle test: col >= row  <=>  row_l <= col_l + delta (columns
                // outside the square are not restricted)
                const int row_max = (TRI && in_square) ? col_l + delta : 0x7fffffff;
#pragma unroll
                for (int t = 0; t < AT; ++t) {
                    const int row_b = wm * AT * 32 + t * 32 + 4 * fh;
                    unsigned m16 = 0;
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        const int row_l = row_b + (r & 3) + 8 * (r >> 2);      // even: rows row_l, row_l + 1
                        const v4f q0 = *reinterpret_cast<const v4f*>(frow + (row_l >> 1) * 8);
                        const v4f q1 = *reinterpret_cast<const v4f*>(frow + (row_l >> 1) * 8 + 4);
                        v2f rhs = v2f{q0[0], q0[1]} * wj;
                        rhs = __builtin_elementwise_fma(v2f{q0[2], q0[3]}, sj, rhs);
                        rhs = __builtin_elementwise_fma(v2f{q1[0], q1[1]}, npj, rhs);
                        rhs = __builtin_elementwise_fma(v2f{q1[2], q1[3]}, nbj, rhs);
                        bool c0 = (float)acc[t][u][0][r] > rhs[0];
                        bool c1 = (float)acc[t][u][0][r + 1] > rhs[1];
                        if (TRI) {
                            c0 = c0 && row_l <= row_max;
                            c1 = c1 && row_l < row_max;
                        }
                        m16 |= (c0 ? 1u << r : 0u) | (c1 ? 2u << r : 0u);
                    }
                    if (ABL >= 2 && acc[t][u][0][0] != 0x7fffffff) m16 = 0;   // ablations compute garbage
                    masks[(u * AT + t) * 64] = m16;
                    mine += (unsigned)__popc(m16);
                }
            }
        };
        if (straddle) sweep(std::true_type{});
        else sweep(std::false_type{});
        if (__ballot(mine != 0) == 0ULL) return;               // the common case: nothing in this wave passes
        unsigned incl = mine;                                   // inclusive prefix sum over the lanes
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        unsigned long long base = 0;
        if (lane == 63) {
            base = atomicAdd(a.cand_counter, (unsigned long long)incl);
            if (base + incl > a.cand_limit) *a.cand_stop = 1u;   // tell the tiles that have not started yet
        }
        base = __shfl(base, 63, 64);
        unsigned long long slot = base + (incl - mine);
#pragma unroll
        for (int u = 0; u < BT; ++u) {
            const int col_l = (wn * BT + u) * 32 + fr;
            const int64_t col = j0 + col_l;
            const bool in_square = a.symmetric && col >= a.sym_begin && col < a.sym_end;
            const int cd = col_l + delta;
#pragma unroll
            for (int t = 0; t < AT; ++t) {
                const int row_b = wm * AT * 32 + t * 32 + 4 * fh;
                unsigned m = masks[(u * AT + t) * 64];            // this lane's own word: no barrier needed
                while (m) {
                    const int r = __ffs((int)m) - 1;
                    m &= m - 1;
                    const int row_l = row_b + (r & 3) + 8 * (r >> 2);
                    const bool mirror = a.mirror_all || (in_square && cd > row_l);
                    if (slot < a.cand_capacity)
                        a.cand[slot] = make_int2((int32_t)(i0 + row_l), mirror ? (int)((unsigned)col | 0x80000000u) : (int)col);
                    ++slot;
                }
            }
        }
        return;
    }
    int32_t* thr = reinterpret_cast<int32_t*>(smem);   // [0,TM): rows, [TM,TM+TN): cols
    if (MODE == 0) {
        for (int x = tid; x < TM + TN; x += kWavesT * 64) {
            const int64_t g = (x < TM ? i0 : j0 - TM) + x;
            thr[x] = a.cand_thr[g];
        }
        __syncthreads();
    }
    auto dot_of = [&](int t, int u, int r) -> int32_t {   // limb products -> int32 dot (mod 2^32)
        uint32_t Pu = (uint32_t)acc[t][u][0][r];
        if (KARA) {
            const uint32_t X = Pu, Z = (uint32_t)acc[t][u][1][r], Y = (uint32_t)acc[t][u][NS - 1][r];
            Pu = X + ((Y - X - Z) << 7) + (Z << 14);
        } else {
#pragma unroll
            for (int s = 1; s < NS; ++s) Pu += (uint32_t)acc[t][u][s][r] << (8 * s);
        }
        return (int32_t)Pu;
    };
    if constexpr (MODE == 1) {
#pragma unroll
        for (int u = 0; u < BT; ++u) {
            const int64_t col = j0 + (wn * BT + u) * 32 + fr;
#pragma unroll
            for (int t = 0; t < AT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t row = i0 + wm * AT * 32 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                    if (row < a.row_end && col < a.col_end)
                        a.dots[(row - a.row_begin) * (a.col_end - a.col_begin) + (col - a.col_begin)] = dot_of(t, u, r);
                }
        }
        return;
    }
    // Pass 1: keep test per cell (integer pre-test, fp64 only for its survivors) -> one 16-bit mask per lane and
    // 32 x 32 block, parked in LDS.  Pass 2: one atomic per wave reserves the wave's cells, then they are written.
    unsigned* masks = reinterpret_cast<unsigned*>(smem + (((TM + TN) * 4 + 15) & ~15)) + wave * (BT * AT * 64) + lane;
    unsigned mine = 0;
#pragma unroll
    for (int u = 0; u < BT; ++u) {
        const int col_l = (wn * BT + u) * 32 + fr;
        const int64_t col = j0 + col_l;
        const bool mirror = a.mirror_all || (mirror_tile && col < a.sym_end);
#pragma unroll
        for (int t = 0; t < AT; ++t) {
            unsigned m16 = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row_l = wm * AT * 32 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh;
                const int64_t row = i0 + row_l;
                const int32_t P = dot_of(t, u, r);
                const bool cand = P >= thr[row_l] + thr[TM + col_l];
                if (__any(cand)) {
                    bool keep = false;
                    if (cand && row < a.row_end && col < a.col_end)
                        keep = keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
                    m16 |= keep ? 1u << r : 0u;
                }
            }
            masks[(u * AT + t) * 64] = m16;
            mine += (unsigned)__popc(m16) << (mirror ? 1 : 0);
        }
    }
    if (__ballot(mine != 0) == 0ULL) return;
    unsigned long long out_slot = wave_reserve(a.counter, mine, lane);
#pragma unroll
    for (int u = 0; u < BT; ++u) {
        const int64_t col = j0 + (wn * BT + u) * 32 + fr;
        const bool mirror = a.mirror_all || (mirror_tile && col < a.sym_end);
#pragma unroll
        for (int t = 0; t < AT; ++t) {
            const unsigned m = masks[(u * AT + t) * 64];   // this lane's own word
            if (__ballot(m != 0) == 0ULL) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (m & (1u << r))
                    write_cell(a, out_slot, mirror, (int32_t)(i0 + wm * AT * 32 + t * 32 + (r & 3) + 8 * (r >> 2) + 4 * fh),
                               (int32_t)col, dot_of(t, u, r));
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Default kernel for two base-256 limbs, on the 16x16x64 int8 MFMA shape (MVS_PAIRWISE_VARIANT=6): same tile (128 x 128, 8
// waves 2 x 4, wave tile 64 x 32), same LDS ring and byte traffic as variant 0, but one MFMA covers a whole
// 64-byte k-slice, so there is a single fragment load + 32 MFMAs (16 cycles each) per slice.  On bf16 the
// 16-wide shape sustains a higher clock under load than the 32-wide one (MI355X_MICROARCH "DVFS give-back"
// item 7); measured here for int8: 5-7 % faster than the 32x32x32 kernel at identical traffic.
// LDS image as in the main kernel but with chunk swizzle key g[(s>>2)&3], g = {0,2,3,1}: with rows on
// lane&15 and chunks on lane>>4 that permutation makes every ds_read_b128 lane group hit 16 distinct slots.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ int swz16(int s) { return (0x78 >> (((s >> 2) & 3) * 2)) & 3; }   // {0,2,3,1}

// Epilogue shared by the two-limb 16x16x64 kernels (wave tile 64 x 32 of a 128 x 128 tile, accumulators
// acc[t][u][a+b]: row = wm*64 + t*16 + (lane>>4)*4 + r, column = wn*32 + u*16 + (lane&15)): recombine the limb
// products, MODE 1: store the dots; MODE 0: integer pre-test, fp64 keep test for its survivors, one atomic per wave.
template <int MODE>
__device__ __forceinline__ void epilogue_exact16(const PairwiseArgs& a, v4i (&acc)[4][2][3], char* smem, int tid, int lane,
                                                 int wave, int wm, int wn, int64_t i0, int64_t j0, bool mirror_tile) {
    constexpr int TM = 128, TN = 128, kWavesT = 8;
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();
    int32_t* thr = reinterpret_cast<int32_t*>(smem);
    if (MODE == 0) {
        for (int x = tid; x < TM + TN; x += kWavesT * 64) thr[x] = a.cand_thr[(x < TM ? i0 : j0 - TM) + x];
        __syncthreads();
    }
    auto dot_of = [&](int t, int u, int r) -> int32_t {
        return (int32_t)((uint32_t)acc[t][u][0][r] + ((uint32_t)acc[t][u][1][r] << 8) + ((uint32_t)acc[t][u][2][r] << 16));
    };
    if constexpr (MODE == 1) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t col = j0 + wn * 32 + u * 16 + fr;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int64_t row = i0 + wm * 64 + t * 16 + fq * 4 + r;
                    if (row < a.row_end && col < a.col_end)
                        a.dots[(row - a.row_begin) * (a.col_end - a.col_begin) + (col - a.col_begin)] = dot_of(t, u, r);
                }
        }
        return;
    }
    if (a.dense) {
        // Dense results (mvs_pairwise_stream where the exact kernel runs): no list, no counter -- every cell of the tile
        // gets a byte in a row-major matrix, q for a kept cell and 0 otherwise, and a later pass turns rows into CSR.
        // The tile is staged in LDS twice, as it is and transposed (a lane holds four consecutive rows of one column:
        // one ds_write_b32 of the transposed image), so that both the tile and -- above the diagonal of the symmetric
        // square -- its mirror image leave as whole 128-byte lines.
        constexpr int LD = 144;                                             // LDS row stride (bytes)
        uint8_t* tile = reinterpret_cast<uint8_t*>(smem) + 2048;            // behind thr[256]
        uint8_t* tileT = tile + 128 * LD;
        bool odd = false;                                                   // a kept cell whose q is not in 1..255
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int col_l = wn * 32 + u * 16 + fr;
            const int64_t col = j0 + col_l;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                unsigned packed = 0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row_l = wm * 64 + t * 16 + fq * 4 + r;
                    const int64_t row = i0 + row_l;
                    const int32_t P = dot_of(t, u, r);
                    const bool cand = P >= thr[row_l] + thr[TM + col_l];
                    unsigned q8 = 0;
                    if (__any(cand)) {
                        if (cand && row < a.row_end && col < a.col_end &&
                            keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff)) {
                            const int32_t q = quantize_cell(P, a.d, a.norms_sq[row], a.norms_sq[col]);
                            odd = odd || q <= 0 || q > 255;
                            q8 = (unsigned)q & 0xffu;
                        }
                    }
                    tile[row_l * LD + col_l] = (uint8_t)q8;
                    packed |= q8 << (8 * r);
                }
                if (mirror_tile) *reinterpret_cast<unsigned*>(tileT + col_l * LD + wm * 64 + t * 16 + fq * 4) = packed;
            }
        }
        if (__any(odd) && lane == 0) *a.dense_flag = 1u;
        __syncthreads();
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + 512 * j, rl = idx >> 3, seg = idx & 7;    // 128 rows x 8 segments of 16 bytes
            if (i0 + rl < a.row_end)
                *reinterpret_cast<v4i*>(a.dense + (i0 + rl - a.dense_row0) * a.dense_ld + j0 + seg * 16) =
                    *reinterpret_cast<const v4i*>(tile + rl * LD + seg * 16);
            if (mirror_tile && j0 + rl < a.sym_end)                          // row j0 + rl of the matrix, columns i0 ..
                *reinterpret_cast<v4i*>(a.dense + (j0 + rl - a.dense_row0) * a.dense_ld + i0 + seg * 16) =
                    *reinterpret_cast<const v4i*>(tileT + rl * LD + seg * 16);
        }
        return;
    }
    // pass 1: keep masks (16 cells per lane and column) parked in LDS; pass 2: one atomic per wave, then the writes
    unsigned* masks = reinterpret_cast<unsigned*>(smem + (TM + TN) * 4) + wave * 128 + lane;
    unsigned mine = 0;
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int col_l = wn * 32 + u * 16 + fr;
        const int64_t col = j0 + col_l;
        const bool in_sq = a.symmetric && col >= a.sym_begin && col < a.sym_end;
        const bool mirror = in_sq ? mirror_tile : a.mirror_all != 0;
        unsigned m16 = 0;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row_l = wm * 64 + t * 16 + fq * 4 + r;
                const int64_t row = i0 + row_l;
                const int32_t P = dot_of(t, u, r);
                const bool cand = P >= thr[row_l] + thr[TM + col_l];
                if (__any(cand)) {
                    bool keep = false;
                    if (cand && row < a.row_end && col < a.col_end)
                        keep = keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
                    m16 |= keep ? 1u << (t * 4 + r) : 0u;
                }
            }
        masks[u * 64] = m16;
        mine += (unsigned)__popc(m16) << (mirror ? 1 : 0);
    }
    if (__ballot(mine != 0) == 0ULL) return;
    unsigned long long out_slot = wave_reserve(a.counter, mine, lane);
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int64_t col = j0 + wn * 32 + u * 16 + fr;
        const bool in_sq = a.symmetric && col >= a.sym_begin && col < a.sym_end;
        const bool mirror = in_sq ? mirror_tile : a.mirror_all != 0;
        const unsigned m = masks[u * 64];   // this lane's own word
        if (__ballot(m != 0) == 0ULL) continue;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (m & (1u << (t * 4 + r)))
                    write_cell(a, out_slot, mirror, (int32_t)(i0 + wm * 64 + t * 16 + fq * 4 + r), (int32_t)col,
                               dot_of(t, u, r));
    }
}

template <int MODE, int NST>
__global__ __launch_bounds__(512, 2) void k_pairwise_mfma16(const PairwiseArgs a, int n_tr, int n_tc, int n_spc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int L = 2, TM = 128, TN = 128, kWavesT = 8, WN = 4;
    constexpr int kRegion = L * TM * kSK, kStage = 2 * kRegion, kPPW = kStage / 1024 / kWavesT;   // 4
    const TileCoord tc = map_tile(blockIdx.x, blockIdx.y, n_tr, n_tc, a.map_mode);
    (void)n_spc;
    if (!tc.valid) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int64_t i0 = a.row_begin + (int64_t)tc.tr * TM, j0 = a.col_begin + (int64_t)tc.tc * TN;
    bool mirror_tile = false;
    if (MODE == 0 && a.symmetric) {
        if (j0 >= a.sym_begin && j0 + TN <= i0) return;
        mirror_tile = j0 >= i0 + TM && j0 < a.sym_end;
    }
    const int8_t* src[kPPW];
#pragma unroll
    for (int p = 0; p < kPPW; ++p) {
        const int row = (wave * kPPW + p) * 16 + (lane >> 2);
        const bool is_b = row >= L * TM;
        const int rr = is_b ? row - L * TM : row;
        const int limb = rr / 128, s = rr % 128;
        const int c = (lane & 3) ^ swz16(s);
        src[p] = a.planes + (((is_b ? j0 : i0) + s) * L + limb) * (int64_t)a.d_pad + c * 16;
    }
    auto stage_copy = [&](int slot, int k0) {
#pragma unroll
        for (int p = 0; p < kPPW; ++p)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src[p] + k0),
                                             (lds_ptr_t)(smem + slot * kStage + (wave * kPPW + p) * 1024), 16, 0, 0);
    };
    const int fr = lane & 15, fq = lane >> 4;                // row / col inside a 16x16 tile, 16-byte k chunk
    const int coff = (fq ^ swz16(fr)) << 4;                  // tile bases are multiples of 16 samples
    const int a_row0 = (wm * 64 + fr) * kSK + coff;          // + t*16*kSK + limb*TM*kSK
    const int b_row0 = kRegion + (wn * 32 + fr) * kSK + coff;

    v4i acc[4][2][3];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int s = 0; s < 3; ++s) acc[t][u][s] = v4i{0, 0, 0, 0};
    v4i fa[2][4][L], fb[2][2][L];
    auto load_frags = [&](int buf, const char* sb) {
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
                fb[buf][u][l] = *reinterpret_cast<const v4i*>(sb + b_row0 + u * 16 * kSK + l * TN * kSK);
#pragma unroll
            for (int t = 0; t < 4; ++t)
                fa[buf][t][l] = *reinterpret_cast<const v4i*>(sb + a_row0 + t * 16 * kSK + l * TM * kSK);
        }
    };
    auto mfma_group = [&](int buf) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int la = 0; la < L; ++la)
#pragma unroll
                    for (int lb = 0; lb < L; ++lb)
                        acc[t][u][la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[buf][t][la], fb[buf][u][lb],
                                                                                  acc[t][u][la + lb], 0, 0, 0);
    };
    const int nk = a.d_pad / kSK;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st)
        if (st < nk) stage_copy(st, st * kSK);
    if (nk >= NST - 1) wait_vmcnt<(NST - 2) * kPPW>();
    else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    load_frags(0, smem);
    int slot = 0;
    for (int kt = 0; kt < nk; kt += 2) {      // two slices per iteration so the fragment buffers have static indices
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (kt + h >= nk) break;
            const int nslot = slot == 0 ? NST - 1 : slot - 1;
            if (kt + h + NST - 1 < nk) stage_copy(nslot, (kt + h + NST - 1) * kSK);
            const int younger = nk - (kt + h) - 2;
            if (younger >= NST - 2) wait_vmcnt<(NST - 2) * kPPW>();
            else if (NST >= 4 && younger == NST - 3) wait_vmcnt<(NST >= 4 ? (NST - 3) : 0) * kPPW>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's fragments of slice kt+h are in registers
            __builtin_amdgcn_s_barrier();
            slot = slot == NST - 1 ? 0 : slot + 1;
            load_frags(h ^ 1, smem + slot * kStage);            // next slice's fragments, in flight during the MFMAs
            __builtin_amdgcn_sched_barrier(0);
            mfma_group(h);
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    epilogue_exact16<MODE>(a, acc, smem, tid, lane, wave, wm, wn, i0, j0, mirror_tile);
}

// ---------------------------------------------------------------------------------------------------
// "Ping-pong" kernel (16x16x64 int8 MFMA).  Measured on the ring kernels above (ablation builds, 100k x 2048
// filter pass of 11.4 ms): the HBM/L2 -> LDS copies + fragment reads alone take 5.2 ms of k-loop, the MFMAs alone
// 5.0 ms, together 9.6 ms -- the eight waves of a workgroup move in lock step (all issue copies, all read
// fragments, all run MFMAs), so the two halves of the work hardly overlap.  Here the two wave groups of the
// workgroup -- waves 0-3 and 4-7, i.e. the two waves of every SIMD -- run half a step apart: a step of a wave is
//     LOAD  : 12 ds_read_b128 (all fragments of one 64-byte k-slice), 4 LDS-DMA pieces of a later slice, counted
//             vmcnt wait, s_barrier
//     MFMA  : 32 x v_mfma_i32_16x16x64_i8 on those fragments, s_barrier
// and group 1 starts with one extra s_barrier (group 0 ends with one), so that in every barrier interval one group
// is in LOAD and the other in MFMA: the SIMD's matrix pipe always has one wave feeding it while its partner moves
// data.  Fragments need no double buffer (a slice is read in one interval and consumed in the next).
// LDS: ring of NST stages of 32 KiB, copies run D = NST - 2 slices ahead:
//   RAW  slice s is read by group 0 in interval 2s and by group 1 in 2s+1; every wave waits (vmcnt) for its own
//        pieces of slice s+1 at the end of LOAD(s), i.e. before the barriers that open intervals 2s+1 / 2s+2;
//   WAR  the copies issued in LOAD(s) overwrite the slot of slice s-2, last read two intervals (group 1: three)
//        earlier and retired by an lgkmcnt wait in between.
// MODE 2 (filter): one coarse plane, tile 256 x 256, wave tile 128 x 64 (8 x 4 MFMA tiles, 128 accumulators).
// MODE 0 / 1 (comparison / dots on two base-256 limbs): tile 128 x 128, wave tile 64 x 32, the four limb products
// of a slice are the 32 MFMAs; accumulators and epilogue as in k_pairwise_mfma16.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
struct PpGeom {
    static constexpr bool kFilter = MODE == 2;
    static constexpr int L = kFilter ? 1 : 2;
    static constexpr int TM = kFilter ? 256 : 128, TN = TM;
    static constexpr int WROWS = TM / 2, WCOLS = TN / 4;          // wave tile (2 x 4 waves)
    static constexpr int kRegion = L * TM * kSK;                  // 16 KiB per operand either way
    static constexpr int kStage = 2 * kRegion;                    // 32 KiB per slice
    static constexpr int kPPW = kStage / 1024 / 8;                // 4 pieces per wave and slice
};

// filter epilogue for the 16x16 accumulator layout (same test as in k_pairwise_mfma<.., MODE 2>)
// fm: this thread's entry of the tile's row / column constants (thread x < 256: row i0 + x, else column j0 + x - 256),
// loaded by the caller before the k-loop so that its latency is not paid here
__device__ __forceinline__ void epilogue_filter16(const PairwiseArgs& a, v4i (&acc)[8][4], char* smem, int tid, int lane,
                                                  int wave, int wm, int wn, int64_t i0, int64_t j0, float4 fm,
                                                  unsigned long long* stamp = nullptr) {
    constexpr int TM = 256, TN = 256;
    using v4f = __attribute__((ext_vector_type(4))) float;
    const int fr = lane & 15, fq = lane >> 4;
    __syncthreads();                                                   // every wave is done with the ring
#ifdef MVS_ABLATIONS
    if (stamp && tid == 0) stamp[4] = wall_clock64();
#endif
    // The threshold of a 16 x 16 block of cells is a rank-4 product,
    //     T_ij = [s_i w_i a_i p_i] . [w_j s_j -p_j -(a_j + p_j)] ,
    // i.e. ONE v_mfma_f32_16x16x4_f32 (an fp32 fma chain per cell, like the vector code it replaces) whose result
    // lands in the accumulator layout of the int8 products: the matrix pipe -- idle in the epilogue -- forms the
    // thresholds and subtracts them.  Operands: lane (fr, fq)
    // holds constant #fq of row / column fr of the block, so the constants are staged in planes of 256 floats.
    // Plane stride PS = 256 + 16 floats: lane (fr, fq) reads word fq * PS + base + fr, i.e. bank 16 fq + fr (+ base) --
    // 64 distinct banks.  With a stride of 256 the four fq groups of a wave met on the same 16 banks: a 4-way conflict on
    // every one of the 12 constant reads per wave (SQ_LDS_BANK_CONFLICT 1.5e7 cycles per 100k filter pass, round 2).
    constexpr int PS = TM + 16;
    float* rowc = reinterpret_cast<float*>(smem);                      // [4][PS]: s, w, a, p of the tile's rows
    float* colc = rowc + 4 * PS;                                       // [4][PS]: w, s, -p, -(a + p) of its columns
    if (tid < TM) {
        rowc[tid] = fm.x;
        rowc[PS + tid] = fm.y;
        rowc[2 * PS + tid] = fm.z;
        rowc[3 * PS + tid] = fm.w;
    } else {
        const int cidx = tid - TM;
        colc[cidx] = fm.y;
        colc[PS + cidx] = fm.x;
        colc[2 * PS + cidx] = -fm.w;
        colc[3 * PS + cidx] = -(fm.z + fm.w);
    }
    __syncthreads();
#ifdef MVS_ABLATIONS
    if (stamp && tid == 0) stamp[5] = wall_clock64();
#endif
    const bool straddle = a.symmetric && j0 < i0 + TM && j0 + TN > i0;
    const int delta = (int)(j0 - i0);                                  // col - row = col_l - row_l + delta
    // one 32-bit mask per lane and 16-column group: bit t*4 + r <=> row wm*128 + t*16 + fq*4 + r passes
    unsigned m32[4] = {0u, 0u, 0u, 0u};
    float bop[4];
    int row_max[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int col_l = wn * 64 + u * 16 + fr;
        const int64_t col = j0 + col_l;
        bop[u] = colc[fq * PS + col_l];
        const bool in_square = a.symmetric && col >= a.sym_begin && col < a.sym_end;
        row_max[u] = (straddle && in_square) ? col_l + delta : 0x7fffffff;   // col >= row  <=>  row_l <= col_l + delta
    }
    // Candidates are rare (1.5e-5 of the cells), so the sweep first only asks "does any lane pass in this row block"
    // -- compares into scalar registers, OR-ed on the scalar unit -- and builds the per-lane masks just for the row
    // blocks that say yes.
    // D = (float)acc - T comes out of the matrix pipe (the converted int8 products go in as the C operand, the row
    // constants negated); a positive float is a positive int32 bit pattern, so "does any cell of this row block
    // pass" is an integer maximum over the 16 values of a lane -- 1.5 vector instructions per cell (convert, half
    // a v_max3) and no traffic through scalar registers.  (A NaN -- padding rows / columns: inf x 0 -- may look
    // positive to the maximum; the exact compare below then drops it.)
#ifdef MVS_ABLATIONS
    // pairwise_debug bit 4: the accumulators are ignored and ONE pseudo-random cell of the wave tile is declared a
    // candidate with probability 225/256 -- the rate of chance candidates on 50k-hash sketches (0.54 M in 76 636 tiles of
    // 8 waves) -- so that a run on constant operands goes through the epilogue's rare path (row-block masks, prefix sum,
    // atomic, append, re-check) as often as a run on sketches does: what that path costs, apart from what the data costs
    const bool inject = (a.debug_flags & 4) != 0;
    unsigned inj = 0xffffffffu;                                        // t | u << 3 | r << 5 | lane << 7, or none
    if (inject) {
        unsigned h = (unsigned)(i0 * 2654435761u) ^ (unsigned)(j0 * 40503u) ^ (unsigned)(wave * 0x9e3779b9u);
        h ^= h >> 15; h *= 0x2c1b3c6du; h ^= h >> 12; h *= 0x297a2d39u; h ^= h >> 15;
        if ((h & 0xffu) < 225u) inj = (h >> 8) & 0x1fffu;
    }
#endif
    auto sweep = [&](auto tri) {
        constexpr bool TRI = decltype(tri)::value;
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const float aop = -rowc[fq * PS + wm * 128 + t * 16 + fr];
            const int row_l = wm * 128 + t * 16 + fq * 4;
            v4f dif[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v4f accf = {(float)acc[t][u][0], (float)acc[t][u][1], (float)acc[t][u][2], (float)acc[t][u][3]};
                dif[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(aop, bop[u], accf, 0, 0, 0);
            }
            int top = 0;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const v4i bits = __builtin_bit_cast(v4i, dif[u]);
                top = max(max(top, bits[0]), max(max(bits[1], bits[2]), bits[3]));
            }
#ifdef MVS_ABLATIONS
            if (inject) top = ((inj & 7u) == (unsigned)t && (inj >> 7) == (unsigned)lane) ? 1 : 0;
#endif
            if (__ballot(top > 0) != 0ULL) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        bool c = dif[u][r] > 0.0f;
#ifdef MVS_ABLATIONS
                        if (inject) c = top > 0 && ((inj >> 3) & 3u) == (unsigned)u && ((inj >> 5) & 3u) == (unsigned)r;
#endif
                        if (TRI) c = c && row_l + r <= row_max[u];
                        m32[u] |= c ? 1u << (t * 4 + r) : 0u;
                    }
            }
        }
    };
    if (straddle) sweep(std::true_type{});
    else sweep(std::false_type{});
#ifdef MVS_ABLATIONS
    if (stamp && tid == 0) stamp[6] = wall_clock64();
#endif
    const unsigned mine = (unsigned)(__popc(m32[0]) + __popc(m32[1]) + __popc(m32[2]) + __popc(m32[3]));
    if (__ballot(mine != 0) == 0ULL) return;               // nothing in this wave passes (its region header stays 0)
    unsigned incl = mine;                                   // inclusive prefix sum over the lanes
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    // Where do the candidates go?  Reserving list space with an atomic means waiting for its RETURN at the very end of
    // the tile, with nothing left to overlap it -- and on 50k-hash sketches nine waves in ten hold a chance candidate:
    // measured on constant operands with candidates injected at that rate, the wait costs 1.5 ms of a 100k filter pass
    // (7.0 -> 8.5 ms).  So a wave with up to kCandRegion candidates writes them, and their count, into a region of its own
    // (indexed by workgroup and wave: plain stores, nothing to wait for) and k_cand_gather moves the regions' contents
    // into the list afterwards; only a wave with more than that (tiles on the diagonal, dense data) takes the atomic.
    const unsigned total = (unsigned)__shfl((int)incl, 63, 64);
    // Tile-granular comparison: where this wave's 128 x 64 cells hold more candidates than re-checking them one by one is
    // worth (a dense region of the result), the whole 256 x 256 tile goes to the exact kernel instead -- the wave flags
    // the tile and lists nothing; what other waves of the tile list is pruned before the re-check.  The first wave to
    // flag a tile counts it; when nearly every tile is flagged the filter is not paying and the launch stops.
    if (a.tile_flag != nullptr && total > a.tile_dense_thr) {
        if (lane == 0) {
            const int64_t t = ((i0 - a.row_begin) >> 8) * (int64_t)a.tile_flag_ld + ((j0 - a.col_begin) >> 8);
            if (atomicExch(a.tile_flag + t, 1u) == 0u) {
                const unsigned before = atomicAdd(a.tile_flag_count, 1u);
                if (before + 1u > a.tile_flag_limit) *a.cand_stop = 1u;
            }
        }
        return;
    }
    const bool to_region = a.cand_hdr != nullptr && total <= (unsigned)kCandRegion;
    unsigned long long slot;
    int2* list;
    unsigned long long list_cap;
    if (to_region) {
        const unsigned long long reg = a.cand_region_base + ((unsigned long long)blockIdx.y * gridDim.x + blockIdx.x) * 8ull + (unsigned)wave;
        if (lane == 63) a.cand_hdr[reg] = total;
        slot = reg * kCandRegion + (incl - mine);
        list = a.cand_ent;
        list_cap = ~0ULL;
    } else {
        unsigned long long base = 0;
        if (lane == 63) {
            base = atomicAdd(a.cand_counter, (unsigned long long)incl);
            if (base + incl > a.cand_limit) *a.cand_stop = 1u;   // tell the tiles that have not started yet
        }
        base = __shfl(base, 63, 64);
        slot = base + (incl - mine);
        list = a.cand;
        list_cap = a.cand_capacity;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int col_l = wn * 64 + u * 16 + fr;
        const int64_t col = j0 + col_l;
        const bool in_square = a.symmetric && col >= a.sym_begin && col < a.sym_end;
        const int cd = col_l + delta;
        unsigned m = m32[u];
        while (m) {
            const int b = __ffs((int)m) - 1;
            m &= m - 1;
            const int row_l = wm * 128 + (b >> 2) * 16 + fq * 4 + (b & 3);
            // inside the symmetric square the diagonal decides; outside it (a block whose transpose no other launch computes)
            // mirror_all does -- the two meet in one launch only for a block plan
            const bool mirror = in_square ? cd > row_l : a.mirror_all != 0;
            if (slot < list_cap)
                list[slot] = make_int2((int32_t)(i0 + row_l), mirror ? (int)((unsigned)col | 0x80000000u) : (int)col);
            ++slot;
        }
    }
}

// The candidates the filter's waves left in their regions (see epilogue_filter16) are appended to the candidate list:
// one thread per region, a block-wide prefix sum of the region counts, ONE atomic per block.
__global__ __launch_bounds__(256) void k_cand_gather(const PairwiseArgs a, unsigned long long n_regions) {
    __shared__ unsigned wave_sum[4];
    __shared__ unsigned long long block_base;
    const unsigned long long reg = (unsigned long long)blockIdx.x * 256ull + threadIdx.x;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const unsigned cnt = reg < n_regions ? a.cand_hdr[reg] : 0u;
    unsigned incl = cnt;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
        if (lane >= o) incl += up;
    }
    if (lane == 63) wave_sum[w] = incl;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        before += i < w ? wave_sum[i] : 0u;
        total += wave_sum[i];
    }
    if (total == 0) return;                                  // block-uniform
    if (threadIdx.x == 0) block_base = atomicAdd(a.cand_counter, (unsigned long long)total);
    __syncthreads();
    unsigned long long slot = block_base + before + (incl - cnt);
    for (unsigned e = 0; e < cnt; ++e, ++slot)
        if (slot < a.cand_capacity) a.cand[slot] = a.cand_ent[reg * kCandRegion + e];
}


// ---------------------------------------------------------------------------------------------------
// Streaming filter for a few rows against very many columns (a search: 1 .. 1023 query sketches against a resident
// database; one of very many shards).  The tile kernels above fetch 64-byte k-slices of 256 columns per workgroup
// through LDS and live on L2 reuse between neighbouring tiles; a block a few rows high has none, and they ran at
// 1.3-3 TB/s of the 8 TB/s HBM peak (LABNOTES.md: section 7, round 3).  Here the ROWS are resident -- the coarse plane of 16 * RB
// query rows in LDS (row stride d_pad + search_row_pad bytes: conflict-free for ds_read_b128's four lane groups, see there) --
// and the COLUMNS stream: a lane loads 16 consecutive k-bytes of one column's coarse row straight from global memory,
// which is exactly the B fragment of v_mfma_i32_16x16x64_i8 (column = lane & 15, k quarter = lane >> 4), three k-slices
// of four column blocks in flight per wave (12 KiB; 8 waves per CU).  Per k-slice a wave reads RB A fragments from LDS
// and issues 4 * RB MFMAs on 4 column blocks, so LDS traffic is a quarter of what one column block per fragment would need.
// HBM traffic = the coarse plane once per group of 16 * RB rows; the workgroups of different groups that walk the same
// columns sit on the same XCD and start together, so the groups after the first mostly hit in that XCD's L2.
// Epilogue: the filter's threshold test (k_filter_meta) per cell, candidates appended with one atomic per wave.
// Not for the symmetric schedule (a block inside its own square has thousands of rows).
// ---------------------------------------------------------------------------------------------------
// Row padding of the resident rows.  Lane (fr = lane & 15, fq = lane >> 4) reads the 16-byte slot  fr * stride / 16 + fq  (mod 16
// slots of the 256-byte bank row).  ds_read_b128 is serviced in four groups of 16 lanes that are NOT contiguous
// (MI355X_MICROARCH.md, LDS: {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...): a group holds rows {0-3, 12-15} at one k quarter and
// rows {4-11} at the next.  With stride / 16 = 1 (mod 16) -- the 16-byte pad of rounds 3-5 -- row 12 at quarter 0 and row 11 at
// quarter 1 share a slot in every group (SQ_LDS_BANK_CONFLICT: one extra cycle per read, profiles/r04_srch_*); any odd
// multiplier has such a pair.  With stride / 16 = 2 (mod 16) the eight rows of a half land on the eight even slots (no two of
// them are 8 apart), the other half, one quarter on, on the odd ones: conflict-free.
__host__ __device__ inline int search_row_pad(int d_pad) { return (d_pad & 255) == 0 ? 32 : 160; }   // d_pad is a multiple of 128

template <int RB, int NB>
__global__ __launch_bounds__(512) void k_search_filter(const PairwiseArgs a, int groups, long long chunks_total) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int QG = 16 * RB;
    const int stride = a.d_pad + search_row_pad(a.d_pad);
    int8_t* As = reinterpret_cast<int8_t*>(smem);
    float4* rowc = reinterpret_cast<float4*>(smem + (size_t)QG * stride);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // walkers of the column chunks x row groups.  map 0: the groups of one walker sit on ONE XCD (consecutive workgroups of
    // an XCD: they start together and share its L2); map 1: on consecutive XCDs (they meet in the memory-side cache)
    int g, slot, slots;
    unsigned xcd;
    if (a.map_mode == 1) {
        g = (int)(blockIdx.x % (unsigned)groups);
        slot = (int)(blockIdx.x / (unsigned)groups);
        slots = (int)(gridDim.x / (unsigned)groups);
        xcd = 0;
    } else {
        xcd = blockIdx.x & 7u;
        const unsigned j = blockIdx.x >> 3;
        g = (int)(j % (unsigned)groups);
        slot = (int)(j / (unsigned)groups);
        slots = (int)((gridDim.x >> 3) / (unsigned)groups);
    }
    const long long chunk_first = a.map_mode == 1 ? slot : (long long)slot * 8 + xcd;
    const long long chunk_step = a.map_mode == 1 ? slots : 8LL * slots;
    const int64_t q0 = a.row_begin + (int64_t)g * QG;
    {   // the group's rows of the coarse plane -> LDS (rows beyond the block: zeros, and a threshold nothing passes)
        const int per_row = a.d_pad / 16;
        for (int idx = tid; idx < QG * per_row; idx += 512) {
            const int r = idx / per_row, c16 = idx - r * per_row;
            v4i v = v4i{0, 0, 0, 0};
            if (q0 + r < a.row_end) v = *reinterpret_cast<const v4i*>(a.coarse + (q0 + r) * (int64_t)a.d_pad + c16 * 16);
            *reinterpret_cast<v4i*>(As + (size_t)r * stride + c16 * 16) = v;
        }
        if (tid < QG) rowc[tid] = q0 + tid < a.row_end ? a.fmeta[q0 + tid] : make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    }
    __syncthreads();
    const int fr = lane & 15, fq = lane >> 4;
    const int nk = a.d_pad / kSK;
    const int8_t* a_base = As + (size_t)fr * stride + fq * 16;
    // the chunk grid starts at a multiple of 16 columns (a block of 16 columns is one unit of the fragment-major plane);
    // columns in front of col_begin get the threshold nothing passes, like those beyond col_end
    const int64_t col_base = a.col_begin & ~(int64_t)15;
    const bool fm = a.coarse_fm != nullptr;
    const int kstep = fm ? 1024 : kSK;                                         // bytes from one k-slice to the next
    for (long long chunk = chunk_first; chunk < chunks_total; chunk += chunk_step) {
        if (*reinterpret_cast<volatile const unsigned int*>(a.cand_stop) != 0u) break;
        const int64_t c0 = col_base + chunk * 512 + wave * 64;
        const int8_t* bp[4];
        float4 cm[4];
#pragma unroll
        for (int cb = 0; cb < 4; ++cb) {
            const int64_t col = c0 + cb * 16 + fr;
            const int64_t cl = col < a.n_alloc ? col : a.n_alloc - 1;          // loads stay inside the plane
            const int64_t blk = (c0 + cb * 16 < a.n_alloc ? c0 + cb * 16 : a.n_alloc - 16) >> 4;
            bp[cb] = fm ? a.coarse_fm + blk * (int64_t)nk * 1024 + lane * 16 : a.coarse + cl * (int64_t)a.d_pad + fq * 16;
            cm[cb] = col >= a.col_begin && col < a.col_end ? a.fmeta[cl] : make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
        }
        v4i acc[RB][4];
#pragma unroll
        for (int rb = 0; rb < RB; ++rb)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) acc[rb][cb] = v4i{0, 0, 0, 0};
        // NB register buffers of one k-slice x four column blocks each: NB - 1 slices (4 KiB per wave each) are in flight while
        // one is consumed.  The columns come from HBM once (the groups that walk the same columns miss together), so what a CU
        // keeps in flight against ~2 us of loaded latency sets the rate: 3 buffers = 12 KiB per wave ran 256 query rows at
        // 12.6 TB/s of L2 -> CU traffic, 0.37 of the HBM peak in algorithmic bytes (profiles/r05_srch_*).
        v4i b[NB][4];
#pragma unroll
        for (int i = 0; i < NB - 1; ++i)
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) b[i][cb] = *reinterpret_cast<const v4i*>(bp[cb] + (size_t)(i < nk ? i : nk - 1) * kstep);
        auto step = [&](int ks, v4i (&cur)[4], v4i (&nxt)[4]) {
            // the slice NB - 1 ahead goes into the buffer that was consumed one step ago (clamped at the end: a harmless reload)
            const int kn = ks + NB - 1 < nk ? ks + NB - 1 : nk - 1;
#pragma unroll
            for (int cb = 0; cb < 4; ++cb) nxt[cb] = *reinterpret_cast<const v4i*>(bp[cb] + (size_t)kn * kstep);
            v4i fa[RB];
#pragma unroll
            for (int rb = 0; rb < RB; ++rb) fa[rb] = *reinterpret_cast<const v4i*>(a_base + (size_t)rb * 16 * stride + ks * kSK);
#pragma unroll
            for (int rb = 0; rb < RB; ++rb)
#pragma unroll
                for (int cb = 0; cb < 4; ++cb)
                    acc[rb][cb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[rb], cur[cb], acc[rb][cb], 0, 0, 0);
        };
        int ks = 0;
        for (; ks + NB <= nk; ks += NB) {
#pragma unroll
            for (int i = 0; i < NB; ++i) step(ks + i, b[i], b[(i + NB - 1) % NB]);
        }
#pragma unroll
        for (int i = 0; i < NB - 1; ++i)
            if (ks + i < nk) step(ks + i, b[i], b[(i + NB - 1) % NB]);
        // ---- threshold test (same expression as the tile filters': four fused operations per cell) ----
        unsigned mine = 0;
        unsigned long long hit[RB];          // bit cb * 4 + r of word rb: cell (row rb*16 + fq*4 + r, column block cb) passes
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            unsigned m16 = 0;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float4 rc = rowc[rb * 16 + fq * 4 + r];
#pragma unroll
                for (int cb = 0; cb < 4; ++cb) {
                    float t = rc.x * cm[cb].y;
                    t = fmaf(rc.y, cm[cb].x, t);
                    t = fmaf(rc.z, -cm[cb].w, t);
                    t = fmaf(rc.w, -(cm[cb].z + cm[cb].w), t);
                    m16 |= ((float)acc[rb][cb][r] > t) ? 1u << (cb * 4 + r) : 0u;
                }
            }
            hit[rb] = m16;
            mine += (unsigned)__popc(m16);
        }
        if (__ballot(mine != 0) == 0ULL) continue;
        unsigned incl = mine;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const unsigned up = (unsigned)__shfl_up((int)incl, o, 64);
            if (lane >= o) incl += up;
        }
        unsigned long long base = 0;
        if (lane == 63) {
            base = atomicAdd(a.cand_counter, (unsigned long long)incl);
            if (base + incl > a.cand_limit) *a.cand_stop = 1u;
        }
        base = __shfl(base, 63, 64);
        unsigned long long out = base + (incl - mine);
#pragma unroll
        for (int rb = 0; rb < RB; ++rb) {
            unsigned m = (unsigned)hit[rb];
            while (m) {
                const int b = __ffs((int)m) - 1;
                m &= m - 1;
                const int64_t col = c0 + (b >> 2) * 16 + fr;
                const int64_t row = q0 + rb * 16 + fq * 4 + (b & 3);
                if (out < a.cand_capacity)
                    a.cand[out] = make_int2((int32_t)row, a.mirror_all ? (int)((unsigned)col | 0x80000000u) : (int)col);
                ++out;
            }
        }
    }
}

// ---- tile-granular comparison: flags -> list, candidate pruning ----
// flagged tiles per tile row (one workgroup per row of the 256 x 256 tile grid)
__global__ __launch_bounds__(256) void k_tile_count(const unsigned int* __restrict__ flags, int n_tc, int* __restrict__ row_count) {
    __shared__ int part[4];
    const unsigned int* row = flags + (size_t)blockIdx.x * n_tc;
    int mine = 0;
    for (int t = threadIdx.x; t < n_tc; t += 256) mine += row[t] != 0u;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) row_count[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}

// row-major list of the flagged tiles' ids (tr * n_tc + tc); workgroup tr sums the counts of the rows before it (the grid
// has a few thousand rows at most) and writes its own row's ids in column order.  list[-1 .. ] : the caller passes
// d_list + 1 and gets the total in d_list[0]; ids beyond `cap` entries are not written (a plan that sized the list from the
// previous step's count: the total tells it)
__global__ __launch_bounds__(256) void k_tile_list(const unsigned int* __restrict__ flags, int n_tr, int n_tc,
                                                   const int* __restrict__ row_count, int* __restrict__ list, int cap) {
    __shared__ int part[4];
    __shared__ int run;
    const int tr = blockIdx.x;
    int before = 0;
    for (int t = threadIdx.x; t < tr; t += 256) before += row_count[t];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = before;
    __syncthreads();
    if (threadIdx.x == 0) {
        run = part[0] + part[1] + part[2] + part[3];
        if (tr == n_tr - 1) list[-1] = run + row_count[tr];
    }
    __syncthreads();
    const unsigned int* row = flags + (size_t)tr * n_tc;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (int t0 = 0; t0 < n_tc; t0 += 256) {
        const int t = t0 + threadIdx.x;
        const bool f = t < n_tc && row[t] != 0u;
        const unsigned long long m = __ballot(f);
        if (lane == 0) part[w] = __popcll(m);
        __syncthreads();
        int pos = run + __popcll(m & ((1ULL << lane) - 1ULL));
        for (int i = 0; i < w; ++i) pos += part[i];
        if (f && pos < cap) list[pos] = tr * n_tc + t;
        __syncthreads();
        if (threadIdx.x == 0) run += part[0] + part[1] + part[2] + part[3];
        __syncthreads();
    }
}

// the candidate list without the pairs whose tile is flagged (those cells come from the exact kernel)
__global__ __launch_bounds__(256) void k_cand_prune(const PairwiseArgs a, unsigned long long n_cand, int2* __restrict__ out,
                                                    unsigned long long* __restrict__ out_count) {
    // 512 entries per wave and round (8 per lane), ONE atomic for all of them: with one per 64 entries the 6 500 atomics of a
    // 100k comparison's list queued on the counter's line for 0.08 ms (k_cand_prune "waiting 0.99 of wave cycles", round 5)
    const int lane = threadIdx.x & 63;
    const unsigned long long waves = (unsigned long long)gridDim.x * 4;
    if (n_cand == ~0ULL) {                       // the count is on the device (a plan that runs ahead of its read-backs)
        n_cand = *reinterpret_cast<volatile const unsigned long long*>(a.cand_counter);
        n_cand = n_cand < a.cand_capacity ? n_cand : a.cand_capacity;
    }
    for (unsigned long long base = ((unsigned long long)blockIdx.x * 4 + (threadIdx.x >> 6)) * 512; base < n_cand; base += waves * 512) {
        int2 pr[8];
        unsigned keep = 0;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const unsigned long long i = base + (unsigned long long)k * 64 + lane;
            pr[k] = make_int2(0, 0);
            if (i < n_cand) {
                pr[k] = a.cand[i];
                const int64_t t = (((int64_t)pr[k].x - a.row_begin) >> 8) * (int64_t)a.tile_flag_ld +
                                  (((int64_t)(pr[k].y & 0x7fffffff) - a.col_begin) >> 8);
                keep |= a.tile_flag[t] == 0u ? 1u << k : 0u;
            }
        }
        const unsigned mine = (unsigned)__popc(keep);
        if (__ballot(mine != 0) == 0ULL) continue;
        unsigned long long slot = wave_reserve(out_count, mine, lane);
#pragma unroll
        for (int k = 0; k < 8; ++k)
            if (keep & (1u << k)) out[slot++] = pr[k];
    }
}

// ORDER: 0 fragment reads then copies, 1 copies then fragment reads, 2 by wave parity (half the group's waves each
// way, so that the LDS reads of some overlap the copy issue of the others).  ABL (ablation builds): 1 no MFMA,
// 2 no copies after the prologue, 3 no fragment reads after the first slice.
// PH: phases per slice (1: 32 MFMAs per interval; 2: the slice's A fragments in two halves, 16 MFMAs per interval).
// NT: cache policy of the HBM/L2 -> LDS copies: 0 default, 1 column panels (B region) non-temporal, 2 row panels (A region),
// 3 both.  An XCD walks its sub-patch along a patch ROW, so consecutive groups of 32 tiles share their 4 row panels and
// stream 8 new column panels through the L2; `nt` marks the stream as evict-first.
// BD (needs the fragment-major planes and a tile origin on the 16-sample grid: the launcher checks): the B operand does not
// go through LDS at all -- a wave loads its four B fragments of the NEXT slice straight from the fragment-major plane into
// registers (one coalesced KiB per instruction, as k_search_filter does) while the matrix cores work on the current one,
// and all eight waves copy the A region (two pieces each).  LDS then carries 8 instead of 12 fragment reads per wave and
// slice and half the copy bytes: 80 instead of 128 bytes per clock at full matrix rate, which is its peak.
template <int MODE, int NST, int ORDER = 0, int ABL = 0, int PH = 1, int NT = 0, int BD = 0>
__global__ __launch_bounds__(512, 2) void k_pairwise_pp(const PairwiseArgs a, int n_tr, int n_tc, const PlanSegs segs) {
#ifndef MVS_ABLATIONS
    static_assert(ABL == 0, "ablations need a -DMVS_ABLATIONS build");
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    using G = PpGeom<MODE>;
    constexpr int L = G::L, TM = G::TM, TN = G::TN, kRegion = G::kRegion, kStage = G::kStage, kPPW = G::kPPW;
    constexpr int D = NST - 2;                                   // slices the copies run ahead
    static_assert(NST >= 3 && NST <= 5 && kPPW == 4, "ring geometry");
    static_assert(BD == 0 || (NST == 4 && PH == 1 && ABL == 0 && NT == 0 && ORDER == 0), "the direct-B loop is written for the default ring");
    constexpr int PA = BD ? 2 : kPPW;                            // pieces a wave copies per slice
#ifdef MVS_ABLATIONS
    unsigned long long* stamp = nullptr;
    if (a.stamps && (unsigned long long)blockIdx.y * gridDim.x + blockIdx.x < kStampSlots) {
        stamp = a.stamps + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 8;
        if (threadIdx.x == 0) {
            unsigned xcc, hw;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
            const unsigned long long t = __builtin_readcyclecounter();
            stamp[0] = ((unsigned long long)xcc << 32) | hw;
            stamp[1] = wall_clock64();
            stamp[2] = stamp[1];
            stamp[3] = t;
        }
    }
#endif
    TileCoord tc;
    int64_t org_i = a.row_begin, org_j = a.col_begin;            // where tile (0, 0) of this workgroup's grid sits
    if (MODE == 0 && a.tile_list != nullptr) {
        // the flagged tiles of the tile-granular comparison: entry = a 256 x 256 filter tile = four tiles of this kernel,
        // consecutive indices; XCD label x (blockIdx.x % 8) takes the x-th contiguous eighth of the row-major list, so the
        // tiles that share row and column panels meet in one L2
        int n_list = a.tile_list_n;
        if (n_list < 0) {                        // -(cap + 1): the count is on the device, in front of the list; the grid holds cap
            const int cap = -(n_list + 1), have = a.tile_list[-1];
            n_list = have < cap ? have : cap;
        }
        const unsigned n4 = 4u * (unsigned)n_list, per = (n4 + 7u) / 8u;
        const unsigned idx = (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
        if ((blockIdx.x >> 3) >= per || idx >= n4) return;
        const int entry = a.tile_list[idx >> 2];
        tc.tr = (entry / a.tile_flag_ld) * 2 + (int)((idx >> 1) & 1u);
        tc.tc = (entry % a.tile_flag_ld) * 2 + (int)(idx & 1u);
        tc.valid = tc.tr < n_tr && tc.tc < n_tc;
    } else if (segs.n > 0) {
        // a block plan: the workgroup's segment (uniform: scalar compares on the kernel arguments), then the single-block
        // map inside it -- a segment starts on a multiple of 256 workgroups, so blockIdx.x % 8 is the XCD label there too
        int sg = 0;
        if (segs.order != nullptr) {
            // the balanced order: this XCD label's next tile (uniform: a scalar load)
            const unsigned e = segs.order[(blockIdx.x & 7u) * segs.order_per + (blockIdx.x >> 3)];
            if (e == ~0u) return;
            sg = (int)(e >> 28);
            tc.tr = (int)((e >> 14) & 0x3fffu);
            tc.tc = (int)(e & 0x3fffu);
            tc.valid = true;
        } else {
            for (int k = 1; k < segs.n; ++k) sg += blockIdx.x >= segs.wg_begin[k] ? 1 : 0;
            const unsigned local = blockIdx.x - segs.wg_begin[sg];
            const unsigned per_row = (unsigned)segs.n_spc[sg] * 256u;
            tc = map_tile(local % per_row, local / per_row, segs.n_tr[sg], segs.n_tc[sg], a.map_mode);
        }
        org_i = segs.i_begin[sg];
        org_j = segs.j_begin[sg];
    } else {
        tc = map_tile(blockIdx.x, blockIdx.y, n_tr, n_tc, a.map_mode);
    }
    if (!tc.valid) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;                     // wm is also the wave group: waves w and w+4 share a SIMD
    const int64_t i0 = org_i + (int64_t)tc.tr * TM, j0 = org_j + (int64_t)tc.tc * TN;
    bool mirror_tile = false;
    if (MODE != 1 && a.symmetric) {
        if (j0 >= a.sym_begin && j0 + TN <= i0) return;
        mirror_tile = j0 >= i0 + TM && j0 < a.sym_end;
    }
    if constexpr (MODE == 2) {
        if (*reinterpret_cast<volatile const unsigned int*>(a.cand_stop) != 0u) return;
    }
    // ---- LDS-DMA sources: piece = 16 LDS rows of 64 B, lane -> row piece*16 + lane/4, 16-byte slot lane%4 ----
    // FM (filter, fragment-major coarse plane, tile origins on multiples of 16 samples): a piece is ONE contiguous KiB of
    // the plane -- 16 samples x 64 k values in fragment order -- so a copy instruction touches 8 whole lines instead of 16
    // half lines, its LDS image is the fragment itself (lane l's 16 bytes at l * 16: no swizzle, no bank conflict) and the
    // next k-slice is 1 KiB further on
    const bool fmode = BD != 0 || ((MODE == 2 ? a.coarse_fm != nullptr : a.planes_fm != nullptr) && (((a.row_begin | a.col_begin) & 15) == 0));
    const int kstep = fmode ? 1024 : kSK;
    const int8_t* src[PA];
#pragma unroll
    for (int p = 0; p < PA; ++p) {
        const int row = (wave * PA + p) * 16 + (lane >> 2);      // [0, 2 * L * TM): A region then B region (BD: A region only)
        const bool is_b = row >= L * TM;
        const int rr = is_b ? row - L * TM : row;
        const int limb = rr / TM, s = rr % TM;
        const int c = (lane & 3) ^ swz16(s);
        const int64_t sample = (is_b ? j0 : i0) + s;
        src[p] = (MODE == 2 ? a.coarse : a.planes) + (sample * L + limb) * (int64_t)a.d_pad + c * 16;
        if (fmode) {                                             // the piece = 16 samples of one limb plane
            const int rr0 = (wave * PA + p) * 16 - (is_b ? L * TM : 0);
            const int64_t blk = ((is_b ? j0 : i0) + rr0 % TM) >> 4;
            src[p] = (MODE == 2 ? a.coarse_fm : a.planes_fm) + (blk * L + rr0 / TM) * (int64_t)(a.d_pad / kSK) * 1024 + lane * 16;
        }
    }
    // waves 0-3 copy the A region (pieces 0..15), waves 4-7 the B region: the policy is wave-uniform
    const bool nt_wave = NT == 3 || (NT == 1 && wave >= 4) || (NT == 2 && wave < 4);
    auto copy_piece = [&](const int8_t* g, char* l) {
        if (NT != 0 && nt_wave) __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 2);
        else __builtin_amdgcn_global_load_lds((gbl_ptr_t)g, (lds_ptr_t)l, 16, 0, 0);
    };
    auto stage_copy = [&](int slot, int k0) {
#pragma unroll
        for (int p = 0; p < PA; ++p) copy_piece(src[p] + k0, smem + slot * kStage + (wave * PA + p) * 1024);
    };
    // BD: where the wave's four B fragments of slice 0 sit in the fragment-major plane (fragment i as in b_off below)
    const int8_t* bsrc[4] = {nullptr, nullptr, nullptr, nullptr};
    if constexpr (BD != 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t colb = (j0 + (MODE == 2 ? wn * 64 + i * 16 : wn * 32 + (i >> 1) * 16)) >> 4;
            const int limb = MODE == 2 ? 0 : (i & 1);
            bsrc[i] = (MODE == 2 ? a.coarse_fm : a.planes_fm) + (colb * L + limb) * (int64_t)(a.d_pad / kSK) * 1024 + lane * 16;
        }
    }
    // ---- fragments: 8 of the A operand, 4 of the B operand per slice ----
    const int fr = lane & 15, fq = lane >> 4;
    const int coff = (fq ^ swz16(fr)) << 4;                      // tile bases are multiples of 16 samples
    // MODE 2: A fragment i = rows wm*128 + i*16..; MODE 0/1: A fragment i = (t = i >> 1, limb = i & 1)
    int a_off[8], b_off[4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
        a_off[i] = MODE == 2 ? (fmode ? (wm * 8 + i) * 1024 + lane * 16 : (wm * 128 + i * 16 + fr) * kSK + coff)
                             : (fmode ? (((i & 1) * TM + wm * 64 + (i >> 1) * 16) >> 4) * 1024 + lane * 16
                                      : ((i & 1) * TM + wm * 64 + (i >> 1) * 16 + fr) * kSK + coff);
#pragma unroll
    for (int i = 0; i < 4; ++i)
        b_off[i] = kRegion + (MODE == 2 ? (fmode ? (wn * 4 + i) * 1024 + lane * 16 : (wn * 64 + i * 16 + fr) * kSK + coff)
                                        : (fmode ? (((i & 1) * TN + wn * 32 + (i >> 1) * 16) >> 4) * 1024 + lane * 16
                                                 : ((i & 1) * TN + wn * 32 + (i >> 1) * 16 + fr) * kSK + coff));
    float4 fm = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
    if constexpr (MODE == 2) {   // the epilogue's row / column constants: one per thread, on their way during the k-loop
        const int64_t g = tid < TM ? i0 + tid : j0 + (tid - TM);
        fm = a.fmeta[g];
        if (g >= (tid < TM ? a.row_end : a.col_end)) fm = make_float4(__builtin_inff(), 0.0f, 0.0f, 0.0f);
    }
    v4i fa[8], fb[4];
    constexpr int NACC = MODE == 2 ? 32 : 24;
    v4i accv[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) accv[i] = v4i{0, 0, 0, 0};

#ifdef MVS_ABLATIONS
    const int nk = (a.debug_flags & 1) ? 0 : a.d_pad / kSK;
#else
    const int nk = a.d_pad / kSK;
#endif
    if constexpr (BD != 0) {
        // ---- direct-B loop: per phase 2 copies (slice s + 2) and 4 register loads (B of slice s + 1) are issued; at the end
        // of the phase everything issued in EARLIER phases has to be through (slice s + 1 of A has landed, B of slice s is in
        // its registers), what this phase issued may stay in flight ----
        v4i fbA[4], fbB[4];
#pragma unroll
        for (int st = 0; st < D; ++st)
            if (st < nk) stage_copy(st, st * 1024);
        // (inline asm: a load the compiler tracks makes it wait for vmcnt(0) in front of the MFMAs that use the registers a
        // phase later -- which would drain this phase's copies and loads as well; the explicit waits below do the counting.
        // The compiler believes the asm's result is there at once, so the scheme relies on it leaving fbA / fbB where they
        // are between issue and use: in the generated code the loads' destinations are the MFMAs' operands, and
        // test_fragment_major_planes_change_no_cell / the pw_filter fixture compare this kernel with the LDS-only one)
        auto load_b = [&](v4i& dst, const int8_t* ptr) __attribute__((always_inline)) {
            asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory");
        };
        if (nk > 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i) load_b(fbA[i], bsrc[i]);
        }
        wait_vmcnt<0>();
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();               // group 1 runs one interval behind group 0
        int slot = 0, fill = D % NST;
        // `last`: the phase of the very last slice (the odd tail below) has no B fragments to load -- said at compile time, so
        // that no dead load is emitted there at all: hipcc gave the four unused results of such loads ONE register quadruple
        // and re-used it for the A fragments, code that would race if the (never true) condition around it ever held;
        // tools/check_isa.py walks the generated code for exactly this kind of thing
        auto phase = [&](int s, v4i (&cur)[4], v4i (&nxt)[4], auto last) __attribute__((always_inline)) {
            const char* sb = smem + slot * kStage;
            const bool more_a = s + D < nk, more_b = !decltype(last)::value && s + 1 < nk;
            if (more_a) {
#pragma unroll
                for (int p = 0; p < PA; ++p) copy_piece(src[p] + (s + D) * 1024, smem + fill * kStage + (wave * PA + p) * 1024);
            }
            if (more_b) {
#pragma unroll
                for (int i = 0; i < 4; ++i) load_b(nxt[i], bsrc[i] + (size_t)(s + 1) * 1024);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const v4i*>(sb + a_off[i]);
            __builtin_amdgcn_sched_barrier(0);
            if (more_a) wait_vmcnt<PA + 4>();
            else if (more_b) wait_vmcnt<4>();
            else wait_vmcnt<0>();
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_setprio(1);
            if constexpr (MODE == 2) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int t = 0; t < 8; ++t)
                        accv[t * 4 + u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[t], cur[u], accv[t * 4 + u], 0, 0, 0);
            } else {
#pragma unroll
                for (int u = 0; u < 2; ++u)
#pragma unroll
                    for (int t = 0; t < 4; ++t)
#pragma unroll
                        for (int la = 0; la < 2; ++la)
#pragma unroll
                            for (int lb = 0; lb < 2; ++lb)
                                accv[(t * 2 + u) * 3 + la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                    fa[t * 2 + la], cur[u * 2 + lb], accv[(t * 2 + u) * 3 + la + lb], 0, 0, 0);
            }
            __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_sched_barrier(0);
            __builtin_amdgcn_s_barrier();
            slot = slot == NST - 1 ? 0 : slot + 1;
            fill = fill == NST - 1 ? 0 : fill + 1;
        };
        int s = 0;
        for (; s + 2 <= nk; s += 2) {
            phase(s, fbA, fbB, std::false_type{});
            phase(s + 1, fbB, fbA, std::false_type{});
        }
        if (s < nk) phase(s, fbA, fbB, std::true_type{});
    } else {
    #pragma unroll
        for (int st = 0; st < D; ++st)
            if (st < nk) stage_copy(st, st * kstep);
        {   // slice 0 has landed <=> only the copies of the slices issued after it are outstanding
            const int younger = (nk < D ? nk : D) - 1;
            if (younger >= 2) wait_vmcnt<2 * kPPW>();
            else if (younger == 1) wait_vmcnt<kPPW>();
            else wait_vmcnt<0>();
        }
        __builtin_amdgcn_s_barrier();
        if (wm == 1) __builtin_amdgcn_s_barrier();                   // group 1 runs one interval behind group 0
        int slot = 0, fill = D % NST;
        for (int s = 0; s < nk; ++s) {
            const char* sb = smem + slot * kStage;
    #pragma unroll
            for (int ph = 0; ph < PH; ++ph) {
                constexpr int AF = 8 / PH, CP = kPPW / PH;               // A fragments / copy pieces per phase
                // ---- LOAD: this phase's fragments (all B fragments belong to phase 0), its share of the copies ----
                const bool copies_first = ORDER == 1 || (ORDER == 2 && (wn & 1));
                auto copies = [&]() {
                    if (s + D < nk && ABL != 2) {
    #pragma unroll
                        for (int p = ph * CP; p < (ph + 1) * CP; ++p)
                            copy_piece(src[p] + (s + D) * kstep, smem + fill * kStage + (wave * kPPW + p) * 1024);
                    }
                };
                if (copies_first) copies();
                if (ABL != 3 || s == 0) {
                    if (ph == 0) {
    #pragma unroll
                        for (int i = 0; i < 4; ++i) fb[i] = *reinterpret_cast<const v4i*>(sb + b_off[i]);
                    }
    #pragma unroll
                    for (int i = ph * AF; i < (ph + 1) * AF; ++i) fa[i] = *reinterpret_cast<const v4i*>(sb + a_off[i]);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!copies_first) copies();
                if (ph == PH - 1) {
                    // slice s+1 must have landed before the barriers that open its readers' intervals: only slices
                    // s+2 .. min(s+D, nk-1) may still be in flight
                    const int last = s + D < nk - 1 ? s + D : nk - 1;
                    const int younger = last - (s + 1);
                    if (younger >= 2) wait_vmcnt<2 * kPPW>();
                    else if (younger == 1) wait_vmcnt<kPPW>();
                    else wait_vmcnt<0>();
                }
                // fragments in registers before the barrier: the MFMA interval then starts on the matrix pipe at once
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                // ---- MFMA ----
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_setprio(1);
                if constexpr (ABL == 1) {
                    // no matrix-core work
                } else if constexpr (MODE == 2) {
    #pragma unroll
                    for (int u = 0; u < 4; ++u)
    #pragma unroll
                        for (int t = ph * AF; t < (ph + 1) * AF; ++t)
                            accv[t * 4 + u] = __builtin_amdgcn_mfma_i32_16x16x64_i8(fa[t], fb[u], accv[t * 4 + u], 0, 0, 0);
                } else {
    #pragma unroll
                    for (int u = 0; u < 2; ++u)
    #pragma unroll
                        for (int t = ph * AF / 2; t < (ph + 1) * AF / 2; ++t)
    #pragma unroll
                            for (int la = 0; la < 2; ++la)
    #pragma unroll
                                for (int lb = 0; lb < 2; ++lb)
                                    accv[(t * 2 + u) * 3 + la + lb] = __builtin_amdgcn_mfma_i32_16x16x64_i8(
                                        fa[t * 2 + la], fb[u * 2 + lb], accv[(t * 2 + u) * 3 + la + lb], 0, 0, 0);
                }
                __builtin_amdgcn_s_setprio(0);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();
            }
            slot = slot == NST - 1 ? 0 : slot + 1;
            fill = fill == NST - 1 ? 0 : fill + 1;
        }
    }
    if (wm == 0) __builtin_amdgcn_s_barrier();                   // group 0 waits for group 1's last interval
#ifdef MVS_ABLATIONS
    if (a.debug_flags & 2) {   // keep the accumulators alive, skip the epilogue
        int x = 0;
#pragma unroll
        for (int i = 0; i < NACC; ++i) x ^= accv[i][0] ^ accv[i][3];
        if (x == 0x7fffffff) a.counter[1] = 1;
        return;
    }
#endif
#ifdef MVS_ABLATIONS
    if (stamp && threadIdx.x == 0) stamp[3] = wall_clock64();   // end of the k-loop
#endif
    if constexpr (MODE == 2) {
#ifdef MVS_ABLATIONS
        epilogue_filter16(a, *reinterpret_cast<v4i(*)[8][4]>(&accv[0]), smem, tid, lane, wave, wm, wn, i0, j0, fm, stamp);
#else
        epilogue_filter16(a, *reinterpret_cast<v4i(*)[8][4]>(&accv[0]), smem, tid, lane, wave, wm, wn, i0, j0, fm);
#endif
#ifdef MVS_ABLATIONS
        if (stamp && threadIdx.x == 0) stamp[2] = wall_clock64();
#endif
    } else {
        epilogue_exact16<MODE>(a, *reinterpret_cast<v4i(*)[4][2][3]>(&accv[0]), smem, tid, lane, wave, wm, wn, i0, j0,
                               mirror_tile);
    }
}

// ---------------------------------------------------------------------------------------------------
// Vector-ALU kernel: one thread per cell, v_dot4_i32_i8 over the limb planes.  Any limb count.
// Independent of the MFMA path (no LDS, no matrix cores): used for limbs > 2 and as a cross-check.
// ---------------------------------------------------------------------------------------------------
template <int MODE>
__global__ __launch_bounds__(256) void k_pairwise_valu(const PairwiseArgs a) {
    const int64_t col = a.col_begin + (int64_t)blockIdx.x * 64 + (threadIdx.x & 63);
    const int64_t row = a.row_begin + (int64_t)blockIdx.y * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    const bool in = row < a.row_end && col < a.col_end;
    const int L = planes_of(a.limbs);
    const bool kara = is_k3(a.limbs);
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    if (in) {
        const int8_t* pa = a.planes + row * L * (int64_t)a.d_pad;
        const int8_t* pb = a.planes + col * L * (int64_t)a.d_pad;
        for (int k = 0; k < a.d_pad; k += 4) {
            int wa[kMaxLimbs], wb[kMaxLimbs];
            for (int l = 0; l < L; ++l) {
                wa[l] = *reinterpret_cast<const int*>(pa + (int64_t)l * a.d_pad + k);
                wb[l] = *reinterpret_cast<const int*>(pb + (int64_t)l * a.d_pad + k);
            }
            if (kara) {
                for (int l = 0; l < 3; ++l) acc[l] = (uint32_t)__builtin_amdgcn_sdot4(wa[l], wb[l], (int)acc[l], false);
            } else {
                for (int la = 0; la < L; ++la)
                    for (int lb = 0; lb < L; ++lb)
                        if (la + lb <= 3)
                            acc[la + lb] = (uint32_t)__builtin_amdgcn_sdot4(wa[la], wb[lb], (int)acc[la + lb], false);
            }
        }
    }
    const int32_t P = kara ? (int32_t)(acc[0] + ((acc[2] - acc[0] - acc[1]) << 7) + (acc[1] << 14))
                           : (int32_t)(acc[0] + (acc[1] << 8) + (acc[2] << 16) + (acc[3] << 24));
    if (MODE == 1) {
        if (in) a.dots[(row - a.row_begin) * (a.col_end - a.col_begin) + (col - a.col_begin)] = P;
    } else {
        bool keep = false;
        if (in) keep = keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
        emit_cell(a, keep, a.mirror_all != 0, (int32_t)row, (int32_t)col, P, lane);
    }
}

// ---------------------------------------------------------------------------------------------------
// A handful of rows against many columns (a search: q query sketches x the whole database), two base-256 limbs.
// The MFMA kernels fetch 64-byte k-slices of 256 rows per step -- a pattern that lives on L2 reuse in the all-vs-all case
// and has none here (1-256 rows x 10^6 columns: 3.1 ms for 4.1 GB of limb planes, 1.3 TB/s).  This kernel streams instead:
// the rows sit in LDS, a wave takes one column at a time and reads its limb rows front to back (1 KiB per load instruction,
// the re-check kernel's access pattern), every lane forms its part of all QT dots, and a transposing butterfly (QT - 1 +
// log2(64 / QT) exchanges instead of 6 QT) leaves the total of row q in the lanes whose upper bits spell q.  The entries of
// a two-limb set fit int16 (|v| <= 32639), so both sides are re-joined to int16 pairs (the rows once, in LDS; a column's
// chunk in registers, 8 instructions per 4 entries) and one v_dot2_i32_i16 does the work of four v_dot4_i32_i8 on limbs;
// its int32 accumulation wraps mod 2^32 exactly like the reference's int32 product.  QT = rows rounded up to a power of two,
// <= 16 (rows beyond the block are zeros in LDS).
// ---------------------------------------------------------------------------------------------------
// two base-256 limb dwords (4 entries: v = lo + 256 hi, lo and hi signed bytes, |v| <= 32639) -> the same entries as int16
// pairs {v0, v1}, {v2, v3}: low byte = lo, high byte = hi - (lo < 0) (a byte-wise subtraction without borrows between bytes)
__device__ __forceinline__ void limbs_to_i16(uint32_t lo, uint32_t hi, int& w01, int& w23) {
    const uint32_t neg = (lo >> 7) & 0x01010101u;
    const uint32_t hs = ((hi | 0x80808080u) - neg) ^ (~hi & 0x80808080u);
    w01 = (int)__builtin_amdgcn_perm(hs, lo, 0x05010400u);
    w23 = (int)__builtin_amdgcn_perm(hs, lo, 0x07030602u);
}

// c + a.lo * b.lo + a.hi * b.hi on int16 pairs, mod 2^32 (v_dot2_i32_i16).  The operands arrive as scalars on purpose:
// __builtin_bit_cast applied directly to a vector ELEMENT (bit_cast<v2s>(vec[e])) is folded to element 0 by hipcc 7.2 -- the
// unrolled loop below then multiplied the first dword of every 16-byte chunk four times (seen in the ISA; the same toolchain
// fault as in the filter epilogue's maximum, LABNOTES.md "Toolchain note").
__device__ __forceinline__ int dot2_i16(int a, int b, int c) {
    using v2s = __attribute__((ext_vector_type(2))) short;
    return __builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, a), __builtin_bit_cast(v2s, b), c, false);
}

template <int QT>
__global__ __launch_bounds__(512) void k_pairwise_skinny(const PairwiseArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    static_assert(QT >= 1 && QT <= 16 && (QT & (QT - 1)) == 0, "rows per pass");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = (int)(a.row_end - a.row_begin);
    const int64_t stride = 2 * (int64_t)a.d_pad;
    // the rows, as int16 pairs: plane A holds the pairs {v0, v1} of every entry quadruple, plane B the pairs {v2, v3} (any
    // pairing does, as long as both operands of a dot use the same one); a two-limb set has |v| <= 32639
    for (int64_t x = (int64_t)tid * 16; x < QT * (int64_t)a.d_pad; x += 512 * 16) {
        const int q = (int)(x / a.d_pad);
        const int64_t k = x - (int64_t)q * a.d_pad;
        v4i wa = v4i{0, 0, 0, 0}, wb = v4i{0, 0, 0, 0};
        if (q < nq) {
            const int8_t* ri = a.planes + (a.row_begin + q) * stride;
            const v4i lo = *reinterpret_cast<const v4i*>(ri + k), hi = *reinterpret_cast<const v4i*>(ri + a.d_pad + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int w01, w23;
                limbs_to_i16((uint32_t)lo[e], (uint32_t)hi[e], w01, w23);
                wa[e] = w01;
                wb[e] = w23;
            }
        }
        *reinterpret_cast<v4i*>(smem + q * stride + k) = wa;
        *reinterpret_cast<v4i*>(smem + q * stride + a.d_pad + k) = wb;
    }
    __syncthreads();
    constexpr int kShift = QT == 1 ? 6 : QT == 2 ? 5 : QT == 4 ? 4 : QT == 8 ? 3 : 2;   // 6 - log2(QT)
    const int my_q = (lane >> kShift) & (QT - 1);                 // the row whose total this lane ends up with
    const bool speaker = (lane & ((1 << kShift) - 1)) == 0 && my_q < nq;
    const int64_t n_waves = (int64_t)gridDim.x * 8, wid = (int64_t)blockIdx.x * 8 + wave;
    for (int64_t col = a.col_begin + wid; col < a.col_end; col += n_waves) {   // wave-uniform
        const int8_t* rj = a.planes + col * stride;
        int acc[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) acc[q] = 0;
        for (int k = lane * 16; k < a.d_pad; k += 1024) {
            const v4i lj = *reinterpret_cast<const v4i*>(rj + k);
            const v4i hj = *reinterpret_cast<const v4i*>(rj + a.d_pad + k);
            v4i ja, jb;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int w01, w23;
                limbs_to_i16((uint32_t)lj[e], (uint32_t)hj[e], w01, w23);
                ja[e] = w01;
                jb[e] = w23;
            }
#pragma unroll
            for (int q = 0; q < QT; ++q) {
                const v4i ia = *reinterpret_cast<const v4i*>(smem + q * stride + k);
                const v4i ib = *reinterpret_cast<const v4i*>(smem + q * stride + a.d_pad + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {                       // int32 accumulation wraps mod 2^32, as the reference's product does
                    acc[q] = dot2_i16(ia[e], ja[e], acc[q]);
                    acc[q] = dot2_i16(ib[e], jb[e], acc[q]);
                }
            }
        }
        uint32_t part[QT];
#pragma unroll
        for (int q = 0; q < QT; ++q) part[q] = (uint32_t)acc[q];
        // halve the rows a lane is responsible for while doubling the lanes behind each value
        int m = 32;
#pragma unroll
        for (int n = QT; n > 1; n >>= 1, m >>= 1) {
            const bool upper = (lane & m) != 0;
#pragma unroll
            for (int i = 0; i < n / 2; ++i) {
                const uint32_t send = upper ? part[i] : part[i + n / 2];
                const uint32_t keepv = upper ? part[i + n / 2] : part[i];
                part[i] = keepv + (uint32_t)__shfl_xor((int)send, m, 64);
            }
        }
#pragma unroll
        for (; m >= 1; m >>= 1) part[0] += (uint32_t)__shfl_xor((int)part[0], m, 64);
        const int32_t P = (int32_t)part[0];
        const int32_t row = (int32_t)a.row_begin + my_q;
        bool keep = false;
        if (speaker) keep = keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
        emit_cell(a, keep, a.mirror_all != 0, row, (int32_t)col, P, lane);
    }
}

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
// at radix_keeps_high_limb.  A workgroup takes 16 rows (one KiB of the fragment-major plane holds 16 rows x 64 k) x 256 k: a lane
// takes 16 consecutive k of one row, reads its 16 coarse bytes where k_recode_rows put them and 16 bytes of the wire.  (One
// workgroup per 16 rows looping over k moved 3.4 TB/s: eight dependent rounds of loads per workgroup.)
// need != NULL: only the rows marked there (k_rows_needed: what a plan's re-check and flagged tiles will read).
__global__ __launch_bounds__(256) void k_planes_from_wire(const int8_t* __restrict__ lo_wire, const int8_t* __restrict__ coarse_fm,
                                                          const CoarseRow* __restrict__ rows, int64_t count, int d_pad,
                                                          int8_t* __restrict__ planes, const unsigned char* __restrict__ need) {
    const int nk = d_pad / 64;
    const int64_t grp = blockIdx.x;                        // 16 rows
    const int chunks = 16 * (d_pad / 16);                  // (row, 16-entry chunk) pairs of the group
    {
        const int idx = (int)blockIdx.y * 256 + (int)threadIdx.x;
        if (idx >= chunks) return;
        const int r = idx & 15, kc = idx >> 4;             // consecutive lanes: the 16 rows of one chunk = 256 contiguous bytes of the plane
        const int64_t row = grp * 16 + r;
        if (row >= count || (need && need[row] == 0)) return;
        const int m = rows[row].radix;
        const v4i c4 = *reinterpret_cast<const v4i*>(coarse_fm + (grp * nk + (kc >> 2)) * 1024 + (((kc & 3) << 4) + r) * 16);
        const v4i l4 = *reinterpret_cast<const v4i*>(lo_wire + row * (int64_t)d_pad + kc * 16);
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
                const int v = t + (int)(int8_t)(l0 - t);
                ph |= (uint32_t)(uint8_t)(int8_t)((v - l0) >> 8) << (8 * e);
            }
            h4[w] = (int)ph;
        }
        *reinterpret_cast<v4i*>(planes + row * 2 * (int64_t)d_pad + kc * 16) = l4;
        *reinterpret_cast<v4i*>(planes + (row * 2 + 1) * (int64_t)d_pad + kc * 16) = h4;
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

// Exact re-check of the candidate pairs.  A wave takes B pairs per round (B = 64 unless stated).  QUAD = 0: all
// 64 lanes stream the limb rows of one pair after the other (1 KiB per load instruction); QUAD = 1: each
// quarter of the wave streams one pair, four pairs in flight.  Lanes 0..B-1 then run the keep test, the
// quantisation and the append for the round's pairs.
template <int B, int QUAD>
__global__ __launch_bounds__(256) void k_exact_pairs(const PairwiseArgs a) {
    static_assert(B == 16 || B == 32 || B == 64, "pairs per round");
    const int lane = threadIdx.x & 63;
    unsigned long long n_cand = *a.cand_counter;
    if (n_cand > a.cand_limit) return;   // the caller will run the exact kernel instead
    if (n_cand > a.cand_capacity) n_cand = a.cand_capacity;
    // The list is in tile order (the filter appends wave by wave), so neighbours in the list share rows.  Each XCD
    // label (blockIdx.x % 8: the workgroups that share an L2) takes one contiguous eighth of the list and its waves
    // stride over that: the limb rows of a tile's pairs are then fetched once per L2 instead of once per pair
    // (round-robin over all waves of the chip: 20 % L2 hit rate, 8.4 GB of fabric reads for 1.3 M pairs).
    const unsigned long long per = ((n_cand + 7) / 8 + B - 1) / B * B;          // candidates per XCD label, multiple of B
    const unsigned long long first = (unsigned long long)(blockIdx.x & 7) * per;
    const unsigned long long last = first + per < n_cand ? first + per : n_cand;
    const unsigned long long waves = (unsigned long long)(gridDim.x >> 3) * 4;  // gridDim.x is a multiple of 8
    const unsigned long long wid = (unsigned long long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    const int64_t stride = 2 * (int64_t)a.d_pad;
    const int width = QUAD ? 16 : 64;                 // lanes per pair
    const int sub = QUAD ? lane >> 4 : 0, sl = QUAD ? lane & 15 : lane;
    for (unsigned long long base = first + wid * B; base < last; base += waves * B) {
        const unsigned long long mine = base + lane;
        const bool have = lane < B && mine < last;
        int2 pr = make_int2(0, 0);
        if (have) pr = a.cand[mine];
        const int cnt = (int)(last - base < (unsigned long long)B ? last - base : (unsigned long long)B);
        int32_t P_mine = 0;
        for (int step = 0; step * (QUAD ? 4 : 1) < cnt; ++step) {
            const int q = QUAD ? step * 4 + sub : step;   // pair handled by these lanes in this step
            const int row = __shfl(pr.x, q, 64);          // beyond the round's count: (0, 0), harmless
            const int col = __shfl(pr.y, q, 64) & 0x7fffffff;
            const int8_t* ri = a.planes + (int64_t)row * stride;
            const int8_t* rj = a.planes + (int64_t)col * stride;
            int acc0 = 0, acc1 = 0, acc2 = 0;
            for (int k = sl * 16; k < a.d_pad; k += width * 16) {
                const v4i li = *reinterpret_cast<const v4i*>(ri + k);
                const v4i hi = *reinterpret_cast<const v4i*>(ri + a.d_pad + k);
                const v4i lj = *reinterpret_cast<const v4i*>(rj + k);
                const v4i hj = *reinterpret_cast<const v4i*>(rj + a.d_pad + k);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0 = __builtin_amdgcn_sdot4(li[e], lj[e], acc0, false);
                    acc1 = __builtin_amdgcn_sdot4(li[e], hj[e], acc1, false);
                    acc1 = __builtin_amdgcn_sdot4(hi[e], lj[e], acc1, false);
                    acc2 = __builtin_amdgcn_sdot4(hi[e], hj[e], acc2, false);
                }
            }
            uint32_t P = (uint32_t)acc0 + ((uint32_t)acc1 << 8) + ((uint32_t)acc2 << 16);
#pragma unroll
            for (int o = width / 2; o > 0; o >>= 1) P += __shfl_xor(P, o, 64);
            if (QUAD) {
                const uint32_t got = __shfl(P, (lane & 3) * 16, 64);     // pair step*4 + (lane&3) -> lane
                if ((lane >> 2) == step) P_mine = (int32_t)got;          // lanes 0..15: lane == its pair
            } else if (lane == step) {
                P_mine = (int32_t)P;
            }
        }
        bool keep = false;
        const int32_t row = pr.x, col = pr.y & 0x7fffffff;
        if (have) keep = keep_cell(P_mine, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
        emit_cell(a, keep, pr.y < 0, row, col, P_mine, lane);
    }
}

// Re-check, tree form.  k_exact_pairs above sums every pair's 64 per-lane partial dots with a 6-step shuffle butterfly
// (384 dependent shuffles per round of 64 pairs).  Here a round is a binary tree over its pairs: two sub-results are
// merged with ONE exchange -- the lanes whose bit `level` is clear keep the first half's running sums and receive the
// partner's, the others keep the second half's -- so after six levels lane l holds the complete dot of pair l with 63
// exchanges per round instead of 384, and the loads of consecutive pairs do not wait on any reduction.
struct PairDots {
    const PairwiseArgs& a;
    int2 pr;          // this lane's candidate of the round (lane = pair index)
    int cnt;          // pairs in the round
    int lane;
    int next = 0;
    __device__ __forceinline__ uint32_t partial() {          // per-lane partial dot of pair `next` (0 beyond the round)
        const int q = next++;
        if (q >= cnt) return 0u;                              // wave-uniform
        const int row = __shfl(pr.x, q, 64), col = __shfl(pr.y, q, 64) & 0x7fffffff;
        const int64_t stride = 2 * (int64_t)a.d_pad;
        const int8_t* ri = a.planes + (int64_t)row * stride;
        const int8_t* rj = a.planes + (int64_t)col * stride;
        int acc0 = 0, acc1 = 0, acc2 = 0;
        for (int k = lane * 16; k < a.d_pad; k += 1024) {
            const v4i li = *reinterpret_cast<const v4i*>(ri + k);
            const v4i hi = *reinterpret_cast<const v4i*>(ri + a.d_pad + k);
            const v4i lj = *reinterpret_cast<const v4i*>(rj + k);
            const v4i hj = *reinterpret_cast<const v4i*>(rj + a.d_pad + k);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                acc0 = __builtin_amdgcn_sdot4(li[e], lj[e], acc0, false);
                acc1 = __builtin_amdgcn_sdot4(li[e], hj[e], acc1, false);
                acc1 = __builtin_amdgcn_sdot4(hi[e], lj[e], acc1, false);
                acc2 = __builtin_amdgcn_sdot4(hi[e], hj[e], acc2, false);
            }
        }
        return (uint32_t)acc0 + ((uint32_t)acc1 << 8) + ((uint32_t)acc2 << 16);
    }
    // Four consecutive pairs at once: ONE loop over k that loads all four pairs' limb rows before any of them is
    // consumed -- 16 loads of 16 bytes in flight per lane, written out so that it does not depend on how the compiler
    // feels about interleaving four copies of partial() (it did in round 2, 132 registers, and stopped doing so after an
    // unrelated edit of the caller's loop: 70 registers, the pairs one after the other, 0.80 -> 1.00 ms on 1.26 M pairs).
    // Returns the level-2 node of the tree (the four partial dots merged with three exchanges).  Pairs beyond the
    // round's count read rows 0 / 0 (their lanes' pr is zero): harmless, their results are never used.
    __device__ __forceinline__ uint32_t quad() {
        const int q0 = next;
        next += 4;
        if (q0 >= cnt) return 0u;                             // wave-uniform
        const int64_t stride = 2 * (int64_t)a.d_pad;
        const int8_t* ri[4];
        const int8_t* rj[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int row = __shfl(pr.x, q0 + p, 64), col = __shfl(pr.y, q0 + p, 64) & 0x7fffffff;
            ri[p] = a.planes + (int64_t)row * stride;
            rj[p] = a.planes + (int64_t)col * stride;
        }
        int acc0[4] = {0, 0, 0, 0}, acc1[4] = {0, 0, 0, 0}, acc2[4] = {0, 0, 0, 0};
        for (int k = lane * 16; k < a.d_pad; k += 1024) {
            v4i li[4], hi[4], lj[4], hj[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                li[p] = *reinterpret_cast<const v4i*>(ri[p] + k);
                hi[p] = *reinterpret_cast<const v4i*>(ri[p] + a.d_pad + k);
                lj[p] = *reinterpret_cast<const v4i*>(rj[p] + k);
                hj[p] = *reinterpret_cast<const v4i*>(rj[p] + a.d_pad + k);
            }
#pragma unroll
            for (int p = 0; p < 4; ++p)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    acc0[p] = __builtin_amdgcn_sdot4(li[p][e], lj[p][e], acc0[p], false);
                    acc1[p] = __builtin_amdgcn_sdot4(li[p][e], hj[p][e], acc1[p], false);
                    acc1[p] = __builtin_amdgcn_sdot4(hi[p][e], lj[p][e], acc1[p], false);
                    acc2[p] = __builtin_amdgcn_sdot4(hi[p][e], hj[p][e], acc2[p], false);
                }
        }
        uint32_t part[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) part[p] = (uint32_t)acc0[p] + ((uint32_t)acc1[p] << 8) + ((uint32_t)acc2[p] << 16);
        const bool up0 = lane & 1, up1 = (lane >> 1) & 1;
        const uint32_t m01 = (up0 ? part[1] : part[0]) + (uint32_t)__shfl_xor((int)(up0 ? part[0] : part[1]), 1, 64);
        const uint32_t m23 = (up0 ? part[3] : part[2]) + (uint32_t)__shfl_xor((int)(up0 ? part[2] : part[3]), 1, 64);
        return (up1 ? m23 : m01) + (uint32_t)__shfl_xor((int)(up1 ? m01 : m23), 2, 64);
    }
    template <int LEVEL>
    __device__ __forceinline__ uint32_t tree() {              // sums of 2^LEVEL consecutive pairs, spread over the lanes
        if constexpr (LEVEL == 2) {
            return quad();
        } else {
            static_assert(LEVEL > 2, "the tree's leaves are quads");
            if (next >= cnt) {                                // nothing left in this subtree (wave-uniform)
                next += 1 << LEVEL;
                return 0u;
            }
            const uint32_t first = tree<LEVEL - 1>();
            const uint32_t second = tree<LEVEL - 1>();
            const bool upper = (lane >> (LEVEL - 1)) & 1;
            const uint32_t got = (uint32_t)__shfl_xor((int)(upper ? first : second), 1 << (LEVEL - 1), 64);
            return (upper ? second : first) + got;
        }
    }
};

template <int RMODE>
__global__ __launch_bounds__(256) void k_exact_pairs_tree(const PairwiseArgs a) {
    const int lane = threadIdx.x & 63;
    unsigned long long n_cand = *a.cand_counter;
    if (n_cand > a.cand_limit) return;
    if (n_cand > a.cand_capacity) n_cand = a.cand_capacity;
    // Work split: the list is cut into chunks of 64 rounds (4096 pairs) and chunk c belongs to XCD label c % 8
    // (blockIdx.x % 8: the workgroups that share an L2).  A chunk keeps neighbours of the list -- pairs of the same tile,
    // which share rows -- in one L2.  Within an XCD the rounds are handed out by a counter, not by a fixed stride: the
    // pairs of the diagonal tiles' clusters are cheap L2 hits, a chance pair costs two rows from HBM, and where the list
    // holds one kind after the other (the filter's waves leave their few chance candidates in regions that are
    // gathered behind the others) a fixed assignment left the HBM-bound rounds to a fifth of the waves: 0.80 -> 1.07 ms.
    // The next index is drawn before the current round is worked on, so the atomic's latency hides behind the loads.
    const unsigned long long rounds = (n_cand + 63) / 64;
    const unsigned long long xcd = blockIdx.x & 7;
    const unsigned long long last = n_cand;
    // recheck_mode 1 (default): every wave's first round is its own index, later rounds come from the XCD's counter (same-
    // address atomics cost ~20 ns each, so a wave does not ask just to learn that nothing is left when the grid already
    // covers every round); 2: every round from the counter; 0: fixed stride over the XCD's rounds; 3: one contiguous eighth
    // of the list per XCD, fixed stride (round 2's split).
    unsigned long long* queue = a.recheck_queue + xcd * 8;     // one counter per XCD, 64 bytes apart
    const unsigned long long waves = (unsigned long long)(gridDim.x >> 3) * 4;
    const unsigned long long wid = (unsigned long long)(blockIdx.x >> 3) * 4 + (threadIdx.x >> 6);
    constexpr int mode = RMODE;
    const unsigned long long per_xcd = (rounds + 7) / 8;      // mode 3
    const bool more = mode == 2 || (mode == 1 && ((rounds + 63) / 64 + 7) / 8 * 64 > waves);
    auto round_of = [&](unsigned long long i) -> unsigned long long {
        if (mode == 3) return i < per_xcd ? xcd * per_xcd + i : ~0ULL;
        return ((i >> 6) * 8 + xcd) * 64 + (i & 63);
    };
    unsigned long long fixed = wid;                             // modes 0 / 3: the wave's next index
    auto draw = [&]() -> unsigned long long {
        if (mode == 0 || mode == 3) {
            fixed += waves;
            return round_of(fixed);
        }
        if (!more) return ~0ULL;
        unsigned long long i = 0;
        if (lane == 0) i = atomicAdd(queue, 1ULL);
        i = (unsigned long long)__shfl((long long)i, 0, 64);
        return round_of(i + (mode == 1 ? waves : 0ULL));
    };
    for (unsigned long long next = mode == 2 ? draw() : round_of(wid);;) {
        const unsigned long long round = next;
        if (round >= rounds) break;                            // indices only grow: every later one is beyond too
        next = draw();
        const unsigned long long base = round * 64;
        const unsigned long long mine = base + lane;
        const bool have = mine < last;
        int2 pr = make_int2(0, 0);
        if (have) pr = a.cand[mine];
        const int cnt = (int)(last - base < 64ULL ? last - base : 64ULL);
        PairDots dots{a, pr, cnt, lane};
        const int32_t P = (int32_t)dots.tree<6>();
        bool keep = false;
        const int32_t row = pr.x, col = pr.y & 0x7fffffff;
        if (have) keep = keep_cell(P, a.d, a.norms_sq[row], a.norms_sq[col], a.keep_mode, a.keep_coeff);
        emit_cell(a, keep, pr.y < 0, row, col, P, lane);
    }
}

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
                                                    const int2* __restrict__ ends, const DenseSizes out) {
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
            col[base + i] = cv;
            q[base + i] = (uint8_t)qv;
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

// Kernel variant (MVS_PAIRWISE_VARIANT):
//   0 -> 8 waves 2x4, wave tile 64x32, tile 128x128, 4-stage ring          (double-buffered fragments)
//   1 -> 4 waves 2x2, wave tile 64x32, tile 128x64,  3-stage ring, two workgroups per CU
//   2 -> 8 waves 2x4, wave tile 64x32, tile 128x128, 5-stage ring
//   3 -> 4 waves 2x2, wave tile 64x32, tile 128x64,  2-stage ring
//   4 -> 8 waves 4x2, wave tile 64x64, tile 256x128, 3-stage ring          (fewer LDS/L2 bytes per MFMA)
//   5 -> 8 waves 2x4, wave tile 64x64, tile 128x256, 3-stage ring
//   6 -> variant 0 on the 16x16x64 MFMA shape (two base-256 limbs only; other limb codes use variant 0) [default]
inline int pairwise_variant(const Options& opt) {
    return (opt.pairwise_variant < 0 || opt.pairwise_variant > 9) ? 8 : opt.pairwise_variant;
}

// A block whose tile grid is less than one patch (16 tile rows) high and not under the symmetric schedule: the XCDs split
// the patch by columns (map_tile mode 3).  The default map's 4 x 8 sub-patches split the ROWS of a patch over the XCDs: with
// <= 4 tile rows 2 of the 8 XCDs have tiles (filter pass, queries x 10^6 samples: 1024 queries 6.9 -> 1.9 ms, 2048 queries
// 6.9 -> 3.6 ms).  From 16 tile rows on the default map uses every XCD and reuses panels better (4096 queries: 7.0 ms
// against 7.3 ms with mode 3).
inline void skinny_map(PairwiseArgs& b, int n_tr) {
    if (!b.symmetric && n_tr < 16 && b.map_mode == 0) b.map_mode = 3;
}

template <int L, bool KARA, int MODE, int NST, int WM, int WN, int BT, int AT = 2, bool DBUF = (BT == 1), int ABL = 0>
int launch_mfma_variant(hipStream_t stream, const PairwiseArgs& a) {
    constexpr int TM = WM * AT * 32, TN = WN * BT * 32;
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return 0;
    const int n_tr = (int)((rows + TM - 1) / TM), n_tc = (int)((cols + TN - 1) / TN);
    const int n_spr = (n_tr + 15) / 16, n_spc = (n_tc + 15) / 16;
    if (n_spr > 65535 || (int64_t)n_spc * 256 * (WM * WN * 64) > 0xffffffffLL) return MVS_E_INVALID;
    const size_t lds = (size_t)NST * L * (TM + TN) * kSK;
    PairwiseArgs b = a;
    // the symmetric schedule needs the row and column tile grids to share their origin modulo TM
    if (b.symmetric && ((a.row_begin - a.col_begin) % TM != 0 || TM % TN != 0 || a.mirror_all)) b.symmetric = 0;
    skinny_map(b, n_tr);
    hipError_t e = hipFuncSetAttribute(
        reinterpret_cast<const void*>(&k_pairwise_mfma<L, KARA, MODE, NST, WM, WN, BT, AT, DBUF, ABL>),
        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_pairwise_mfma<L, KARA, MODE, NST, WM, WN, BT, AT, DBUF, ABL>), dim3((unsigned)n_spc * 256u, (unsigned)n_spr),
                       dim3(WM * WN * 64), lds, stream, b, n_tr, n_tc, n_spc);
    return 0;
}

template <int MODE>
int launch_mfma16(hipStream_t stream, const PairwiseArgs& a) {
    constexpr int TM = 128, TN = 128, NST = 4;
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return 0;
    const int n_tr = (int)((rows + TM - 1) / TM), n_tc = (int)((cols + TN - 1) / TN);
    const int n_spr = (n_tr + 15) / 16, n_spc = (n_tc + 15) / 16;
    if (n_spr > 65535 || (int64_t)n_spc * 256 * 512 > 0xffffffffLL) return MVS_E_INVALID;
    const size_t lds = (size_t)NST * 2 * (TM + TN) * kSK;
    PairwiseArgs b = a;
    if (b.symmetric && ((a.row_begin - a.col_begin) % TM != 0 || a.mirror_all)) b.symmetric = 0;
    skinny_map(b, n_tr);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_mfma16<MODE, NST>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_pairwise_mfma16<MODE, NST>), dim3((unsigned)n_spc * 256u, (unsigned)n_spr), dim3(512), lds,
                       stream, b, n_tr, n_tc, n_spc);
    return 0;
}

// may the ping-pong kernel take its B operand straight from the fragment-major plane (k_pairwise_pp<.., BD = 1>)?
static bool pp_direct_b(const PairwiseArgs& a, int mode, const Options& opt) {
    return opt.pairwise_bdirect != 0 && (mode == 2 ? a.coarse_fm != nullptr : a.planes_fm != nullptr) &&
           ((a.row_begin | a.col_begin) & 15) == 0;
}

template <int MODE, int NST, int ORDER = 0, int ABL = 0, int PH = 1, int NT = 0, int BD = 0>
int launch_pp(hipStream_t stream, const PairwiseArgs& a) {
    using G = PpGeom<MODE>;
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return 0;
    const int n_tr = (int)((rows + G::TM - 1) / G::TM), n_tc = (int)((cols + G::TN - 1) / G::TN);
    const int n_spr = (n_tr + 15) / 16, n_spc = (n_tc + 15) / 16;
    if (n_spr > 65535 || (int64_t)n_spc * 256 * 512 > 0xffffffffLL) return MVS_E_INVALID;
    const size_t lds = (size_t)NST * G::kStage;
    PairwiseArgs b = a;
    if (b.symmetric && ((a.row_begin - a.col_begin) % G::TM != 0 || a.mirror_all)) b.symmetric = 0;
    skinny_map(b, n_tr);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_pp<MODE, NST, ORDER, ABL, PH, NT, BD>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_pairwise_pp<MODE, NST, ORDER, ABL, PH, NT, BD>), dim3((unsigned)n_spc * 256u, (unsigned)n_spr), dim3(512), lds, stream,
                       b, n_tr, n_tc, PlanSegs{});
    return 0;
}

// the default ping-pong kernel (4-stage ring), with the B operand straight from the fragment-major plane when that exists
template <int MODE>
int launch_pp_default(hipStream_t stream, const PairwiseArgs& a, const Options& opt) {
    if (pp_direct_b(a, MODE, opt)) return launch_pp<MODE, 4, 0, 0, 1, 0, 1>(stream, a);
    return launch_pp<MODE, 4>(stream, a);
}

template <int L, bool KARA, int MODE>
int launch_mfma(hipStream_t stream, const PairwiseArgs& a, const Options& opt) {
    if constexpr (L == 2 && !KARA) {
        if (pairwise_variant(opt) == 7) return launch_pp<MODE, 5>(stream, a);
        if (pairwise_variant(opt) == 8) return launch_pp_default<MODE>(stream, a, opt);
        if (pairwise_variant(opt) == 9) return launch_pp<MODE, 4, 0, 0, 2>(stream, a);
        if (pairwise_variant(opt) == 6) return launch_mfma16<MODE>(stream, a);
    }
    if constexpr (KARA) {   // 48 KB per 128x128 stage: at most three stages fit the 160 KB of LDS
        switch (pairwise_variant(opt)) {
            case 1: return launch_mfma_variant<L, KARA, MODE, 3, 2, 2, 1>(stream, a);
            case 3: return launch_mfma_variant<L, KARA, MODE, 2, 2, 2, 1>(stream, a);
            case 2: return launch_mfma_variant<L, KARA, MODE, 2, 2, 4, 1>(stream, a);
            default: return launch_mfma_variant<L, KARA, MODE, 3, 2, 4, 1>(stream, a);
        }
    } else {
        switch (pairwise_variant(opt)) {
            case 1: return launch_mfma_variant<L, KARA, MODE, 3, 2, 2, 1>(stream, a);
            case 2: return launch_mfma_variant<L, KARA, MODE, 5, 2, 4, 1>(stream, a);
            case 3: return launch_mfma_variant<L, KARA, MODE, 2, 2, 2, 1>(stream, a);
            case 4: return launch_mfma_variant<L, KARA, MODE, 3, 4, 2, 2>(stream, a);
            case 5: return launch_mfma_variant<L, KARA, MODE, 3, 2, 4, 2>(stream, a);
            default: return launch_mfma_variant<L, KARA, MODE, 4, 2, 4, 1>(stream, a);
        }
    }
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

// rows [0, count) (count a multiple of 16; all pointers at the range's first row, which is a multiple of 16)
int launch_planes_from_wire(hipStream_t stream, const int8_t* d_lo_wire, const int8_t* d_coarse_fm, const CoarseRow* d_rows,
                            int64_t count, int d_pad, int8_t* d_planes, const unsigned char* d_need) {
    if (count <= 0) return 0;
    hipLaunchKernelGGL(k_planes_from_wire, dim3((unsigned)(count / 16), (unsigned)((d_pad / 16 * 16 + 255) / 256)), dim3(256), 0, stream,
                       d_lo_wire, d_coarse_fm, d_rows, count, d_pad, d_planes, d_need);
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

static int filter_variant_for(const PairwiseArgs& a, const Options& opt);
static int launch_search_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt);

int launch_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt) {
    if (a.limbs != 2 || a.d_pad > 32768) return MVS_E_INVALID;
    // opt.filter_variant: tile shape / ring depth of the one-pass filter.
    // Default (-1): the ping-pong kernel on 256 x 256 tiles (half the L2 -> LDS bytes per cell of 128 x 128 tiles; its
    // two wave groups overlap copies and MFMAs: 11.4 -> 10.0 ms at 100k samples against the ring kernel on the same
    // tiles) from 128 tiles on (filter_variant_for), 128 x 128 ring tiles for the small blocks below that.
    const int v = filter_variant_for(a, opt);
    switch (v) {
        case 50: return launch_search_filter(stream, a, opt);
        case 7: return launch_pp<2, 5>(stream, a);   // ping-pong wave groups, 256 x 256, 5-stage ring (all 160 KiB of LDS)
        case 8: return launch_pp_default<2>(stream, a, opt);   // the same on a 4-stage ring (B operand direct when the fragment-major plane exists)
        case 9: return launch_pp<2, 4, 0, 0, 2>(stream, a);    // two phases per slice
        case 10: return launch_pp<2, 4, 2, 0, 2>(stream, a);   // two phases, copy / read order by wave parity
        case 40: return launch_pp<2, 4, 0, 0, 1, 1>(stream, a);   // variant 8 with non-temporal column-panel copies
        case 41: return launch_pp<2, 4, 0, 0, 1, 2>(stream, a);   // ... non-temporal row-panel copies
        case 42: return launch_pp<2, 4, 0, 0, 1, 3>(stream, a);   // ... both
#ifdef MVS_ABLATIONS
        case 31: return launch_pp<2, 4, 0, 1>(stream, a);
        case 32: return launch_pp<2, 4, 0, 2>(stream, a);
        case 33: return launch_pp<2, 4, 0, 3>(stream, a);
#endif
        case 1: return launch_mfma_variant<1, false, 2, 4, 2, 4, 2, 4, true>(stream, a);   // 256 x 256, waves 128 x 64
        case 3: return launch_mfma_variant<1, false, 2, 4, 4, 2, 2, 2, true>(stream, a);   // 256 x 128, waves 64 x 64
#ifdef MVS_ABLATIONS   // k-loop ablations (results are garbage): 1x no MFMA, x2 no copies, x3 no fragment reads
        case 11: return launch_mfma_variant<1, false, 2, 4, 2, 4, 1, 2, true, 1>(stream, a);   // of variant 0
        case 12: return launch_mfma_variant<1, false, 2, 4, 2, 4, 1, 2, true, 2>(stream, a);
        case 13: return launch_mfma_variant<1, false, 2, 4, 2, 4, 1, 2, true, 3>(stream, a);
        case 21: return launch_mfma_variant<1, false, 2, 4, 2, 4, 2, 4, true, 1>(stream, a);   // of variant 1
        case 22: return launch_mfma_variant<1, false, 2, 4, 2, 4, 2, 4, true, 2>(stream, a);
        case 23: return launch_mfma_variant<1, false, 2, 4, 2, 4, 2, 4, true, 3>(stream, a);
#endif
        case 5: return launch_mfma_variant<1, false, 2, 5, 2, 4, 1>(stream, a);            // 5-stage ring
        case 6: return launch_mfma_variant<1, false, 2, 3, 2, 4, 1>(stream, a);            // 3-stage ring, 3 workgroups / CU
        default: return launch_mfma_variant<1, false, 2, 4, 2, 4, 1>(stream, a);           // 128 x 128, waves 64 x 32
    }
}

// ---- block plans: several rectangles of 256 x 256 filter tiles in ONE launch of the ping-pong filter ----
// blocks[k] = {row_begin, row_end, col_begin, col_end}, origins on multiples of 256 rows / columns (the caller checks);
// fills segs and returns the 1-D grid size in workgroups, 0 if the plan is empty, -1 if it does not fit one launch
long long plan_segments(const int64_t (*blocks)[4], int n, PlanSegs* segs) {
    *segs = PlanSegs{};
    if (n > kPlanSegs) return -1;
    unsigned long long wg = 0;
    int m = 0;
    for (int k = 0; k < n; ++k) {
        const int64_t rows = blocks[k][1] - blocks[k][0], cols = blocks[k][3] - blocks[k][2];
        if (rows <= 0 || cols <= 0) continue;
        const int n_tr = (int)((rows + 255) / 256), n_tc = (int)((cols + 255) / 256);
        const int n_spr = (n_tr + 15) / 16, n_spc = (n_tc + 15) / 16;
        segs->wg_begin[m] = (unsigned)wg;
        segs->n_tr[m] = n_tr;
        segs->n_tc[m] = n_tc;
        segs->n_spc[m] = n_spc;
        segs->i_begin[m] = blocks[k][0];
        segs->j_begin[m] = blocks[k][2];
        wg += (unsigned long long)n_spr * (unsigned long long)n_spc * 256ull;
        if (wg * 512ull > 0xffffffffull) return -1;             // a dispatch holds at most 2^32 work-items per dimension
        ++m;
    }
    segs->n = m;
    for (int k = m; k <= kPlanSegs; ++k) segs->wg_begin[k] = (unsigned)wg;
    return (long long)wg;
}

// The tiles a plan launch computes, dealt out evenly: super-patch by super-patch (segment, patch row, patch column -- the order
// the static map's workgroups are dispatched in), the valid tiles of a super-patch in sub-patch order (4 x 8 tiles: 12 operand
// panels per 32 tiles) are cut into eight contiguous runs, one per XCD label, of equal length but for a remainder that rotates
// from super-patch to super-patch.  A tile is valid by the kernel's own rule: inside the segment's grid and not strictly below
// the diagonal of the symmetric square.
bool plan_tile_order(const PairwiseArgs& a, const PlanSegs& segs, std::vector<unsigned>* order, unsigned* per) {
    std::vector<unsigned> lists[8];
    std::vector<unsigned> patch;
    unsigned long long total = 0;
    unsigned rot = 0;
    for (int sg = 0; sg < segs.n; ++sg) {
        const int n_tr = segs.n_tr[sg], n_tc = segs.n_tc[sg];
        if (n_tr > 0x3fff || n_tc > 0x3fff || sg > 15) return false;
        total += (unsigned long long)n_tr * (unsigned long long)n_tc;
        if (total > (1ull << 20)) return false;
        const int n_spr = (n_tr + 15) / 16, n_spc = (n_tc + 15) / 16;
        for (int spr = 0; spr < n_spr; ++spr)
            for (int spc = 0; spc < n_spc; ++spc) {
                patch.clear();
                for (int sub = 0; sub < 8; ++sub)
                    for (int ql = 0; ql < 32; ++ql) {
                        int tr, tc;
                        if (a.map_mode == 1) {          // (the sub-patch shapes of map_tile)
                            tr = spr * 16 + (sub >> 2) * 8 + (ql >> 2);
                            tc = spc * 16 + (sub & 3) * 4 + (ql & 3);
                        } else if (a.map_mode == 2) {
                            tr = spr * 16 + sub * 2 + (ql >> 4);
                            tc = spc * 16 + (ql & 15);
                        } else {
                            tr = spr * 16 + (sub >> 1) * 4 + (ql >> 3);
                            tc = spc * 16 + (sub & 1) * 8 + (ql & 7);
                        }
                        if (tr >= n_tr || tc >= n_tc) continue;
                        const long long i0 = segs.i_begin[sg] + (long long)tr * 256, j0 = segs.j_begin[sg] + (long long)tc * 256;
                        if (a.symmetric && j0 >= a.sym_begin && j0 + 256 <= i0) continue;      // k_pairwise_pp: produced by mirroring
                        patch.push_back((unsigned)sg << 28 | (unsigned)tr << 14 | (unsigned)tc);
                    }
                const unsigned n = (unsigned)patch.size(), base = n / 8, extra = n % 8;
                unsigned at = 0;
                for (unsigned j = 0; j < 8; ++j) {
                    const unsigned x = (rot + j) & 7u, take = base + (j < extra ? 1u : 0u);
                    lists[x].insert(lists[x].end(), patch.begin() + at, patch.begin() + at + take);
                    at += take;
                }
                rot = (rot + extra) & 7u;
            }
    }
    size_t longest = 0;
    for (const auto& l : lists) longest = std::max(longest, l.size());
    if (longest == 0) return false;
    order->assign(longest * 8, ~0u);
    for (int x = 0; x < 8; ++x) std::copy(lists[x].begin(), lists[x].end(), order->begin() + (size_t)x * longest);
    *per = (unsigned)longest;
    return true;
}

// the ping-pong filter (4-stage ring, B operand direct) over the segments of a plan; `a` carries the frame (PairwiseArgs::plan)
int launch_filter_plan(hipStream_t stream, const PairwiseArgs& a, const PlanSegs& segs, long long workgroups) {
    if (a.limbs != 2 || a.d_pad > 32768 || !a.plan || a.coarse_fm == nullptr || ((a.row_begin | a.col_begin) & 255) != 0) return MVS_E_INVALID;
    if (workgroups <= 0 || segs.n <= 0) return 0;
    if (segs.order != nullptr) workgroups = (long long)segs.order_per * 8;
    using G = PpGeom<2>;
    const size_t lds = (size_t)4 * G::kStage;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_pp<2, 4, 0, 0, 1, 0, 1>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_pairwise_pp<2, 4, 0, 0, 1, 0, 1>), dim3((unsigned)workgroups), dim3(512), lds, stream, a, 0, 0, segs);
    return 0;
}

// row blocks of 16 the streaming search filter keeps resident for this sketch length (0: the rows do not fit the LDS)
static int search_filter_rb(const PairwiseArgs& a) {
    for (int rb : {4, 2, 1})
        if ((size_t)16 * rb * ((size_t)a.d_pad + search_row_pad(a.d_pad)) + (size_t)16 * rb * sizeof(float4) <= (size_t)150 * 1024) return rb;
    return 0;
}

template <int RB, int NB>
static int launch_search_filter_rb(hipStream_t stream, const PairwiseArgs& a) {
    const int64_t rows = a.row_end - a.row_begin;
    const int groups = (int)((rows + 16 * RB - 1) / (16 * RB));
    const long long chunks = (a.col_end - (a.col_begin & ~(int64_t)15) + 511) / 512;   // the chunk grid starts on a multiple of 16
    // one workgroup per CU (its LDS holds the group's rows); per XCD `slots` column walkers x `groups` row groups
    // (rounded DOWN: the workgroups hold one CU each, 8 x slots x groups of them must fit the 256 CUs in ONE round -- rounded up,
    // three groups made 264 workgroups and the last eight ran after all the others)
    const int slots = std::max(1, std::min<int>(32 / std::max(1, groups), (int)((chunks + 7) / 8)));
    const size_t lds = (size_t)16 * RB * ((size_t)a.d_pad + search_row_pad(a.d_pad)) + (size_t)16 * RB * sizeof(float4);
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_search_filter<RB, NB>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_search_filter<RB, NB>), dim3(8u * (unsigned)slots * (unsigned)groups), dim3(512), lds, stream, a, groups, chunks);
    return 0;
}

static int launch_search_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt) {
    const int deep = opt.search_depth;            // register buffers of one k-slice each (k_search_filter: NB)
    switch (search_filter_rb(a)) {
        case 4: return deep >= 6 ? launch_search_filter_rb<4, 6>(stream, a) : deep == 5 ? launch_search_filter_rb<4, 5>(stream, a)
                                 : deep == 4 ? launch_search_filter_rb<4, 4>(stream, a) : launch_search_filter_rb<4, 3>(stream, a);
        case 2: return deep >= 6 ? launch_search_filter_rb<2, 6>(stream, a) : deep >= 4 ? launch_search_filter_rb<2, 4>(stream, a)
                                 : launch_search_filter_rb<2, 3>(stream, a);
        case 1: return deep >= 6 ? launch_search_filter_rb<1, 6>(stream, a) : deep >= 4 ? launch_search_filter_rb<1, 4>(stream, a)
                                 : launch_search_filter_rb<1, 3>(stream, a);
        default: return MVS_E_INVALID;
    }
}

// variant the filter launcher picks (launch_filter) for this block
static int filter_variant_for(const PairwiseArgs& a, const Options& opt) {
    int v = opt.filter_variant;
    // 50: the streaming search filter -- a block of few rows that is not under the symmetric schedule, against at least
    // 4096 columns (option search_stream = 0 leaves such blocks to the tile kernels).  Few = up to 640 when the kernel is
    // picked by size: 64 resident rows read the coarse plane once (10^6 columns: 0.35 ms), every further group of 64 reads
    // it again, mostly from the XCD's L2 -- streamed from the fragment-major plane 256 rows take 0.65 ms, 512 rows 1.15 ms
    // and 640 rows 1.48 ms (1.0 / 1.83 / 3.7 ms from the row-major plane; the 256 x 256 tile filter 1.3 / ~1.5 / 1.6 ms);
    // beyond ten groups fewer than 240 of the 256 CUs get a workgroup (1023 rows: 2.2 ms against 1.9 on tiles); asked for
    // by number (filter_variant 50) it takes up to 1023 rows.
    const int64_t few = v == 50 ? 1023 : 640;
    if ((v < 0 || v == 50) && opt.search_stream != 0 && !a.symmetric && a.row_end - a.row_begin <= few &&
        a.col_end - a.col_begin >= 4096 && search_filter_rb(a) > 0)
        return 50;
    if (v == 50) v = -1;
    if (v < 0) {
        const double tiles = (double)(a.row_end - a.row_begin) * (double)(a.col_end - a.col_begin) / 65536.0 *
                             (a.symmetric ? 0.5 : 1.0);
        // [r5] the ping-pong kernel on the fragment-major plane (B operand direct) wins from ~128 tiles on: symmetric
        // 4096^2 0.032 against 0.041 ms, 10 000^2 0.126 / 0.147, 12 544 x 12 544 without symmetry 0.291 / 0.421, 26 000^2
        // 0.594 / 0.870 (profiles/r05_exp_small_blocks.log); below that the 128 x 128 ring tiles keep more CUs busy
        v = tiles >= 128.0 ? 8 : 0;
    }
    return v;
}

int64_t filter_region_count(const PairwiseArgs& a, const Options& opt) {
    if (opt.cand_regions == 0) return 0;
    const int v = filter_variant_for(a, opt);
    const bool pp = (v >= 7 && v <= 10) || (v >= 40 && v <= 42) || (v >= 31 && v <= 33);
    if (!pp) return 0;                       // the ring kernels' epilogue appends with the per-wave atomic
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return 0;
    const int64_t n_tr = (rows + 255) / 256, n_tc = (cols + 255) / 256;
    const int64_t n = ((n_tr + 15) / 16) * ((n_tc + 15) / 16) * 256 * 8;      // workgroups of the launch x 8 waves
    // one 64-byte region + header per wave of the padded grid: 68 B x 123 M for a single 1M x 1M block (8 GB, all of it
    // cleared and scanned per attempt).  Beyond 8 M regions (a 180k x 180k block) the waves append with the atomic.
    return n <= (8 << 20) ? n : 0;
}

int launch_cand_gather(hipStream_t stream, const PairwiseArgs& a, int64_t n_regions) {
    if (n_regions <= 0) return 0;
    hipLaunchKernelGGL(k_cand_gather, dim3((unsigned)((n_regions + 255) / 256)), dim3(256), 0, stream, a,
                       (unsigned long long)n_regions);
    return 0;
}

bool filter_streams_rows(const PairwiseArgs& a, const Options& opt) {
    return a.limbs == 2 && a.d_pad <= 32768 && filter_variant_for(a, opt) == 50;
}

bool filter_streams(const PairwiseArgs& a, const Options& opt) {
    if (a.limbs != 2 || a.d_pad > 32768) return false;
    const int v = filter_variant_for(a, opt);
    return v == 50 || (v >= 7 && v <= 10) || (v >= 40 && v <= 42);      // the streaming search filter, the ping-pong tile filter
}

// the ping-pong exact kernel (two base-256 limbs) copies its LDS pieces from the fragment-major limb planes; the copy of
// the planes is worth building for blocks that run long enough (the kernel is picked by number 7 / 8 / 9, 8 by default)
bool exact_reads_fm(const PairwiseArgs& a, const Options& opt) {
    const int v = pairwise_variant(opt);
    return a.limbs == 2 && a.d_pad <= 32768 && v >= 7 && v <= 9 &&
           (double)(a.row_end - a.row_begin) * (double)(a.col_end - a.col_begin) >= 4194304.0;
}

bool filter_flags_tiles(const PairwiseArgs& a, const Options& opt) {
    if (opt.tile_dense_thr <= 0 || a.limbs != 2 || a.d_pad > 32768) return false;
    const int v = filter_variant_for(a, opt);
    const bool pp = (v >= 7 && v <= 10) || (v >= 40 && v <= 42);
    // the flagged tiles go to the ping-pong exact kernel (any other choice of exact kernel by number: list everything)
    return pp && pairwise_variant(opt) == 8;
}

void filter_tile_grid(const PairwiseArgs& a, int* n_tr, int* n_tc) {
    *n_tr = (int)((a.row_end - a.row_begin + 255) / 256);
    *n_tc = (int)((a.col_end - a.col_begin + 255) / 256);
}

int launch_tile_count(hipStream_t stream, const unsigned int* d_flags, int n_tr, int n_tc, int* d_row_count) {
    if (n_tr <= 0) return 0;
    hipLaunchKernelGGL(k_tile_count, dim3((unsigned)n_tr), dim3(256), 0, stream, d_flags, n_tc, d_row_count);
    return 0;
}

int launch_tile_list(hipStream_t stream, const unsigned int* d_flags, int n_tr, int n_tc, const int* d_row_count, int* d_list,
                     int cap) {
    if (n_tr <= 0) return 0;
    hipLaunchKernelGGL(k_tile_list, dim3((unsigned)n_tr), dim3(256), 0, stream, d_flags, n_tr, n_tc, d_row_count, d_list + 1, cap);
    return 0;
}

// n_hint < 0: n_cand is the count.  n_hint >= 0: the count is read on the device (a.cand_counter, at most a.cand_capacity)
// and the grid is sized for about n_hint entries (the kernel strides: any count is handled)
int launch_cand_prune(hipStream_t stream, const PairwiseArgs& a, unsigned long long n_cand, int2* d_out,
                      unsigned long long* d_out_count, long long n_hint) {
    const unsigned long long size_for = n_hint >= 0 ? (unsigned long long)n_hint : n_cand;
    if (n_hint < 0 && n_cand == 0) return 0;
    const unsigned long long blocks = std::max<unsigned long long>(1, std::min<unsigned long long>(4096ULL, (size_for + 2047) / 2048));
    hipLaunchKernelGGL(k_cand_prune, dim3((unsigned)blocks), dim3(256), 0, stream, a, n_hint >= 0 ? ~0ULL : n_cand, d_out, d_out_count);
    return 0;
}

// the exact ping-pong kernel on n_list flagged filter tiles (d_list: their ids in the grid of `a`, which is the filter
// launch's: same row / column origin and ranges, same symmetric square)
// device_count: n_list is the capacity the list (and this grid) was sized for, the count itself is read from d_list[-1]
int launch_exact_tiles(hipStream_t stream, const PairwiseArgs& a, const int* d_list, int n_list, const Options& opt,
                       bool device_count) {
    if (n_list <= 0) return 0;
    if (a.limbs != 2 || a.d_pad > 32768 || pairwise_variant(opt) != 8) return MVS_E_INVALID;
    using G = PpGeom<0>;
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    const int n_tr = (int)((rows + G::TM - 1) / G::TM), n_tc = (int)((cols + G::TN - 1) / G::TN);
    PairwiseArgs b = a;
    if (b.symmetric && !a.plan && ((a.row_begin - a.col_begin) % 256 != 0 || a.mirror_all)) b.symmetric = 0;   // as launch_pp<2> decided
    b.tile_list = d_list;
    b.tile_list_n = device_count ? -(n_list + 1) : n_list;
    const size_t lds = (size_t)4 * G::kStage;
    const unsigned per = (4u * (unsigned)n_list + 7u) / 8u;
    if (pp_direct_b(b, 0, opt)) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_pp<0, 4, 0, 0, 1, 0, 1>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return MVS_E_HIP;
        hipLaunchKernelGGL((k_pairwise_pp<0, 4, 0, 0, 1, 0, 1>), dim3(per * 8u), dim3(512), lds, stream, b, n_tr, n_tc, PlanSegs{});
        return 0;
    }
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_pp<0, 4>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    hipLaunchKernelGGL((k_pairwise_pp<0, 4>), dim3(per * 8u), dim3(512), lds, stream, b, n_tr, n_tc, PlanSegs{});
    return 0;
}

// A plan that ran its second half on the previous step's counts (mvs_plan_finish, speculative): did they hold?  counter =
// the context's counter block: [0] kept cells, [2] candidates, [12] verdict (out), [13] flagged tiles (out).  A plan whose
// candidate list or flagged-tile list was cut short says so in [12] and puts kPlanStale into the cell count, where every
// consumer of the cells looks first.
__global__ void k_plan_verdict(unsigned long long* __restrict__ counter, unsigned long long cand_capacity,
                               const int* __restrict__ tile_total, int tile_cap, int tiles_skipped) {
    const unsigned long long cand = counter[2];
    const long long flagged = tile_total ? (long long)*tile_total : 0;
    unsigned long long bad = 0;
    if (cand > cand_capacity) bad |= 1;
    if (flagged > (long long)tile_cap) bad |= 2;
    if (tiles_skipped && flagged > 0) bad |= 4;              // no tile was flagged last time: the tile passes were not launched
    counter[12] = bad;
    counter[13] = (unsigned long long)flagged;
    if (bad) counter[0] = kPlanStale;
}

int launch_plan_verdict(hipStream_t stream, unsigned long long* d_counter, unsigned long long cand_capacity, const int* d_tile_total,
                        int tile_cap, bool tiles_skipped) {
    hipLaunchKernelGGL(k_plan_verdict, dim3(1), dim3(1), 0, stream, d_counter, cand_capacity, d_tile_total, tile_cap, tiles_skipped ? 1 : 0);
    return 0;
}

int launch_exact_pairs(hipStream_t stream, const PairwiseArgs& a, const Options& opt, long long n_hint) {
    if (a.limbs != 2) return MVS_E_INVALID;
    // the candidate count lives on the device: a fixed grid of waves strides over the list.
    // opt.exact_variant 3 = tree reduction (default: 1.09 -> 0.88 ms on 1.3 M candidates at d = 2048); 0 = 64 pairs per
    // round, one shuffle butterfly per pair (10-15 % faster than 16 per round or a quarter wave per pair)
    const dim3 grid(256 * 16), block(256);
    if (opt.exact_variant == 3) {
        // n_hint >= 0: the caller read the candidate count back already -- a short list (a rank's share of a multi-GPU plan) does
        // not need 6144 workgroups to come and go: one round of 64 pairs per wave, at least one workgroup per CU
        unsigned wgs = 256u * (unsigned)(opt.recheck_blocks > 0 ? opt.recheck_blocks : 16);
        if (n_hint >= 0) {
            const unsigned long long want = ((unsigned long long)n_hint / 64 + 1 + 3) / 4;
            wgs = (unsigned)std::min<unsigned long long>(wgs, std::max<unsigned long long>(256, (want + 7) / 8 * 8));
        }
        const dim3 g(wgs);
        switch (opt.recheck_mode) {
            case 0: hipLaunchKernelGGL(k_exact_pairs_tree<0>, g, block, 0, stream, a); break;
            case 2: hipLaunchKernelGGL(k_exact_pairs_tree<2>, g, block, 0, stream, a); break;
            case 3: hipLaunchKernelGGL(k_exact_pairs_tree<3>, g, block, 0, stream, a); break;
            default: hipLaunchKernelGGL(k_exact_pairs_tree<1>, g, block, 0, stream, a); break;
        }
        return 0;
    }
    switch (opt.exact_variant) {
        case 1: hipLaunchKernelGGL((k_exact_pairs<16, 1>), grid, block, 0, stream, a); break;
        case 2: hipLaunchKernelGGL((k_exact_pairs<16, 0>), grid, block, 0, stream, a); break;
        default: hipLaunchKernelGGL((k_exact_pairs<64, 0>), grid, block, 0, stream, a); break;
    }
    return 0;
}

bool exact_kernel_writes_dense(const PairwiseArgs& a, const Options& opt) {
    const int v = pairwise_variant(opt);
    return a.limbs == 2 && a.d_pad <= 32768 && v >= 6 && v <= 9;      // the kernels that end in epilogue_exact16
}

// rows <= 16, two base-256 limbs, kept cells as a list: the streaming kernel.  Returns -1 when the block is not its kind.
template <int QT>
static int launch_skinny_qt(hipStream_t stream, const PairwiseArgs& a, size_t lds) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pairwise_skinny<QT>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return MVS_E_HIP;
    const unsigned per_cu = (unsigned)std::max<size_t>(1, std::min<size_t>(4, (size_t)(160 * 1024) / std::max<size_t>(lds, 1)));
    const int64_t cols = a.col_end - a.col_begin;
    const unsigned blocks = (unsigned)std::min<int64_t>(256 * per_cu, (cols + 7) / 8);
    hipLaunchKernelGGL(k_pairwise_skinny<QT>, dim3(blocks), dim3(512), lds, stream, a);
    return 0;
}

static int launch_skinny(hipStream_t stream, const PairwiseArgs& a, const Options& opt) {
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    // up to 16 rows: 10^6 columns x 2048 in 0.67 ms (1 row: 6.2 TB/s), 0.71 (4), 0.87 (8), 1.44 (16: bound by the rate of
    // v_dot2_i32_i16, ~12 cycles per wave instruction; with v_dot4_i32_i8 on the limb planes 1.05 / 1.9 ms); 32 rows would
    // take 2.9 ms, what the MFMA kernel takes
    if (a.limbs != 2 || rows < 1 || rows > 16 || cols < 1024 || a.dense != nullptr) return -1;
    if (opt.pairwise_variant != 8) return -1;                     // the caller asked for another MFMA kernel by number (8: default)
    int qt = 1;
    while (qt < rows) qt *= 2;
    const size_t lds = (size_t)qt * 2 * (size_t)a.d_pad;
    if (lds > 144 * 1024 || a.d_pad > 32768) return -1;
    switch (qt) {
        case 1: return launch_skinny_qt<1>(stream, a, lds);
        case 2: return launch_skinny_qt<2>(stream, a, lds);
        case 4: return launch_skinny_qt<4>(stream, a, lds);
        case 8: return launch_skinny_qt<8>(stream, a, lds);
        default: return launch_skinny_qt<16>(stream, a, lds);
    }
}

int launch_pairwise(hipStream_t stream, const PairwiseArgs& a, int mode, int algo, const Options& opt) {
    if (mode == 0 && algo == 0) {
        const int r = launch_skinny(stream, a, opt);
        if (r >= 0) return r;
    }
    // int32 accumulators hold up to two limb-pair products per k: exact while 2 * 128 * 128 * d_pad < 2^31;
    // longer sketches take the vector-ALU path, which wraps mod 2^32 by construction
    if (algo == 0 && (a.limbs <= 2 || is_k3(a.limbs)) && a.d_pad <= 32768) {
        if (is_k3(a.limbs))
            return mode == 0 ? launch_mfma<3, true, 0>(stream, a, opt) : launch_mfma<3, true, 1>(stream, a, opt);
        if (a.limbs == 1)
            return mode == 0 ? launch_mfma<1, false, 0>(stream, a, opt) : launch_mfma<1, false, 1>(stream, a, opt);
        return mode == 0 ? launch_mfma<2, false, 0>(stream, a, opt) : launch_mfma<2, false, 1>(stream, a, opt);
    }
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return 0;
    const int64_t gx = (cols + 63) / 64, gy = (rows + 3) / 4;
    // grid.y is limited to 65535: walk the rows in slabs
    for (int64_t y0 = 0; y0 < gy; y0 += 65535) {
        PairwiseArgs s = a;
        s.row_begin = a.row_begin + y0 * 4;
        const int64_t ny = gy - y0 < 65535 ? gy - y0 : 65535;
        if (mode == 1) s.dots = a.dots + (s.row_begin - a.row_begin) * cols;
        dim3 grid((unsigned)gx, (unsigned)ny);
        if (mode == 0)
            hipLaunchKernelGGL(k_pairwise_valu<0>, grid, dim3(256), 0, stream, s);
        else
            hipLaunchKernelGGL(k_pairwise_valu<1>, grid, dim3(256), 0, stream, s);
    }
    return 0;
}

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
                      const int2* d_ends, unsigned long long* d_size, unsigned int* d_jac, unsigned int* d_first_col, EncRow* d_par) {
    if (rows <= 0) return 0;
    int tr0, n_trows, n_tc;
    dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    const DenseSizes out{d_size, d_jac, d_first_col, d_par};
    if (d_size)
        hipLaunchKernelGGL(k_dense_fill<true>, dim3((unsigned)rows), dim3(256), 0, stream, d_dense, (long long)ld, (long long)n_cols,
                           d_row_ptr, d_col, d_q, (long long)active.row_rel0, tr0, n_tc, d_list, d_list_n, d_ends, out);
    else
        hipLaunchKernelGGL(k_dense_fill<false>, dim3((unsigned)rows), dim3(256), 0, stream, d_dense, (long long)ld, (long long)n_cols,
                           d_row_ptr, d_col, d_q, (long long)active.row_rel0, tr0, n_tc, d_list, d_list_n, d_ends, out);
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
