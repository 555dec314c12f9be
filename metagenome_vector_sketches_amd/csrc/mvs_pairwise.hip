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
#include "mvs_pairwise_dev.h"

#include <algorithm>
#include <cstring>
#include <type_traits>

namespace mvs {

namespace {

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

// what launch_pp makes of a block before it launches: the tile grid and the arguments the kernel gets (symmetric schedule only
// with aligned grids, the skinny map for grids of less than a patch row)
template <int MODE>
static bool pp_geometry(const PairwiseArgs& a, PairwiseArgs* b, int* n_tr, int* n_tc) {
    using G = PpGeom<MODE>;
    const int64_t rows = a.row_end - a.row_begin, cols = a.col_end - a.col_begin;
    if (rows <= 0 || cols <= 0) return false;
    *n_tr = (int)((rows + G::TM - 1) / G::TM);
    *n_tc = (int)((cols + G::TN - 1) / G::TN);
    *b = a;
    if (b->symmetric && ((a.row_begin - a.col_begin) % G::TM != 0 || a.mirror_all)) b->symmetric = 0;
    skinny_map(*b, *n_tr);
    return true;
}

template <int MODE, int NST, int ORDER = 0, int ABL = 0, int PH = 1, int NT = 0, int BD = 0>
int launch_pp(hipStream_t stream, const PairwiseArgs& a, const unsigned* order = nullptr, unsigned order_per = 0) {
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
    if (order != nullptr && order_per > 0 && b.map_mode != 3) {
        // the block as ONE segment of a balanced tile order (filter_order_geometry + plan_tile_order built the list for exactly
        // these arguments): a 1-D grid of order_per workgroups per XCD label
        PlanSegs segs{};
        segs.n = 1;
        segs.n_tr[0] = n_tr;
        segs.n_tc[0] = n_tc;
        segs.n_spc[0] = n_spc;
        segs.i_begin[0] = a.row_begin;
        segs.j_begin[0] = a.col_begin;
        for (int k = 1; k <= kPlanSegs; ++k) segs.wg_begin[k] = order_per * 8u;
        segs.order = order;
        segs.order_per = order_per;
        hipLaunchKernelGGL((k_pairwise_pp<MODE, NST, ORDER, ABL, PH, NT, BD>), dim3(order_per * 8u), dim3(512), lds, stream, b, n_tr, n_tc, segs);
        return 0;
    }
    hipLaunchKernelGGL((k_pairwise_pp<MODE, NST, ORDER, ABL, PH, NT, BD>), dim3((unsigned)n_spc * 256u, (unsigned)n_spr), dim3(512), lds, stream,
                       b, n_tr, n_tc, PlanSegs{});
    return 0;
}

// the default ping-pong kernel (4-stage ring), with the B operand straight from the fragment-major plane when that exists
template <int MODE>
int launch_pp_default(hipStream_t stream, const PairwiseArgs& a, const Options& opt, const unsigned* order = nullptr, unsigned order_per = 0) {
    if (pp_direct_b(a, MODE, opt)) return launch_pp<MODE, 4, 0, 0, 1, 0, 1>(stream, a, order, order_per);
    return launch_pp<MODE, 4>(stream, a, order, order_per);
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

static int filter_variant_for(const PairwiseArgs& a, const Options& opt);
static int launch_search_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt);

// The one-block filter launch with a balanced tile order (the default ping-pong kernel only): the segment the launch will be and
// the arguments its kernel will get, for plan_tile_order -- false where the launch takes another kernel or map.
bool filter_order_geometry(const PairwiseArgs& a, const Options& opt, PairwiseArgs* b, PlanSegs* segs) {
    if (a.limbs != 2 || a.d_pad > 32768 || filter_variant_for(a, opt) != 8) return false;
    int n_tr = 0, n_tc = 0;
    if (!pp_geometry<2>(a, b, &n_tr, &n_tc) || b->map_mode == 3) return false;
    *segs = PlanSegs{};
    segs->n = 1;
    segs->n_tr[0] = n_tr;
    segs->n_tc[0] = n_tc;
    segs->n_spc[0] = (n_tc + 15) / 16;
    segs->i_begin[0] = a.row_begin;
    segs->j_begin[0] = a.col_begin;
    return true;
}

int launch_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt, const unsigned* order, unsigned order_per) {
    if (a.limbs != 2 || a.d_pad > 32768) return MVS_E_INVALID;
    // opt.filter_variant: tile shape / ring depth of the one-pass filter.
    // Default (-1): the ping-pong kernel on 256 x 256 tiles (half the L2 -> LDS bytes per cell of 128 x 128 tiles; its
    // two wave groups overlap copies and MFMAs: 11.4 -> 10.0 ms at 100k samples against the ring kernel on the same
    // tiles) from 128 tiles on (filter_variant_for), 128 x 128 ring tiles for the small blocks below that.
    const int v = filter_variant_for(a, opt);
    switch (v) {
        case 50: return launch_search_filter(stream, a, opt);
        case 7: return launch_pp<2, 5>(stream, a);   // ping-pong wave groups, 256 x 256, 5-stage ring (all 160 KiB of LDS)
        case 8: return launch_pp_default<2>(stream, a, opt, order, order_per);   // the same on a 4-stage ring (B operand direct when the fragment-major plane exists)
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

}  // namespace mvs
