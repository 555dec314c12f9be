// pp128.hip -- VERDICT r4 item 5, the structural variant that had not been built: the filter's k-loop with ONE wave per SIMD,
// 128 x 128 cells per wave (64 accumulator tiles = 256 registers, in AGPRs), 256 x 256 cells per workgroup of four waves.
// A standalone k-loop benchmark, not a product kernel: both operands come through an LDS ring filled by global_load_lds from a
// fragment-major int8 plane (the layout of the product's coarse_fm: one KiB = 16 samples x 64 k, lane l = (k/16 % 4) * 16 + sample % 16),
// the epilogue only counts the cells above a threshold.  What it answers: how busy does this structure keep the matrix pipes on the
// filter's own problem size (277 x 277 tiles of 256 x 256, d = 2048: the 76.6k tiles of the 100k comparison), against the product's
// ping-pong kernel (two waves per SIMD, 128 x 64 per wave: 8.4-8.5 ms, SQ_VALU_MFMA_BUSY 0.65).
//
//   hipcc -O3 --offload-arch=gfx950 -o pp128 pp128.hip && ./pp128 [tiles_per_side=277] [d=2048] [reps=5]
//
// Per k-slice (64 k) a wave issues 8 LDS-DMA copies of one KiB (slice s + 3), reads the 16 fragments of slice s + 1 from the LDS
// into its second register set while the 64 MFMAs of slice s run, waits for its own copies of slice s + 2 and meets the other
// three waves at ONE barrier.  Ring of 4 stages x 32 KiB.
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v4i __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

#define CHECK(x)                                                                                   \
    do {                                                                                           \
        hipError_t e_ = (x);                                                                       \
        if (e_ != hipSuccess) {                                                                    \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);            \
            exit(1);                                                                               \
        }                                                                                          \
    } while (0)

constexpr int kStage = 32 * 1024;   // A region (16 fragments) + B region (16 fragments) of one k-slice
constexpr int kStages = 4;

// the product's tile map (mvs_pairwise.hip: map_tile, mode 0): 16 x 16-tile super-patches, XCD label = workgroup id % 8 picks a
// 4-row x 8-column sub-patch (rotated by the patch row), (id / 8) % 32 walks it
__device__ __forceinline__ bool map_tile(unsigned b, unsigned patch_row, int n_tr, int n_tc, int* tr, int* tc) {
    const unsigned x = (b + patch_row) & 7u, q = b >> 3, ql = q & 31u;
    *tr = (int)patch_row * 16 + (int)(x >> 1) * 4 + (int)(ql >> 3);
    *tc = (int)(q >> 5) * 16 + (int)(x & 1u) * 8 + (int)(ql & 7u);
    return *tr < n_tr && *tc < n_tc;
}

template <bool COUNT_ONLY>
__global__ __launch_bounds__(256) void k_pp128(const int8_t* __restrict__ fm, int n_tr, int n_tc, int nk, int threshold,
                                               unsigned long long* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int tr, tc;
    if (!map_tile(blockIdx.x, blockIdx.y, n_tr, n_tc, &tr, &tc)) return;
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    // this wave's eight copy pieces of a slice: fragments wave * 8 .. + 7 of the stage (waves 0-1: rows, waves 2-3: columns)
    const int64_t blk0 = wave < 2 ? (int64_t)tr * 16 + wave * 8 : (int64_t)tc * 16 + (wave - 2) * 8;
    const int8_t* src = fm + blk0 * (int64_t)nk * 1024 + lane * 16;
    char* dst = smem + wave * 8 * 1024;
    auto copy_slice = [&](int ks) {
        char* st = dst + (ks & (kStages - 1)) * kStage;
#pragma unroll
        for (int p = 0; p < 8; ++p)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ((int64_t)p * nk + ks) * 1024), (lds_ptr_t)(st + p * 1024), 16, 0, 0);
    };
    const char* a_base = smem + (wm * 8) * 1024 + lane * 16;
    const char* b_base = smem + (16 + wn * 8) * 1024 + lane * 16;
    auto read_frags = [&](int ks, v4i (&fa)[8], v4i (&fb)[8]) {
        const int off = (ks & (kStages - 1)) * kStage;
#pragma unroll
        for (int i = 0; i < 8; ++i) fa[i] = *reinterpret_cast<const v4i*>(a_base + off + i * 1024);
#pragma unroll
        for (int j = 0; j < 8; ++j) fb[j] = *reinterpret_cast<const v4i*>(b_base + off + j * 1024);
    };
    v4i acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = v4i{0, 0, 0, 0};
    // The MFMAs as inline asm with the accumulator tied to an AGPR quadruple ("+a"): written with the builtin, the compiler
    // moved the 256 accumulator registers through VGPRs on every trip of the unrolled loop (428 v_accvgpr_* per two slices).
    auto mfma1 = [&](const v4i& fa, const v4i& fb, v4i& c) {
        asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(fa), "v"(fb));
    };
    auto mfmas = [&](const v4i (&fa)[8], const v4i (&fb)[8]) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) mfma1(fa[i], fb[j], acc[i][j]);
    };
    // the same with the 16 fragment reads of the NEXT slice spread over them: one read in front of every four MFMAs
    auto mfmas_and_reads = [&](const v4i (&fa)[8], const v4i (&fb)[8], int ks_next, v4i (&na)[8], v4i (&nb)[8]) {
        // (volatile asm on both sides: the order below is the order issued -- left to the scheduler, the 16 reads went out in a
        // block behind the 64 MFMAs, a step before they were needed, and the matrix pipe waited for the LDS.  The compiler does
        // not know these reads are in flight: the step that consumes them starts with s_waitcnt lgkmcnt(0).)
        const unsigned off = (unsigned)((ks_next & (kStages - 1)) * kStage);
        const unsigned pa = (unsigned)(size_t)(lds_ptr_t)a_base + off, pb = (unsigned)(size_t)(lds_ptr_t)b_base + off;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(na[i]) : "v"(pa), "n"(i * 1024));
#pragma unroll
            for (int j = 0; j < 4; ++j) mfma1(fa[i], fb[j], acc[i][j]);
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(nb[i]) : "v"(pb), "n"(i * 1024));
#pragma unroll
            for (int j = 4; j < 8; ++j) mfma1(fa[i], fb[j], acc[i][j]);
        }
    };
    // prologue: slices 0, 1, 2 on their way; slice 0 in the registers
    copy_slice(0);
    if (nk > 1) copy_slice(1);
    if (nk > 2) copy_slice(2);
    if (nk > 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    v4i a0[8], b0[8], a1[8], b1[8];
    read_frags(0, a0, b0);
    // One step.  The wave waits for ITS copies of slice ks + 1 (all but the 8 newest copy instructions: slice ks + 2's), the barrier
    // makes everybody's visible -- and says that every wave has consumed the fragments of slice ks - 1 (their reads were waited
    // for by the MFMAs of step ks - 1), whose stage the copies of slice ks + 3 may now overwrite.  Then the 16 fragment reads of
    // slice ks + 1 go out among the 64 MFMAs of slice ks (one read, four MFMAs: sched_group_barrier).
    // FULL: no conditions inside (the waitcnt pass is only exact in straight-line code: at a join it waited for every LDS read).
    auto step_full = [&](int ks, v4i (&ca)[8], v4i (&cb)[8], v4i (&na)[8], v4i (&nb)[8]) {
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        copy_slice(ks + 3);
        mfmas_and_reads(ca, cb, ks + 1, na, nb);
    };
    auto step_tail = [&](int ks, v4i (&ca)[8], v4i (&cb)[8], v4i (&na)[8], v4i (&nb)[8]) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ks + 1 < nk) {
            if (ks + 2 < nk) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            if (ks + 3 < nk) copy_slice(ks + 3);
            read_frags(ks + 1, na, nb);
        }
        mfmas(ca, cb);
    };
    int ks = 0;
    for (; ks + 5 <= nk; ks += 2) {            // both steps: ks + 1 + 3 < nk
        step_full(ks, a0, b0, a1, b1);
        step_full(ks + 1, a1, b1, a0, b0);
    }
    for (; ks + 2 <= nk; ks += 2) {
        step_tail(ks, a0, b0, a1, b1);
        step_tail(ks + 1, a1, b1, a0, b0);
    }
    if (ks < nk) step_tail(ks, a0, b0, a1, b1);
    // epilogue: cells above the threshold (the product's filter runs one fp32 MFMA per 16 x 16 cells here and appends candidates)
    unsigned mine = 0;
    long long sum = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                mine += acc[i][j][r] > threshold ? 1u : 0u;
                if (!COUNT_ONLY) sum += acc[i][j][r];
            }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mine += (unsigned)__shfl_xor((int)mine, o, 64);
        if (!COUNT_ONLY) sum += __shfl_xor(sum, o, 64);
    }
    if (lane == 0) {
        atomicAdd(out, (unsigned long long)mine);
        if (!COUNT_ONLY) atomicAdd(out + 1, (unsigned long long)sum);
    }
}

int main(int argc, char** argv) {
    const int nt = argc > 1 ? atoi(argv[1]) : 277;
    const int d = argc > 2 ? atoi(argv[2]) : 2048;
    const int reps = argc > 3 ? atoi(argv[3]) : 5;
    const int mode = argc > 4 ? atoi(argv[4]) : 0;      // operand values: 0 coarse-plane-like (|c| <= 127, sigma 36), 1 all zero, 2 .. 6 see below
    const int nk = d / 64;
    const int64_t rows = (int64_t)nt * 256;
    const size_t bytes = (size_t)rows * d;
    std::vector<int8_t> h(bytes);
    uint64_t s = 0x9e3779b97f4a7c15ULL;
    for (size_t i = 0; i < bytes; ++i) {       // values of a coarse plane: roughly normal, |c| <= 127
        s = s * 6364136223846793005ULL + 1442695040888963407ULL;
        const int a = (int)((s >> 33) & 63) + (int)((s >> 40) & 63) + (int)((s >> 47) & 63) + (int)((s >> 54) & 63) - 126;
        h[i] = (int8_t)(a > 127 ? 127 : (a < -127 ? -127 : a));
        if (mode == 1) h[i] = 0;
        if (mode == 2) h[i] = (int8_t)(a / 32);
        if (mode == 3) h[i] = (int8_t)(a / 2);            // |c| <= 63
        if (mode == 4) h[i] = (int8_t)(a / 4);            // |c| <= 31
        if (mode == 5) h[i] = (int8_t)(a / 8);            // |c| <= 15
        if (mode == 6) h[i] = (int8_t)(a / 2 + 64);       // 1 .. 127: the |c| <= 63 values on an offset, no sign changes
    }
    int8_t* dfm = nullptr;
    unsigned long long* dout = nullptr;
    CHECK(hipMalloc(&dfm, bytes));
    CHECK(hipMalloc(&dout, 16));
    CHECK(hipMemcpy(dfm, h.data(), bytes, hipMemcpyHostToDevice));
    const int lds = kStages * kStage;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pp128<true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_pp128<false>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const int n_sp = (nt + 15) / 16;
    const dim3 grid((unsigned)n_sp * 256u, (unsigned)n_sp);
    const int threshold = mode == 0 ? 40000 : (mode == 6 ? 8400000 : 5);
    // ---- correctness on the first tiles: the kernel's count and sum against the host's, from the fragment-major bytes ----
    {
        const int vt = nt < 2 ? nt : 2;          // 2 x 2 tiles
        CHECK(hipMemset(dout, 0, 16));
        const int vsp = 1;
        hipLaunchKernelGGL(k_pp128<false>, dim3(vsp * 256u, vsp), dim3(256), lds, 0, dfm, vt, vt, nk, threshold, dout);
        CHECK(hipDeviceSynchronize());
        unsigned long long got[2];
        CHECK(hipMemcpy(got, dout, 16, hipMemcpyDeviceToHost));
        auto at = [&](int64_t row, int k) {       // fragment-major: block row/16, slice k/64, lane (k/16 % 4) * 16 + row % 16, byte k % 16
            return (int)h[((row >> 4) * nk + (k >> 6)) * 1024 + ((((k >> 4) & 3) << 4) + (row & 15)) * 16 + (k & 15)];
        };
        unsigned long long cnt = 0;
        long long sum = 0;
        const int64_t vr = (int64_t)vt * 256;
        for (int64_t i = 0; i < vr; ++i)
            for (int64_t j = 0; j < vr; ++j) {
                int dot = 0;
                for (int k = 0; k < d; ++k) dot += at(i, k) * at(j, k);
                cnt += dot > threshold;
                sum += dot;
            }
        printf("check on %d x %d tiles: count %llu (host %llu), sum %lld (host %lld): %s\n", vt, vt, got[0], cnt, (long long)got[1], sum,
               got[0] == cnt && (long long)got[1] == sum ? "equal" : "DIFFERENT");
        if (got[0] != cnt || (long long)got[1] != sum) return 1;
    }
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int r = 0; r < reps + 2; ++r) {
        CHECK(hipMemset(dout, 0, 16));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(k_pp128<true>, grid, dim3(256), lds, 0, dfm, nt, nt, nk, threshold, dout);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms = 0.0f;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double tiles = (double)nt * nt, ops = tiles * 2.0 * 256 * 256 * (double)d;
        if (r >= 2)
            printf("pp128 [values %s]: %d x %d tiles of 256 x 256, d = %d: %.3f ms, %.1f TOP/s issued = %.3f of 5 POP/s; %.2f us per tile and CU "
                   "(product ping-pong filter: 76 636 tiles in 8.4-8.5 ms = 28.2 us, 0.48-0.49)\n",
                   mode == 0 ? "|c| <= 127" : mode == 1 ? "all zero" : mode == 2 ? "|c| <= 3" : mode == 3 ? "|c| <= 63" : mode == 4 ? "|c| <= 31" : mode == 5 ? "|c| <= 15" : "1 .. 127 (offset 64)", nt, nt, d, ms, ops / (ms * 1e-3) / 1e12, ops / (ms * 1e-3) / 5e15, ms * 1e3 / (tiles / 256.0));
    }
    return 0;
}
