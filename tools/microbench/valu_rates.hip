// valu_rates.hip -- measures issue cost (SIMD cycles per wave64 instruction) of the integer VALU
// instructions the projection kernel is made of, on the device it runs on.  Evidence for DESIGN.md.
//   hipcc -O3 --offload-arch=gfx950 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <string>

#define REP 64
#define ITER 256

// each body: 8 independent chains so that latency is hidden; REP instructions per chain element
#define KERNEL(name, decl, body)                                                         \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) {          \
        decl;                                                                            \
        for (int it = 0; it < ITER; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { body; }                \
        }                                                                                \
        out[blockIdx.x * 256 + threadIdx.x] = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;     \
    }

#define DECL32 uint32_t a0 = seed + threadIdx.x, a1 = a0 * 3 + 1, a2 = a0 * 5 + 2, a3 = a0 * 7 + 3, \
    a4 = a0 * 11 + 4, a5 = a0 * 13 + 5, a6 = a0 * 17 + 6, a7 = a0 * 19 + 7; uint32_t c = seed | 1

#define ASM8(op)                                                                   \
    asm volatile(op " %0, %0, %8\n\t" op " %1, %1, %8\n\t" op " %2, %2, %8\n\t" op " %3, %3, %8\n\t" \
                 op " %4, %4, %8\n\t" op " %5, %5, %8\n\t" op " %6, %6, %8\n\t" op " %7, %7, %8"     \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)   \
                 : "v"(c))

KERNEL(k_xor, DECL32, ASM8("v_xor_b32"))
KERNEL(k_add, DECL32, ASM8("v_add_u32"))
KERNEL(k_mul_lo, DECL32, ASM8("v_mul_lo_u32"))
KERNEL(k_mul_hi, DECL32, ASM8("v_mul_hi_u32"))
KERNEL(k_mul_u24, DECL32, ASM8("v_mul_u32_u24"))
KERNEL(k_mul_hi_u24, DECL32, ASM8("v_mul_hi_u32_u24"))
KERNEL(k_lshl, DECL32, ASM8("v_lshlrev_b32"))

#define ASM8_3(op, extra)                                                                \
    asm volatile(op " %0, %0, %8, %1" extra "\n\t" op " %1, %1, %8, %2" extra "\n\t" op " %2, %2, %8, %3" extra "\n\t" \
                 op " %3, %3, %8, %4" extra "\n\t" op " %4, %4, %8, %5" extra "\n\t" op " %5, %5, %8, %6" extra "\n\t" \
                 op " %6, %6, %8, %7" extra "\n\t" op " %7, %7, %8, %0" extra                                       \
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)                 \
                 : "v"(c))

KERNEL(k_bitop3, DECL32, ASM8_3("v_bitop3_b32", " bitop3:0x96"))
KERNEL(k_add3, DECL32, ASM8_3("v_add3_u32", ""))
KERNEL(k_mad_u24, DECL32, ASM8_3("v_mad_u32_u24", ""))
KERNEL(k_alignbit, DECL32, ASM8_3("v_alignbit_b32", ""))
KERNEL(k_mad_u32_u16, DECL32, ASM8_3("v_mad_u32_u16", ""))
KERNEL(k_lshl_add, DECL32, ASM8_3("v_lshl_add_u32", ""))
KERNEL(k_dot4, DECL32, ASM8_3("v_dot4_i32_i8", ""))
KERNEL(k_pk_mul_lo_u16, DECL32, ASM8("v_pk_mul_lo_u16"))
KERNEL(k_pk_mad_u16, DECL32, ASM8_3("v_pk_mad_u16", ""))

// 64-bit ops
#define DECL64 uint64_t b0 = seed + threadIdx.x, b1 = b0 * 3 + 1, b2 = b0 * 5 + 2, b3 = b0 * 7 + 3, \
    b4 = b0 * 11 + 4, b5 = b0 * 13 + 5, b6 = b0 * 17 + 6, b7 = b0 * 19 + 7; uint32_t c = seed | 1;   \
    uint64_t c64 = ((uint64_t)seed << 32) | 12345u
#define KERNEL64(name, body)                                                             \
    __global__ __launch_bounds__(256) void name(uint32_t* out, uint32_t seed) {          \
        DECL64;                                                                          \
        for (int it = 0; it < ITER; ++it) {                                              \
            _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) { body; }                \
        }                                                                                \
        uint64_t x = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;                              \
        out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);        \
    }
#define ASM8_64(fmt)                                                                     \
    asm volatile(fmt(0) fmt(1) fmt(2) fmt(3) fmt(4) fmt(5) fmt(6) fmt(7)                 \
                 : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7) \
                 : "v"(c), "v"(c64))
#define F_LSHR64(i) "v_lshrrev_b64 %" #i ", 27, %" #i "\n\t"
#define F_LSHLADD64(i) "v_lshl_add_u64 %" #i ", %" #i ", 0, %9\n\t"
#define F_MAD64(i) "v_mad_u64_u32 %" #i ", vcc, %8, %8, %" #i "\n\t"
#define F_MULF64(i) "v_mul_f64 %" #i ", %" #i ", %9\n\t"
#define F_FMAF64(i) "v_fma_f64 %" #i ", %" #i ", %9, %9\n\t"
KERNEL64(k_lshr64, ASM8_64(F_LSHR64))
KERNEL64(k_lshl_add64, ASM8_64(F_LSHLADD64))
__global__ __launch_bounds__(256) void k_mad64(uint32_t* out, uint32_t seed) {
    DECL64;
    for (int it = 0; it < ITER; ++it) {
        _Pragma("unroll") for (int r = 0; r < REP / 8; ++r) {
            asm volatile(F_MAD64(0) F_MAD64(1) F_MAD64(2) F_MAD64(3) F_MAD64(4) F_MAD64(5) F_MAD64(6) F_MAD64(7)
                         : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3), "+v"(b4), "+v"(b5), "+v"(b6), "+v"(b7)
                         : "v"(c), "v"(c64) : "vcc");
        }
    }
    uint64_t x = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}
KERNEL64(k_mulf64, ASM8_64(F_MULF64))
KERNEL64(k_fmaf64, ASM8_64(F_FMAF64))

// full splitmix64 tail as the compiler builds it (per call: 2 64-bit multiplies)
__global__ __launch_bounds__(256) void k_splitmix(uint32_t* out, uint32_t seed) {
    DECL64;
    (void)c; (void)c64;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
#define SM(z) z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL; z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL; z = z ^ (z >> 31);
            SM(b0) SM(b1) SM(b2) SM(b3) SM(b4) SM(b5) SM(b6) SM(b7)
        }
    }
    uint64_t x = b0 ^ b1 ^ b2 ^ b3 ^ b4 ^ b5 ^ b6 ^ b7;
    out[blockIdx.x * 256 + threadIdx.x] = (uint32_t)x ^ (uint32_t)(x >> 32);
}

using v4i = __attribute__((ext_vector_type(4))) int;
using v16i = __attribute__((ext_vector_type(16))) int;
// MFMA alone, and MFMA interleaved with independent VALU work (does the matrix pipe run beside it?)
template <int VALU_PER_MFMA>
__global__ __launch_bounds__(256) void k_mfma_mix(uint32_t* out, uint32_t seed) {
    DECL32;
    v4i fa = {(int)a0, (int)a1, (int)a2, (int)a3}, fb = {(int)a4, (int)a5, (int)a6, (int)a7};
    v16i acc0 = {0}, acc1 = {0};
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            acc0 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fa, fb, acc0, 0, 0, 0);
            if (VALU_PER_MFMA >= 8) { ASM8("v_xor_b32"); }
            if (VALU_PER_MFMA >= 16) { ASM8("v_add_u32"); }
            acc1 = __builtin_amdgcn_mfma_i32_32x32x32_i8(fb, fa, acc1, 0, 0, 0);
            if (VALU_PER_MFMA >= 8) { ASM8("v_xor_b32"); }
            if (VALU_PER_MFMA >= 16) { ASM8("v_add_u32"); }
        }
    }
    uint32_t x = a0 ^ a1 ^ a2 ^ a3 ^ a4 ^ a5 ^ a6 ^ a7;
    for (int i = 0; i < 16; ++i) x ^= (uint32_t)(acc0[i] ^ acc1[i]);
    out[blockIdx.x * 256 + threadIdx.x] = x;
}

typedef void (*kern_t)(uint32_t*, uint32_t);
struct Case { const char* name; kern_t k; double instr_per_thread; };

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    int clk_khz = 0; hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    printf("device %s, %d CUs, clock attr %.0f MHz\n", p.gcnArchName, cus, clk_khz / 1000.0);
    const int blocks = cus * 8;   // 8 blocks x 4 waves = 32 waves per CU = 8 per SIMD
    uint32_t* d; hipMalloc(&d, (size_t)blocks * 256 * 4);
    const double n = (double)ITER * REP;
    std::vector<Case> cases = {
        {"v_xor_b32", k_xor, n}, {"v_add_u32", k_add, n}, {"v_lshlrev_b32", k_lshl, n},
        {"v_bitop3_b32", k_bitop3, n}, {"v_add3_u32", k_add3, n}, {"v_lshl_add_u32", k_lshl_add, n},
        {"v_alignbit_b32", k_alignbit, n},
        {"v_mul_lo_u32", k_mul_lo, n}, {"v_mul_hi_u32", k_mul_hi, n}, {"v_mul_u32_u24", k_mul_u24, n},
        {"v_mul_hi_u32_u24", k_mul_hi_u24, n}, {"v_mad_u32_u24", k_mad_u24, n}, {"v_mad_u32_u16", k_mad_u32_u16, n},
        {"v_pk_mul_lo_u16", k_pk_mul_lo_u16, n}, {"v_pk_mad_u16", k_pk_mad_u16, n}, {"v_dot4_i32_i8", k_dot4, n},
        {"v_lshrrev_b64", k_lshr64, n}, {"v_lshl_add_u64", k_lshl_add64, n}, {"v_mad_u64_u32", k_mad64, n},
        {"v_mul_f64", k_mulf64, n}, {"v_fma_f64", k_fmaf64, n},
        {"splitmix64 tail (per call)", k_splitmix, n},
        {"mfma_i32_32x32x32_i8 alone", k_mfma_mix<0>, n / 8 * 2},
        {"mfma + 8 VALU each (per mfma)", k_mfma_mix<8>, n / 8 * 2},
        {"mfma + 16 VALU each (per mfma)", k_mfma_mix<16>, n / 8 * 2},
    };
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (auto& c : cases) {
        hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
        hipDeviceSynchronize();
        float best = 1e30f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(c.k, dim3(blocks), dim3(256), 0, 0, d, 12345u);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
        }
        // wave-instructions per SIMD = instr_per_thread * waves_per_SIMD(8)
        const double wave_instr_per_simd = c.instr_per_thread * 8.0;
        const double ns_per = best * 1e6 / wave_instr_per_simd;
        printf("%-34s %8.3f ms  %7.3f ns per wave-instr per SIMD  = %6.2f cycles @2.4GHz\n", c.name, best, ns_per,
               ns_per * 2.4);
    }
    return 0;
}
