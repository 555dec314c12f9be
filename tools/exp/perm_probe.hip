// gfx950 probes for k_pairwise_skinny:  hipcc --offload-arch=gfx950 -O3 -o /tmp/perm_probe tools/exp/perm_probe.hip && /tmp/perm_probe
//  * what __builtin_amdgcn_perm(a, b, sel) selects, v_dot2_i32_i16 on int16 pairs
//  * limbs_to_i16 (copied from csrc/mvs_pairwise.hip) on every value of a two-limb set, in every byte position
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef short v2s __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void limbs_to_i16(uint32_t lo, uint32_t hi, int& w01, int& w23) {
    const uint32_t neg = (lo >> 7) & 0x01010101u;
    const uint32_t hs = ((hi | 0x80808080u) - neg) ^ (~hi & 0x80808080u);
    w01 = (int)__builtin_amdgcn_perm(hs, lo, 0x05010400u);
    w23 = (int)__builtin_amdgcn_perm(hs, lo, 0x07030602u);
}
__global__ void k(unsigned* o) {
    const unsigned a = 0xA3A2A1A0u, b = 0xB3B2B1B0u;
    o[0] = __builtin_amdgcn_perm(a, b, 0x03020100u);
    o[1] = __builtin_amdgcn_perm(a, b, 0x07060504u);
    o[2] = __builtin_amdgcn_perm(a, b, 0x05010400u);
    const int x = (int)(((unsigned)(unsigned short)(short)-5 << 16) | (unsigned short)(short)300);   // {300, -5}
    const int y = (int)(((unsigned)(unsigned short)(short)7 << 16) | (unsigned short)(short)-2);     // {-2, 7}
    o[3] = (unsigned)__builtin_amdgcn_sdot2(__builtin_bit_cast(v2s, x), __builtin_bit_cast(v2s, y), 1000, false);  // 1000 - 600 - 35 = 365
}
__global__ void conv(const uint32_t* lo, const uint32_t* hi, int* w01, int* w23, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) limbs_to_i16(lo[i], hi[i], w01[i], w23[i]);
}
int main() {
    unsigned* d; unsigned h[4];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(1), 0, 0, d);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("perm(a=A3A2A1A0, b=B3B2B1B0): sel 03020100 -> %08X, sel 07060504 -> %08X, sel 05010400 -> %08X; dot2 -> %d (expect 365)\n", h[0], h[1], h[2], (int)h[3]);
    const int n = 65279;                       // v = -32639 .. 32639, four consecutive values per dword quadruple
    std::vector<uint32_t> lo(n), hi(n);
    auto digit = [](int v, int& l0, int& l1) { l0 = (int)(int8_t)(v & 0xff); l1 = (v - l0) >> 8; };
    for (int i = 0; i < n; ++i) {
        uint32_t L = 0, H = 0;
        for (int e = 0; e < 4; ++e) {
            int v = -32639 + (i + e * 7919) % 65279, l0, l1;
            digit(v, l0, l1);
            L |= (uint32_t)(uint8_t)l0 << (8 * e);
            H |= (uint32_t)(uint8_t)l1 << (8 * e);
        }
        lo[i] = L; hi[i] = H;
    }
    uint32_t *dl, *dh; int *d01, *d23;
    hipMalloc(&dl, n * 4); hipMalloc(&dh, n * 4); hipMalloc(&d01, n * 4); hipMalloc(&d23, n * 4);
    hipMemcpy(dl, lo.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dh, hi.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(conv, dim3((n + 255) / 256), dim3(256), 0, 0, dl, dh, d01, d23, n);
    std::vector<int> w01(n), w23(n);
    hipMemcpy(w01.data(), d01, n * 4, hipMemcpyDeviceToHost); hipMemcpy(w23.data(), d23, n * 4, hipMemcpyDeviceToHost);
    long bad = 0;
    for (int i = 0; i < n; ++i) {
        const int want[4] = {-32639 + (i) % 65279, -32639 + (i + 7919) % 65279, -32639 + (i + 2 * 7919) % 65279, -32639 + (i + 3 * 7919) % 65279};
        const int got[4] = {(int16_t)(w01[i] & 0xffff), (int16_t)((uint32_t)w01[i] >> 16), (int16_t)(w23[i] & 0xffff), (int16_t)((uint32_t)w23[i] >> 16)};
        for (int e = 0; e < 4; ++e)
            if (got[e] != want[e] && bad++ < 5) printf("  value %d in byte %d came out as %d\n", want[e], e, got[e]);
    }
    printf("limbs_to_i16: %ld wrong of %d\n", bad, 4 * n);
    return 0;
}
