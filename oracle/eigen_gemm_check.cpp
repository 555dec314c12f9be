// eigen_gemm_check -- what `block_i.transpose() * block_j` (src/pairwise_comp_optimized.cpp:135) evaluates to with the
// reference's OWN vendored Eigen (include/Eigen of the reference tree, nothing else of the reference is compiled):
// Eigen::MatrixXi blocks in the column-major d x c layout load_matrix_block() builds (:42-52), int32 arithmetic that
// wraps modulo 2^32.  TEST INFRASTRUCTURE: built into oracle/_ref/ by `make -C oracle ref` where /root/reference
// exists; tests/golden/make_golden.py records its output as a fixture, tests compare mvs_oracle_dot_i32 with it.
//   eigen_gemm_check <d> <c_i> <c_j> <seed> <magnitude>   ->  c_i * c_j int32 values, row-major, one per line
#include <Eigen/Dense>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

static uint64_t mix(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    return x ^ (x >> 31);
}

int main(int argc, char** argv) {
    if (argc != 6) {
        std::fprintf(stderr, "usage: %s d c_i c_j seed magnitude\n", argv[0]);
        return 1;
    }
    const int d = std::atoi(argv[1]), ci = std::atoi(argv[2]), cj = std::atoi(argv[3]);
    const uint64_t seed = std::strtoull(argv[4], nullptr, 10);
    const int64_t mag = std::atoll(argv[5]);
    // the same generator the Python side uses (tests/golden/make_golden.py): value(sample, k)
    auto value = [&](int sample, int k) -> int32_t {
        const uint64_t r = mix(seed * 1000003ULL + (uint64_t)sample * 65537ULL + (uint64_t)k);
        return (int32_t)((int64_t)(r % (uint64_t)(2 * mag + 1)) - mag);
    };
    Eigen::MatrixXi block_i(d, ci), block_j(d, cj);             // column-major: one sample per column (:42)
    for (int s = 0; s < ci; ++s)
        for (int k = 0; k < d; ++k) block_i(k, s) = value(s, k);
    for (int s = 0; s < cj; ++s)
        for (int k = 0; k < d; ++k) block_j(k, s) = value(1000 + s, k);
    const Eigen::MatrixXi dot_products = block_i.transpose() * block_j;   // :135
    for (int i = 0; i < ci; ++i)
        for (int j = 0; j < cj; ++j) std::printf("%d\n", dot_products(i, j));
    return 0;
}
