// standalone_projection -- drop-in for src/standalone_projection.cpp: one hash set per input line ->
// one line of <dimension> numbers on stdout (consumed by src/jaccard.py:98-118).
#include "mvs_host.hpp"

using namespace mvs_host;

int main(int argc, char* argv[]) {
    if (argc < 3) {                                                              // :12-15
        std::cerr << "Usage: " << argv[0] << " <hashes_file> <dimension>" << std::endl;
        return 1;
    }
    const std::string filename = argv[1];
    char* endp = nullptr;
    const long dl = strtol(argv[2], &endp, 10);                                  // :18 std::stoi
    if (endp == argv[2] || dl <= 0 || dl > (1 << 24)) {
        std::cerr << "Invalid dimension: " << argv[2] << std::endl;
        return 1;
    }
    const int d = (int)dl;
    HashSets sets;
    if (!read_hash_file(filename, false, sets)) {                                // :21-25
        std::cerr << "Error opening file: " << filename << std::endl;
        return 1;
    }
    const int64_t n = (int64_t)sets.offsets.size() - 1;
    if (n == 0) return 0;
    mvs_ctx* ctx = nullptr;
    if (mvs_ctx_create(pick_device(), &ctx) != MVS_OK) {
        std::cerr << "standalone_projection: " << mvs_last_error() << std::endl;
        return 2;
    }
    std::vector<int32_t> vec((size_t)n * d);
    if (mvs_project_csr(ctx, sets.hashes.data(), MVS_MEM_HOST, sets.offsets.data(), n, d, vec.data(),
                        MVS_MEM_HOST) != MVS_OK) {
        std::cerr << "standalone_projection: " << mvs_last_error() << std::endl;
        mvs_ctx_destroy(ctx);
        return 2;
    }
    mvs_ctx_destroy(ctx);
    std::string line;
    for (int64_t s = 0; s < n; ++s) {                                            // :39-42
        line.clear();
        for (int i = 0; i < d; ++i) {
            line += format_g_float(static_cast<float>(vec[(size_t)s * d + i]));
            if (i != d - 1) line += ' ';
        }
        line += '\n';
        std::cout << line;
    }
    std::cout.flush();
    return 0;
}
