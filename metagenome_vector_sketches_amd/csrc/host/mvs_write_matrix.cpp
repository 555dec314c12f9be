// mvs_write_matrix -- inverse of mvs_dump_matrix: "row col q" text lines (sorted by row, then col) ->
// <matrix_folder>/shard_k/ for k in [0, num_shards), rows assigned with the reference's formula
// (src/pairwise_comp_optimized.cpp:938-940).  CPU only; lets the reader stack be tested without a device.
#include "mvs_host.hpp"

int main(int argc, char* argv[]) {
    if (argc < 5) {
        std::cerr << "Usage: " << argv[0] << " <cells.txt> <matrix_folder> <total_vectors> <num_shards>" << std::endl;
        return 1;
    }
    std::ifstream in(argv[1]);
    if (!in) {
        std::cerr << "Error opening " << argv[1] << std::endl;
        return 1;
    }
    std::string folder = argv[2];
    if (folder.empty() || folder.back() != '/') folder += '/';
    const long long total = atoll(argv[3]);
    const int shards = atoi(argv[4]);
    if (total <= 0 || shards <= 0) {
        std::cerr << "total_vectors and num_shards must be positive" << std::endl;
        return 1;
    }
    std::vector<std::vector<mvs_cell>> per_shard((size_t)shards);
    const long long rps = (total + shards - 1) / shards;
    long long r, c, q;
    while (in >> r >> c >> q) {
        if (r < 0 || r >= total || r / rps >= shards) {
            std::cerr << "row " << r << " out of range" << std::endl;
            return 1;
        }
        per_shard[(size_t)(r / rps)].push_back(mvs_cell{(int32_t)r, (int32_t)c, 0, (int32_t)q});
    }
    try {
        for (int k = 0; k < shards; ++k)
            mvs_host::write_shard(folder + "shard_" + std::to_string(k) + "/", per_shard[(size_t)k].data(),
                                  per_shard[(size_t)k].size());
    } catch (const std::exception& e) {
        std::cerr << "Error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
