// mvs_dump_matrix -- decode <matrix_folder>/shard_k/ back to text "row col q" lines (file order).
// A test/debug companion of pairwise_comp_optimized; the reference's reader stack (query_pc_mat,
// read_pc_mat_module) is the next scope row.
#include "mvs_host.hpp"

int main(int argc, char* argv[]) {
    if (argc < 2) {
        std::cerr << "Usage: " << argv[0] << " <shard_folder/> [--legacy16]" << std::endl;
        return 1;
    }
    std::string folder = argv[1];
    if (folder.empty() || folder.back() != '/') folder += '/';
    const bool legacy16 = argc > 2 && std::string(argv[2]) == "--legacy16";   // lines are then "row col round(dot/d)"
    std::vector<mvs_cell> cells;
    try {
        if (!(legacy16 ? mvs_host::read_shard_legacy16(folder, cells) : mvs_host::read_shard(folder, cells))) {
            std::cerr << "Error opening shard files in " << folder << std::endl;
            return 1;
        }
    } catch (const std::exception& e) {
        std::cerr << "Error decoding " << folder << ": " << e.what() << std::endl;
        return 1;
    }
    std::string out;
    for (const mvs_cell& c : cells)
        out += std::to_string(c.row) + " " + std::to_string(c.col) + " " + std::to_string(legacy16 ? c.dot : c.q) + "\n";
    std::cout << out;
    return 0;
}
