// read_pc_mat.hpp -- the query library over the matrix shard folders: namespace pc_mat with the
// reference's interface (include/read_pc_mat.h:38-107; implementation src/read_pc_mat_cmp.cpp), on top of
// this build's codec (mvs_codec.hpp).  CPU only: this is the consumer that defines the matrix layout, not
// part of the accelerated path.
//
// Same names, argument meaning and error behaviour as the reference:
//   load_vector_identifiers / load_vector_norms / get_total_vectors   src/read_pc_mat_cmp.cpp:29-93
//   discover_shards / get_shard_for_row                                :96-120
//   parse_query_to_index / read_queries_from_file                      :674-721
//   query                                                              :989-1046
//   query_sliced                                                       :1136-1171
#ifndef MVS_READ_PC_MAT_HPP
#define MVS_READ_PC_MAT_HPP

#include <algorithm>
#include <cstdint>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <regex>
#include <sstream>
#include <string>
#include <unordered_map>
#include <vector>

#include "mvs_codec.hpp"

namespace pc_mat {

struct Neighbors {
    std::vector<std::pair<uint64_t, uint32_t>> index_jaccard;
};

struct Result {
    std::string self_id;
    std::vector<std::string> neighbor_ids;
    std::vector<float> jaccard_similarities;
};

// :29-55  "<identifier> <norm>" per line of <folder>/vector_norms.txt; duplicate names overwrite (:49)
inline std::unordered_map<std::string, int> load_vector_identifiers(const std::string& matrix_folder,
                                                                    std::vector<std::string>& identifiers) {
    std::unordered_map<std::string, int> id_to_index;
    const std::string norms_file = matrix_folder + "/vector_norms.txt";
    std::ifstream norms_in(norms_file);
    if (!norms_in) {
        std::cerr << "Error: Could not open " << norms_file << std::endl;
        return id_to_index;
    }
    std::string line;
    int index = 0;
    while (std::getline(norms_in, line)) {
        if (line.empty()) continue;
        std::istringstream iss(line);
        std::string identifier;
        double norm;
        if (iss >> identifier >> norm) {
            identifiers.push_back(identifier);
            id_to_index[identifier] = index;
            index++;
        }
    }
    return id_to_index;
}

// :57-76  exits when the file cannot be opened
inline void load_vector_norms(const std::string& matrix_folder, std::vector<float>& norms) {
    const std::string norms_file = matrix_folder + "/vector_norms.txt";
    std::ifstream norms_in(norms_file);
    if (!norms_in) {
        std::cerr << "Error: Could not open " << norms_file << std::endl;
        exit(1);
    }
    std::string line;
    while (std::getline(norms_in, line)) {
        if (line.empty()) continue;
        std::istringstream iss(line);
        std::string identifier;
        float norm;
        if (iss >> identifier >> norm) norms.push_back(norm);
    }
}

// :79-93
inline int get_total_vectors(const std::string& matrix_folder) {
    const std::string norms_file = matrix_folder + "/vector_norms.txt";
    std::ifstream norms_in(norms_file);
    if (!norms_in) {
        std::cerr << "Error: Could not open " << norms_file << std::endl;
        return -1;
    }
    int count = 0;
    std::string line;
    while (std::getline(norms_in, line))
        if (!line.empty()) count++;
    return count;
}

// :96-113  number of shards = 1 + largest N of a directory named shard_N
inline int discover_shards(const std::string& matrix_folder) {
    int max_shard = -1;
    std::error_code ec;
    for (const auto& entry : std::filesystem::directory_iterator(matrix_folder, ec)) {
        if (!entry.is_directory()) continue;
        const std::string dirname = entry.path().filename().string();
        static const std::regex shard_pattern(R"(shard_(\d+))");
        std::smatch matches;
        if (std::regex_match(dirname, matches, shard_pattern)) max_shard = std::max(max_shard, std::stoi(matches[1].str()));
    }
    return max_shard + 1;
}

// :117-120
inline int get_shard_for_row(int row, int total_vectors, int num_shards) {
    const int rows_per_shard = (total_vectors + num_shards - 1) / num_shards;
    return row / rows_per_shard;
}

// :145-175  row -> (position in the shard's build order, byte address in matrix.bin)
inline std::unordered_map<uint32_t, std::pair<uint32_t, uint64_t>> get_shard_row_to_address_map_jaccard(
    const std::string& shard_folder) {
    std::unordered_map<uint32_t, std::pair<uint32_t, uint64_t>> row_to_address_map;
    const std::string index_filename = shard_folder + "/row_index.bin";
    std::ifstream index_file(index_filename, std::ios::binary);
    if (!index_file) {
        std::cerr << "Error: Could not open " << index_filename << std::endl;
        return row_to_address_map;
    }
    mvs_codec::compact_vector row_cv, delta_address_cv;
    try {
        row_cv.load(index_file);
        delta_address_cv.load(index_file);
    } catch (const std::exception& e) {
        std::cerr << "Error: " << index_filename << ": " << e.what() << std::endl;
        return row_to_address_map;
    }
    if (row_cv.size() == 0) return row_to_address_map;   // empty shard (the reference dereferences row 0, :161)
    uint64_t addr = 0;
    for (uint64_t i = 0; i < row_cv.size(); ++i) {
        if (i > 0) addr += delta_address_cv.access(i - 1);   // first position is always 0, rest are delta coded
        row_to_address_map[(uint32_t)row_cv.access(i)] = std::make_pair((uint32_t)i, addr);
    }
    return row_to_address_map;
}

namespace detail {
// decode one row at `addr`: (column, quantised jaccard) pairs, columns ascending
inline void read_row(std::ifstream& bin_in, const mvs_codec::rice_sequence& rs_start, uint32_t build_index,
                     uint64_t addr, std::vector<std::pair<uint64_t, uint32_t>>& out) {
    bin_in.clear();
    bin_in.seekg((std::streamoff)addr, std::ios::beg);
    mvs_codec::compact_vector cv_jc;
    cv_jc.load(bin_in);
    std::vector<uint64_t> delta;
    if (cv_jc.size() > 1) {
        mvs_codec::rice_sequence rs_delta;
        rs_delta.load(bin_in);
        rs_delta.decode(delta);
    }
    out.resize(cv_jc.size());
    if (out.empty()) return;
    uint64_t col = rs_start.access(build_index);
    out[0] = std::make_pair(col, (uint32_t)cv_jc.access(0));
    for (uint64_t i = 1; i < cv_jc.size(); ++i) {
        col += delta[i - 1];
        out[i] = std::make_pair(col, (uint32_t)cv_jc.access(i));
    }
}

struct ShardFiles {
    std::unordered_map<uint32_t, std::pair<uint32_t, uint64_t>> rows;
    std::ifstream bin_in;
    mvs_codec::rice_sequence rs_start;
    bool ok = false;
    explicit ShardFiles(const std::string& shard_folder) {
        rows = get_shard_row_to_address_map_jaccard(shard_folder);
        bin_in.open(shard_folder + "/matrix.bin", std::ios::binary);
        std::ifstream ngh_in(shard_folder + "/neighbor_start.bin", std::ios::binary);
        if (!bin_in || !ngh_in) return;
        try {
            rs_start.load(ngh_in);
            ok = true;
        } catch (const std::exception& e) {
            std::cerr << "Error: " << shard_folder << "/neighbor_start.bin: " << e.what() << std::endl;
        }
    }
};
}  // namespace detail

// :597-671  queries are grouped by shard; a row that is absent from its shard yields no neighbours
inline std::vector<Neighbors> load_neighbors_for_rows_jaccard_wo_sort(const std::string& matrix_folder,
                                                                     const std::vector<int>& rows,
                                                                     uint32_t total_vectors, int num_shards) {
    std::unordered_map<int, std::vector<uint32_t>> shard_to_queries;
    for (uint32_t i = 0; i < rows.size(); ++i)
        shard_to_queries[get_shard_for_row(rows[i], (int)total_vectors, num_shards)].emplace_back(i);
    std::vector<Neighbors> results(rows.size());
    for (const auto& [shard_idx, query_index_vec] : shard_to_queries) {
        detail::ShardFiles sf(matrix_folder + "/shard_" + std::to_string(shard_idx));
        if (!sf.ok) continue;
        for (const uint32_t query_index : query_index_vec) {
            const auto it = sf.rows.find((uint32_t)rows[query_index]);
            if (it == sf.rows.end()) continue;
            detail::read_row(sf.bin_in, sf.rs_start, it->second.first, it->second.second,
                             results[query_index].index_jaccard);
        }
    }
    return results;
}

// :674-689  numbers are taken as row indices first, anything else is looked up as an identifier
inline int parse_query_to_index(const std::string& query_str, const std::unordered_map<std::string, int>& id_to_index) {
    try {
        return std::stoi(query_str);
    } catch (const std::exception&) {
        const auto it = id_to_index.find(query_str);
        if (it != id_to_index.end()) return it->second;
        std::cerr << "Warning: Could not find identifier '" << query_str << "'" << std::endl;
        return -1;
    }
}

// :692-721  one id per line, '#' comments and empty lines skipped, unknown ids dropped
inline std::vector<int> read_queries_from_file(const std::string& filename,
                                               const std::unordered_map<std::string, int>& id_to_index,
                                               std::vector<std::string>& id_vec) {
    std::vector<int> queries;
    std::ifstream file(filename);
    if (!file) {
        std::cerr << "Error: Could not open query file " << filename << std::endl;
        return queries;
    }
    std::string line;
    while (std::getline(file, line)) {
        if (line.empty() || line[0] == '#') continue;
        const size_t b = line.find_first_not_of(" \t\r\n");
        if (b == std::string::npos) continue;   // whitespace only (the reference would throw in stoi's fallback path)
        line.erase(0, b);
        line.erase(line.find_last_not_of(" \t\r\n") + 1);
        const int index = parse_query_to_index(line, id_to_index);
        if (index >= 0) {
            queries.push_back(index);
            id_vec.push_back(line);
        }
    }
    return queries;
}

// :989-1046  neighbours of every query row, sorted by quantised jaccard (descending), jaccard = q / 255
inline std::vector<Result> query(std::string matrix_folder, std::vector<int>& queries, std::vector<float>& vector_norms,
                                 std::vector<std::string>& identifiers) {
    const int num_shards = discover_shards(matrix_folder);
    if (num_shards <= 0) {
        std::cerr << "Error: No shard folders found in " << matrix_folder << std::endl;
        return std::vector<Result>(queries.size());
    }
    const uint32_t total_vectors = (uint32_t)vector_norms.size();
    const double MULT_CONST = (1ULL << 8) - 1;
    std::vector<int> valid_rows(queries);
    for (int& r : valid_rows)
        if (r < 0 || (uint32_t)r >= total_vectors) r = 0;   // looked up but ignored below
    std::vector<Neighbors> all_neighbors = load_neighbors_for_rows_jaccard_wo_sort(matrix_folder, valid_rows,
                                                                                  total_vectors, num_shards);
    std::vector<Result> all_results(queries.size());
    for (size_t q = 0; q < queries.size(); ++q) {
        const int query_row = queries[q];
        if (query_row < 0 || (uint32_t)query_row >= total_vectors) {
            std::cout << "  Error: Query row " << query_row << " is out of range [0, " << total_vectors << ")" << std::endl;
            continue;
        }
        Neighbors& neighbors = all_neighbors[q];
        if (neighbors.index_jaccard.empty()) continue;
        std::stable_sort(neighbors.index_jaccard.begin(), neighbors.index_jaccard.end(),
                         [](const std::pair<uint64_t, uint32_t>& a, const std::pair<uint64_t, uint32_t>& b) {
                             return a.second > b.second;
                         });
        Result res;
        res.self_id = identifiers[(size_t)query_row];
        for (const auto& [neighbor_idx, neighbor_jaccard] : neighbors.index_jaccard) {
            res.neighbor_ids.push_back(neighbor_idx < total_vectors ? identifiers[neighbor_idx] : "UNKNOWN");
            res.jaccard_similarities.push_back((float)(static_cast<double>(neighbor_jaccard) / MULT_CONST));
        }
        all_results[q] = std::move(res);
    }
    return all_results;
}

// :1048-1171  rows x cols slice; an absent cell is 0
inline std::vector<std::vector<float>> query_sliced(std::string matrix_folder, std::vector<int32_t>& row_queries_vec,
                                                    std::vector<int32_t>& col_queries_vec, int32_t total_vectors,
                                                    std::vector<float>& /*vector_norms*/) {
    const int num_shards = discover_shards(matrix_folder);
    std::vector<std::vector<float>> all_results(row_queries_vec.size(),
                                                std::vector<float>(col_queries_vec.size(), 0.0f));
    if (num_shards <= 0) {
        std::cerr << "Error: No shard folders found in " << matrix_folder << std::endl;
        return all_results;
    }
    const double MULT_CONST = (1ULL << 8) - 1;
    std::unordered_map<int, std::vector<uint32_t>> shard_to_queries;
    for (uint32_t i = 0; i < row_queries_vec.size(); ++i)
        shard_to_queries[get_shard_for_row(row_queries_vec[i], total_vectors, num_shards)].emplace_back(i);
    for (const auto& [shard_idx, query_index_vec] : shard_to_queries) {
        detail::ShardFiles sf(matrix_folder + "/shard_" + std::to_string(shard_idx));
        if (!sf.ok) continue;
        for (const uint32_t query_index : query_index_vec) {
            const auto it = sf.rows.find((uint32_t)row_queries_vec[query_index]);
            if (it == sf.rows.end()) continue;
            std::vector<std::pair<uint64_t, uint32_t>> row;
            detail::read_row(sf.bin_in, sf.rs_start, it->second.first, it->second.second, row);
            std::unordered_map<int64_t, uint32_t> col_to_q;
            for (const auto& [col, q] : row) col_to_q[(int64_t)col] = q;
            for (size_t c = 0; c < col_queries_vec.size(); ++c) {
                const auto hit = col_to_q.find((int64_t)col_queries_vec[c]);
                if (hit != col_to_q.end()) all_results[query_index][c] = (float)(static_cast<double>(hit->second) / MULT_CONST);
            }
        }
    }
    return all_results;
}

}  // namespace pc_mat

#endif
