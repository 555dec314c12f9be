// read_pc_mat.hpp -- query library over the matrix shard folders that pairwise_comp_optimized writes.
// CPU only: this is the consumer side of the drop-in, not the accelerated path.
//
// Design (this build's own; only the public surface is the reference's):
//   SampleTable   one pass over <db>/vector_norms.txt: sample names, float norms and the name -> index
//                 table come out of the same scan (the reference scans the file once per accessor);
//   ShardView     one shard folder opened for reading: row directory as a sorted array (binary search), the
//                 first-column sequence, and matrix.bin behind one stream;
//   MatrixView    the shard set of an index folder; shards open lazily and stay open;
//   for_each_row  the one row-batch iterator both query modes are built on: requested rows are grouped by shard,
//                 visited in file order inside a shard, decoded into a reusable (column, q) buffer and handed to
//                 a callback together with their position in the request.
// The namespace-level functions below keep the names, argument meaning and failure behaviour of the reference's
// interface (include/read_pc_mat.h:38-107, behaviour src/read_pc_mat_cmp.cpp:29-120, :674-721, :989-1171) so that
// callers written against it (query_pc_mat, the pybind11 module) read the same.
#ifndef MVS_READ_PC_MAT_HPP
#define MVS_READ_PC_MAT_HPP

#include <algorithm>
#include <cerrno>
#include <charconv>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <filesystem>
#include <fstream>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <string_view>
#include <unordered_map>
#include <vector>

#include "mvs_codec.hpp"

namespace pc_mat {

struct Result {   // include/read_pc_mat.h:96-100
    std::string self_id;
    std::vector<std::string> neighbor_ids;
    std::vector<float> jaccard_similarities;
};

constexpr double kQuantLevels = 255.0;   // jaccard on read = q / 255 (writer: round(J * 255))

namespace text {

inline bool is_blank(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n' || c == '\v' || c == '\f'; }

inline std::string_view trimmed(std::string_view s) {
    size_t b = 0, e = s.size();
    while (b < e && is_blank(s[b])) ++b;
    while (e > b && is_blank(s[e - 1])) --e;
    return s.substr(b, e - b);
}

// next whitespace-delimited token of `s` starting at `pos` (advanced past it); empty when none is left
inline std::string_view next_token(std::string_view s, size_t& pos) {
    while (pos < s.size() && is_blank(s[pos])) ++pos;
    const size_t b = pos;
    while (pos < s.size() && !is_blank(s[pos])) ++pos;
    return s.substr(b, pos - b);
}

// What `stream >> double` accepts at the start of a token: optional sign, then digits / '.'; a prefix is enough
// ("1.5x" reads 1.5), words such as "nan" are a failure.
inline bool leading_number(std::string_view tok, double& out) {
    size_t i = 0;
    if (i < tok.size() && (tok[i] == '+' || tok[i] == '-')) ++i;
    if (i >= tok.size() || !((tok[i] >= '0' && tok[i] <= '9') || tok[i] == '.')) return false;
    const char* first = tok.data() + (tok[0] == '+' ? 1 : 0);
    const auto r = std::from_chars(first, tok.data() + tok.size(), out);
    return r.ec == std::errc() || r.ec == std::errc::result_out_of_range;
}

// std::stoi's verdict on a string: leading blanks and a sign are fine, at least one digit is needed, trailing text
// is ignored, values outside int fail
inline bool leading_int(const std::string& s, int& out) {
    const char* p = s.c_str();
    char* end = nullptr;
    errno = 0;
    const long v = std::strtol(p, &end, 10);
    if (end == p || errno == ERANGE || v < INT_MIN || v > INT_MAX) return false;
    out = (int)v;
    return true;
}

}  // namespace text

// ---------------------------------------------------------------------------------------------------
// vector_norms.txt: "<name> <norm>\n", line number = sample index
// ---------------------------------------------------------------------------------------------------
class SampleTable {
public:
    std::vector<std::string> names;                  // samples with a well-formed line, in file order
    std::vector<float> norms;                        // same length as names
    std::unordered_map<std::string, int> index_of;   // a repeated name keeps its LAST index (:49)
    int nonempty_lines = 0;                          // what get_total_vectors() reports
    bool opened = false;

    static std::string path_in(const std::string& folder) { return folder + "/vector_norms.txt"; }

    explicit SampleTable(const std::string& folder) {
        std::ifstream in(path_in(folder), std::ios::binary);
        if (!in) return;
        opened = true;
        std::string buf((std::istreambuf_iterator<char>(in)), std::istreambuf_iterator<char>());
        size_t at = 0;
        while (at < buf.size()) {
            size_t nl = buf.find('\n', at);
            if (nl == std::string::npos) nl = buf.size();
            const std::string_view line(buf.data() + at, nl - at);
            at = nl + 1;
            if (line.empty()) continue;
            ++nonempty_lines;
            size_t pos = 0;
            const std::string_view name = text::next_token(line, pos);
            const std::string_view value = text::next_token(line, pos);
            double norm = 0.0;
            if (name.empty() || !text::leading_number(value, norm)) continue;   // malformed: not a sample
            index_of[std::string(name)] = (int)names.size();
            names.emplace_back(name);
            norms.push_back((float)norm);
        }
    }
};

// ---------------------------------------------------------------------------------------------------
// one shard folder:  row_index.bin = row ids + byte-offset deltas, neighbor_start.bin = first column of each
// row, matrix.bin = per row the q values and (rows with more than one entry) the column deltas
// ---------------------------------------------------------------------------------------------------
struct Entry {
    uint64_t col;
    uint32_t q;
};

class ShardView {
public:
    explicit ShardView(const std::string& folder) {
        std::ifstream dir(folder + "/row_index.bin", std::ios::binary);
        std::ifstream first(folder + "/neighbor_start.bin", std::ios::binary);
        body_.open(folder + "/matrix.bin", std::ios::binary);
        if (!dir) {
            std::cerr << "Error: Could not open " << folder << "/row_index.bin" << std::endl;
            return;
        }
        if (!first || !body_) return;
        try {
            mvs_codec::compact_vector ids, gaps;
            ids.load(dir);
            gaps.load(dir);
            first_col_.load(first);
            slots_.reserve(ids.size());
            uint64_t offset = 0;   // the first row starts at 0, every later one at the running sum of the gaps
            for (uint64_t k = 0; k < ids.size(); ++k) {
                if (k) offset += gaps.access(k - 1);
                slots_.push_back(Slot{(uint32_t)ids.access(k), (uint32_t)k, offset});
            }
            std::sort(slots_.begin(), slots_.end(), [](const Slot& a, const Slot& b) { return a.row < b.row; });
            usable_ = true;
        } catch (const std::exception& e) {
            std::cerr << "Error: " << folder << ": " << e.what() << std::endl;
        }
    }

    bool usable() const { return usable_; }

    // byte offset of `row` in matrix.bin, or -1 if the shard holds nothing for it
    int64_t locate(uint32_t row, uint32_t& ordinal) const {
        const auto it = std::lower_bound(slots_.begin(), slots_.end(), row,
                                         [](const Slot& s, uint32_t r) { return s.row < r; });
        if (it == slots_.end() || it->row != row) return -1;
        ordinal = it->ordinal;
        return (int64_t)it->offset;
    }

    // entries of the row stored at `offset` (columns ascending) into `out`
    void decode(int64_t offset, uint32_t ordinal, std::vector<Entry>& out) {
        body_.clear();
        body_.seekg((std::streamoff)offset);
        mvs_codec::compact_vector levels;
        levels.load(body_);
        const uint64_t n = levels.size();
        out.resize(n);
        if (n == 0) return;
        gaps_.clear();
        if (n > 1) {   // a single-entry row carries no delta sequence (writer :732)
            mvs_codec::rice_sequence deltas;
            deltas.load(body_);
            deltas.decode(gaps_);
        }
        uint64_t col = first_col_.access(ordinal);
        for (uint64_t k = 0; k < n; ++k) {
            if (k) col += gaps_[k - 1];
            out[k] = Entry{col, (uint32_t)levels.access(k)};
        }
    }

private:
    struct Slot {
        uint32_t row, ordinal;   // ordinal = position in the shard's write order (indexes neighbor_start.bin)
        uint64_t offset;
    };
    std::vector<Slot> slots_;
    mvs_codec::rice_sequence first_col_;
    std::ifstream body_;
    std::vector<uint64_t> gaps_;
    bool usable_ = false;
};

// number of shards = 1 + the largest N among sub-directories called shard_N (:96-113)
inline int discover_shards(const std::string& matrix_folder) {
    long best = -1;
    std::error_code ec;
    std::filesystem::directory_iterator it(matrix_folder, ec), end;
    for (; !ec && it != end; it.increment(ec)) {
        if (!it->is_directory(ec)) continue;
        const std::string name = it->path().filename().string();
        if (name.size() <= 6 || name.compare(0, 6, "shard_") != 0) continue;
        long n = 0;
        bool digits = true;
        for (size_t k = 6; k < name.size() && digits; ++k) {
            digits = name[k] >= '0' && name[k] <= '9' && n < INT_MAX / 10;
            if (digits) n = n * 10 + (name[k] - '0');
        }
        if (digits) best = std::max(best, n);
    }
    return (int)(best + 1);
}

// shard that owns `row` when total_vectors rows are dealt out in equal runs of ceil(total / shards) (:117-120)
inline int get_shard_for_row(int row, int total_vectors, int num_shards) {
    const int run = (total_vectors + num_shards - 1) / num_shards;
    return row / run;
}

class MatrixView {
public:
    MatrixView(std::string folder, int total_vectors)
        : folder_(std::move(folder)), total_(total_vectors), shards_(discover_shards(folder_)) {}

    int shard_count() const { return shards_; }

    // Visit the stored entries of rows[i] for every i: fn(i, entries).  Rows nothing is stored for are not visited.
    template <typename Fn>
    void for_each_row(const std::vector<int>& rows, Fn&& fn) {
        struct Want {
            int64_t offset;
            uint32_t ordinal, position;
        };
        std::map<int, std::vector<uint32_t>> by_shard;
        for (uint32_t i = 0; i < rows.size(); ++i)
            if (rows[i] >= 0 && rows[i] < total_) by_shard[get_shard_for_row(rows[i], total_, shards_)].push_back(i);
        std::vector<Entry> entries;
        std::vector<Want> wants;
        for (const auto& [shard, positions] : by_shard) {
            ShardView* view = open(shard);
            if (!view) continue;
            wants.clear();
            for (const uint32_t p : positions) {
                uint32_t ordinal = 0;
                const int64_t off = view->locate((uint32_t)rows[p], ordinal);
                if (off >= 0) wants.push_back(Want{off, ordinal, p});
            }
            std::sort(wants.begin(), wants.end(), [](const Want& a, const Want& b) { return a.offset < b.offset; });
            for (const Want& w : wants) {
                view->decode(w.offset, w.ordinal, entries);
                fn(w.position, entries);
            }
        }
    }

private:
    ShardView* open(int shard) {
        auto it = open_.find(shard);
        if (it == open_.end())
            it = open_.emplace(shard, std::make_unique<ShardView>(folder_ + "/shard_" + std::to_string(shard))).first;
        return it->second->usable() ? it->second.get() : nullptr;
    }
    std::string folder_;
    int total_, shards_;
    std::unordered_map<int, std::unique_ptr<ShardView>> open_;
};

// ---------------------------------------------------------------------------------------------------
// the reference's interface
// ---------------------------------------------------------------------------------------------------

// names in file order + name -> index; an unreadable file gives an empty table and a message (:29-55)
inline std::unordered_map<std::string, int> load_vector_identifiers(const std::string& matrix_folder,
                                                                    std::vector<std::string>& identifiers) {
    SampleTable t(matrix_folder);
    if (!t.opened) {
        std::cerr << "Error: Could not open " << SampleTable::path_in(matrix_folder) << std::endl;
        return {};
    }
    identifiers.insert(identifiers.end(), t.names.begin(), t.names.end());
    return std::move(t.index_of);
}

// norms in file order; an unreadable file ends the process (:57-76)
inline void load_vector_norms(const std::string& matrix_folder, std::vector<float>& norms) {
    SampleTable t(matrix_folder);
    if (!t.opened) {
        std::cerr << "Error: Could not open " << SampleTable::path_in(matrix_folder) << std::endl;
        std::exit(1);
    }
    norms.insert(norms.end(), t.norms.begin(), t.norms.end());
}

// non-empty lines of vector_norms.txt, -1 if it cannot be read (:79-93)
inline int get_total_vectors(const std::string& matrix_folder) {
    const SampleTable t(matrix_folder);
    if (!t.opened) {
        std::cerr << "Error: Could not open " << SampleTable::path_in(matrix_folder) << std::endl;
        return -1;
    }
    return t.nonempty_lines;
}

// a query string that starts like an integer IS a row index; anything else is looked up by name (:674-689)
inline int parse_query_to_index(const std::string& query_str, const std::unordered_map<std::string, int>& id_to_index) {
    int as_number = 0;
    if (text::leading_int(query_str, as_number)) return as_number;
    const auto hit = id_to_index.find(query_str);
    if (hit == id_to_index.end()) {
        std::cerr << "Warning: Could not find identifier '" << query_str << "'" << std::endl;
        return -1;
    }
    return hit->second;
}

// one query per line; '#' lines and blank lines are skipped, unknown names dropped; id_vec receives the text of
// every query that was accepted (:692-721)
inline std::vector<int> read_queries_from_file(const std::string& filename,
                                               const std::unordered_map<std::string, int>& id_to_index,
                                               std::vector<std::string>& id_vec) {
    std::vector<int> rows;
    std::ifstream in(filename);
    if (!in) {
        std::cerr << "Error: Could not open query file " << filename << std::endl;
        return rows;
    }
    for (std::string raw; std::getline(in, raw);) {
        if (raw.empty() || raw.front() == '#') continue;
        const std::string token(text::trimmed(raw));
        if (token.empty()) continue;
        const int row = parse_query_to_index(token, id_to_index);
        if (row < 0) continue;
        rows.push_back(row);
        id_vec.push_back(token);
    }
    return rows;
}

// neighbours of every query row, strongest first (ties keep ascending column), jaccard = q / 255 (:989-1046)
inline std::vector<Result> query(std::string matrix_folder, std::vector<int>& queries, std::vector<float>& vector_norms,
                                 std::vector<std::string>& identifiers) {
    std::vector<Result> answers(queries.size());
    const int total = (int)vector_norms.size();
    MatrixView index(matrix_folder, total);
    if (index.shard_count() <= 0) {
        std::cerr << "Error: No shard folders found in " << matrix_folder << std::endl;
        return answers;
    }
    for (const int row : queries)
        if (row < 0 || row >= total)
            std::cout << "  Error: Query row " << row << " is out of range [0, " << total << ")" << std::endl;
    std::vector<uint32_t> order;
    index.for_each_row(queries, [&](uint32_t position, const std::vector<Entry>& entries) {
        if (entries.empty()) return;
        order.resize(entries.size());
        for (uint32_t k = 0; k < order.size(); ++k) order[k] = k;
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return entries[a].q > entries[b].q; });
        Result& r = answers[position];
        r.self_id = identifiers[(size_t)queries[position]];
        r.neighbor_ids.reserve(order.size());
        r.jaccard_similarities.reserve(order.size());
        for (const uint32_t k : order) {
            const Entry& e = entries[k];
            r.neighbor_ids.push_back(e.col < (uint64_t)total ? identifiers[e.col] : std::string("UNKNOWN"));
            r.jaccard_similarities.push_back((float)((double)e.q / kQuantLevels));
        }
    });
    return answers;
}

// rows x cols slice of the matrix; a cell that is not stored reads 0 (:1048-1171)
inline std::vector<std::vector<float>> query_sliced(std::string matrix_folder, std::vector<int32_t>& row_queries_vec,
                                                    std::vector<int32_t>& col_queries_vec, int32_t total_vectors,
                                                    std::vector<float>& /*vector_norms*/) {
    std::vector<std::vector<float>> slice(row_queries_vec.size(), std::vector<float>(col_queries_vec.size(), 0.0f));
    MatrixView index(matrix_folder, total_vectors);
    if (index.shard_count() <= 0) {
        std::cerr << "Error: No shard folders found in " << matrix_folder << std::endl;
        return slice;
    }
    // requested columns, ascending, each with the output positions that asked for it: one merge per row
    std::vector<std::pair<int64_t, uint32_t>> wanted;
    wanted.reserve(col_queries_vec.size());
    for (uint32_t c = 0; c < col_queries_vec.size(); ++c) wanted.emplace_back((int64_t)col_queries_vec[c], c);
    std::sort(wanted.begin(), wanted.end());
    index.for_each_row(row_queries_vec, [&](uint32_t position, const std::vector<Entry>& entries) {
        std::vector<float>& line = slice[position];
        size_t w = 0;
        for (const Entry& e : entries) {
            while (w < wanted.size() && wanted[w].first < (int64_t)e.col) ++w;
            for (size_t x = w; x < wanted.size() && wanted[x].first == (int64_t)e.col; ++x)
                line[wanted[x].second] = (float)((double)e.q / kQuantLevels);
        }
    });
    return slice;
}

}  // namespace pc_mat

#endif
