// query_pc_mat -- drop-in for the reference's query tool (src/query_pc_mat.cpp): nearest neighbours of
// query rows, or a row x column slice, out of the shard folders written by pairwise_comp_optimized.
// CPU only (IO bound); same options, stdout text and output files as the reference.
//
//   query_pc_mat --matrix M --db D  (--query_file F | --query_ids id... | --row_file R --col_file C)
//                [--top N] [--batch_size B] [--write_to_file OUT] [--show_all] [--print] [--help]
#include <chrono>
#include <cmath>
#include <iomanip>

#include "read_pc_mat.hpp"

namespace fs = std::filesystem;
using std::string;

static void show_error_and_exit(const std::string& msg) {   // :9-13
    std::cerr << msg << std::endl;
    std::cerr << "Aborting...\n";
    exit(1);
}

static std::pair<double, std::string> get_time_unit(double total_time) {   // :19-35
    if (total_time < 60) return {total_time, "seconds"};
    if (total_time < 60 * 60) return {total_time / 60.0, "minutes"};
    return {total_time / (60.0 * 60), "hours"};
}

static std::pair<std::string, std::string> split_path(const std::string& fullpath) {   // :37-46
    const size_t pos = fullpath.find_last_of("/\\");
    if (pos == std::string::npos) return {fullpath, "./"};
    return {fullpath.substr(pos + 1), fullpath.substr(0, pos)};
}

static std::string get_file_extension(const std::string& filename) {   // :225-232
    const size_t dot_pos = filename.find_last_of(".");
    return dot_pos != std::string::npos ? filename.substr(dot_pos + 1) : "";
}

// Minimal .npy writer with cnpy's "w"/"a" semantics for a float32 array of shape (rows, cols) growing along
// axis 0 (src/query_pc_mat.cpp:207-212 appends one (1, n) row per query row).  Fixed 128-byte v1.0 header.
static bool npy_append_row(const std::string& fname, const float* data, size_t cols, bool truncate) {
    size_t rows = 0;
    if (!truncate) {
        std::ifstream in(fname, std::ios::binary);
        char hdr[128];
        if (in.read(hdr, 128)) {
            const std::string h(hdr + 10, 118);
            const size_t p = h.find("'shape': (");
            if (p != std::string::npos) rows = (size_t)std::strtoull(h.c_str() + p + 10, nullptr, 10);
        }
    }
    std::fstream f;
    if (truncate || rows == 0)
        f.open(fname, std::ios::binary | std::ios::out | std::ios::trunc);
    else
        f.open(fname, std::ios::binary | std::ios::in | std::ios::out);
    if (!f) return false;
    std::string dict = "{'descr': '<f4', 'fortran_order': False, 'shape': (" + std::to_string(rows + 1) + ", " +
                       std::to_string(cols) + "), }";
    dict.resize(117, ' ');
    dict += '\n';
    const char magic[10] = {(char)0x93, 'N', 'U', 'M', 'P', 'Y', 1, 0, 118, 0};
    f.seekp(0);
    f.write(magic, 10);
    f.write(dict.data(), 118);
    f.seekp((std::streamoff)(128 + rows * cols * sizeof(float)));
    f.write(reinterpret_cast<const char*>(data), (std::streamsize)(cols * sizeof(float)));
    return (bool)f;
}

// :48-139
static void query_nearest_neighbors(const std::string& matrix_folder, const std::string& db_folder,
                                    const std::string& query_file, std::vector<std::string>& query_ids_str,
                                    bool write_to_file, bool show_all_neighbors, int64_t top_n, uint32_t batch_size,
                                    const std::string& out_fn, const std::string& sep, bool print_to_screen) {
    std::vector<string> identifiers;
    std::unordered_map<string, int> id_to_index = pc_mat::load_vector_identifiers(db_folder, identifiers);
    std::vector<std::string> query_id_vec;
    std::vector<int32_t> queries;
    if (!query_file.empty()) {
        queries = pc_mat::read_queries_from_file(query_file, id_to_index, query_id_vec);
    } else if (!query_ids_str.empty()) {
        for (const string& query_str : query_ids_str) {
            const int index = pc_mat::parse_query_to_index(query_str, id_to_index);
            if (index >= 0) queries.push_back(index);
        }
    } else {
        show_error_and_exit("Error: No queries specified. Use --query_file, --query_ids");
    }
    if (queries.empty()) show_error_and_exit("Error: No valid queries found");

    std::vector<float> vector_norms;
    pc_mat::load_vector_norms(db_folder, vector_norms);
    const int total_vectors = (int)identifiers.size();
    std::cout << "Total vectors loaded: " << total_vectors << std::endl << std::endl;
    if (total_vectors <= 0) show_error_and_exit("Error: Could not determine total number of vectors");

    auto [fname, out_file_path] = split_path(out_fn);
    std::chrono::duration<double> elapsed = std::chrono::duration<double>::zero();
    uint64_t start_indx = 0, end_indx;
    while (1) {
        end_indx = std::min<uint64_t>(start_indx + batch_size, queries.size());
        std::vector<int32_t> sub_queries(queries.begin() + start_indx, queries.begin() + end_indx);
        auto start = std::chrono::high_resolution_clock::now();
        std::vector<pc_mat::Result> all_results = pc_mat::query(matrix_folder, sub_queries, vector_norms, identifiers);
        auto end = std::chrono::high_resolution_clock::now();
        elapsed += (end - start);
        for (size_t i = 0; i < all_results.size(); i++) {
            const pc_mat::Result& res = all_results[i];
            if (print_to_screen)
                std::cout << "Query: " << res.self_id << " #Neighbors: " << res.neighbor_ids.size() << std::endl;
            std::ofstream out;
            if (write_to_file) {
                const std::string nfn = out_file_path + "/" + res.self_id + "_" + fname;
                std::cout << "Writing in file: " << nfn << std::endl << std::endl;
                out.open(nfn.c_str());
                out << "ID" + sep + "Jaccard\n";
            }
            const int64_t num_neighbors_to_show =
                show_all_neighbors ? (int64_t)res.neighbor_ids.size() : std::min<int64_t>(top_n, (int64_t)res.neighbor_ids.size());
            if (print_to_screen) std::cout << "Top " << num_neighbors_to_show << " neighbors:\n";
            for (int64_t j = 0; j < num_neighbors_to_show; ++j) {
                if (print_to_screen)
                    std::cout << j + 1 << ". Neighbor: " << res.neighbor_ids[(size_t)j]
                              << " Jaccard Similarity: " << res.jaccard_similarities[(size_t)j] << std::endl;
                if (write_to_file) out << res.neighbor_ids[(size_t)j] << sep << res.jaccard_similarities[(size_t)j] << std::endl;
            }
            if (print_to_screen) std::cout << std::endl;
            out.close();
        }
        auto time_unit = get_time_unit(elapsed.count());
        std::cout << "--------- Completed\t" << end_indx << "\tqueries in\t" << std::fixed << std::setprecision(2)
                  << time_unit.first << "\t" << time_unit.second << " ---------\n";
        if (end_indx == queries.size()) break;
        start_indx += batch_size;
    }
    auto time_unit = get_time_unit(elapsed.count());
    std::cout << "Query completed in " << std::fixed << std::setprecision(2) << time_unit.first << "\t" << time_unit.second
              << "\n" << std::endl;
}

// :141-223
static void query_sliced_matrix(const std::string& matrix_folder, const std::string& db_folder,
                                const std::string& row_file, const std::string& col_file, bool write_to_file,
                                const std::string& out_fn, uint32_t batch_size, bool print_to_screen,
                                const std::string& sep) {
    std::vector<string> identifiers;
    std::unordered_map<string, int> id_to_index = pc_mat::load_vector_identifiers(db_folder, identifiers);
    std::vector<std::string> row_vec, col_vec;
    std::vector<int32_t> row_query_vec = pc_mat::read_queries_from_file(row_file, id_to_index, row_vec);
    std::vector<int32_t> col_query_vec = pc_mat::read_queries_from_file(col_file, id_to_index, col_vec);
    if (row_query_vec.empty() || col_query_vec.empty()) show_error_and_exit("Empty row or col accessions.");
    std::vector<float> vector_norms;
    pc_mat::load_vector_norms(db_folder, vector_norms);
    const int total_vectors = (int)identifiers.size();
    std::cout << "Total vectors loaded: " << total_vectors << std::endl << std::endl;
    if (total_vectors <= 0) show_error_and_exit("Error: Could not determine total number of vectors");
    std::chrono::duration<double> elapsed = std::chrono::duration<double>::zero();
    uint64_t start_indx = 0, end_indx;

    std::ofstream out;
    if (write_to_file && sep != "-1") {
        std::cout << "Writing in file: " << out_fn << std::endl << std::endl;
        out.open(out_fn.c_str());
        out << "Accession" + sep;
        for (size_t i = 0; i < col_vec.size(); i++) out << col_vec[i] << sep;
        out << "\n";
    }
    if (print_to_screen) std::cout << "Accession\t";
    for (size_t i = 0; i < col_vec.size(); i++)
        if (print_to_screen) std::cout << col_vec[i] << "\t";
    if (print_to_screen) std::cout << "\n";

    while (1) {
        end_indx = std::min<uint64_t>(start_indx + batch_size, row_query_vec.size());
        std::vector<int32_t> row_sub_queries(row_query_vec.begin() + start_indx, row_query_vec.begin() + end_indx);
        auto start = std::chrono::high_resolution_clock::now();
        std::vector<std::vector<float>> all_results =
            pc_mat::query_sliced(matrix_folder, row_sub_queries, col_query_vec, total_vectors, vector_norms);
        auto end = std::chrono::high_resolution_clock::now();
        elapsed += (end - start);
        for (size_t i = 0; i < all_results.size(); i++) {
            std::vector<float>& res = all_results[i];
            if (print_to_screen) std::cout << row_vec[start_indx + i] << "\t";
            if (write_to_file && sep != "-1") out << row_vec[start_indx + i] << sep;
            if (print_to_screen || (write_to_file && sep != "-1")) {
                for (size_t j = 0; j < res.size(); ++j) {
                    if (print_to_screen) std::cout << res[j] << "\t";
                    if (write_to_file && sep != "-1") out << res[j] << sep;
                }
            }
            if (write_to_file && sep == "-1") {
                if (!npy_append_row(out_fn, res.data(), res.size(), start_indx == 0 && i == 0))
                    show_error_and_exit("Error: could not write " + out_fn);
            }
            if (print_to_screen) std::cout << std::endl;
            if (write_to_file && sep != "-1") out << "\n";
        }
        auto time_unit = get_time_unit(elapsed.count());
        std::cout << "--------- Completed\t" << end_indx << "\trows in\t" << std::fixed << std::setprecision(2)
                  << time_unit.first << "\t" << time_unit.second << " ---------\n";
        if (end_indx == row_query_vec.size()) break;
        start_indx += batch_size;
    }
    auto time_unit = get_time_unit(elapsed.count());
    std::cout << "Query completed in " << std::fixed << std::setprecision(2) << time_unit.first << "\t" << time_unit.second
              << "\n" << std::endl;
    if (write_to_file && sep != "-1") out.close();
}

static void print_help(const char* argv0) {   // :283-303
    std::cout << "Query Pairwise Comparison Matrix\n\n";
    std::cout << "Usage:\n        " << argv0
              << " [--matrix <folder>] [--db <folder>] [(--query_file <file> | --query_ids <ids>... | --row_file <row>"
                 " --col_file <col>)] [--top <int>] [--batch_size <int>] [--write_to_file <file>] [--show_all] [--print]"
                 " [--help]\n\n";
    std::cout << "Options:\n";
    std::cout << "  --matrix\t Folder containing the pairwise matrix files\n";
    std::cout << "  --db\t Folder containing the matrix meta data\n";
    std::cout << "  --query_file\t File containing query IDs (one per line)\n";
    std::cout << "  --query_ids\t Query IDs as command line arguments (numeric indices or identifiers)\n";
    std::cout << "  --row_file\t File containing query row IDs (one per line)\n";
    std::cout << "  --col_file\t File containing query col IDs (one per line)\n";
    std::cout << "  --top\t Number of top jaccard values to show [default 10]\n";
    std::cout << "  --batch_size\t Number of queries to process per batch [default 1000]\n";
    std::cout << "  --write_to_file\t Where to save the output (expected format: *.csv/*.tsv/*.npy/*npz for row-col query. "
                 "*.csv/*tsv/*txt for regular query).\n";
    std::cout << "  --show_all\t Whether to show all neighbors instead of top N\n";
    std::cout << "  --print\t Whether to print the outputs to screen\n";
    std::cout << "  --help\t Show this help message\n\n";
}

int main(int argc, char* argv[]) {
    string matrix_folder, db_folder, query_file, row_file, col_file, out_fn = "out.txt";
    uint32_t top_n = 10, batch_size = 1000;
    std::vector<string> query_ids_str;
    bool show_help = false, write_to_file = false, print_to_screen = false, show_all_neighbors = false;
    bool use_query_file = false, use_query_ids = false, use_row_col_files = false, ok = true;

    auto is_flag = [](const std::string& s) { return s.rfind("--", 0) == 0; };
    for (int i = 1; ok && i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](std::string& dst) {
            if (i + 1 >= argc) return false;
            dst = argv[++i];
            return true;
        };
        auto uvalue = [&](uint32_t& dst) {
            std::string v;
            if (!value(v)) return false;
            char* end = nullptr;
            const long x = strtol(v.c_str(), &end, 10);
            if (end == v.c_str() || *end || x < 0) return false;
            dst = (uint32_t)x;
            return true;
        };
        if (a == "--matrix") ok = value(matrix_folder);
        else if (a == "--db") ok = value(db_folder);
        else if (a == "--query_file") { use_query_file = true; ok = value(query_file); }
        else if (a == "--query_ids") {
            use_query_ids = true;
            while (i + 1 < argc && !is_flag(argv[i + 1])) query_ids_str.push_back(argv[++i]);
            ok = !query_ids_str.empty();
        }
        else if (a == "--row_file") { use_row_col_files = true; ok = value(row_file); }
        else if (a == "--col_file") ok = value(col_file);
        else if (a == "--top") ok = uvalue(top_n);
        else if (a == "--batch_size") ok = uvalue(batch_size);
        else if (a == "--write_to_file") { write_to_file = true; ok = value(out_fn); }
        else if (a == "--show_all") show_all_neighbors = true;
        else if (a == "--print") print_to_screen = true;
        else if (a == "--help") show_help = true;
        else ok = false;
    }
    if ((int)use_query_file + (int)use_query_ids + (int)use_row_col_files > 1) ok = false;   // alternatives (:268-275)
    if (use_row_col_files && col_file.empty()) ok = false;
    if (!ok || show_help) {
        print_help(argv[0]);
        return show_help ? 0 : 1;
    }
    if (matrix_folder.empty()) show_error_and_exit("Error: matrix folder is required.");
    if (!use_query_file && !use_query_ids && !use_row_col_files) show_error_and_exit("No query files given.");
    if (!fs::exists(matrix_folder)) show_error_and_exit("Error: Matrix folder does not exist.");
    if (matrix_folder.back() != '/' && matrix_folder.back() != '\\') matrix_folder += '/';
    if (!db_folder.empty() && db_folder.back() != '/' && db_folder.back() != '\\') db_folder += '/';
    if (write_to_file && out_fn.empty()) show_error_and_exit("No output filename given.");
    if (batch_size == 0) batch_size = 1;
    if (!write_to_file) print_to_screen = true;

    const std::string file_extension = get_file_extension(out_fn);
    if (use_query_file || use_query_ids) {
        if (write_to_file && file_extension != "csv" && file_extension != "tsv" && file_extension != "txt")
            show_error_and_exit("Output file extension is: " + file_extension + ". Expected: csv, tsv or txt.");
        const std::string sep = file_extension == "csv" ? "," : "\t";
        query_nearest_neighbors(matrix_folder, db_folder, query_file, query_ids_str, write_to_file, show_all_neighbors,
                                top_n, batch_size, out_fn, sep, print_to_screen);
    } else {
        if (row_file.empty() || col_file.empty()) show_error_and_exit("Either row or col file is not specified.");
        if (write_to_file && file_extension != "csv" && file_extension != "tsv" && file_extension != "npy" &&
            file_extension != "npz")
            show_error_and_exit("Output file extension is: " + file_extension + ". Expected: csv, tsv, npy or npz.");
        std::string sep = "-1";
        if (file_extension == "csv" || file_extension == "tsv") sep = file_extension == "csv" ? "," : "\t";
        query_sliced_matrix(matrix_folder, db_folder, row_file, col_file, write_to_file, out_fn, batch_size, print_to_screen,
                            sep);
    }
    return 0;
}
