// query_pc_mat -- command-line reader of the matrix shards pairwise_comp_optimized writes: nearest neighbours of
// query samples, or a rows x columns slice.  Drop-in for the reference's tool of the same name (src/query_pc_mat.cpp):
// the options, the text on stdout / stderr, the exit codes and the files written are the reference's; the program
// behind them is this build's own:
//   * one table describes the options (parsing and the help text both come from it);
//   * output goes through small sink classes (screen, per-query neighbour file, delimited slice, .npy slice) so the
//     two query modes share one batch driver;
//   * the .npy sink writes its 128-byte header once and patches the row count at the end.
// CPU only (IO bound).
//
//   query_pc_mat --matrix M --db D  (--query_file F | --query_ids id... | --row_file R --col_file C)
//                [--top N] [--batch_size B] [--write_to_file OUT] [--show_all] [--print] [--help]
#include <chrono>
#include <cstdio>
#include <functional>
#include <iomanip>

#include "read_pc_mat.hpp"

namespace {

[[noreturn]] void abort_with(const std::string& message) {
    std::cerr << message << std::endl << "Aborting...\n";
    std::exit(1);
}

// ---------------------------------------------------------------------------------------------------
// options
// ---------------------------------------------------------------------------------------------------
struct Settings {
    std::string matrix, db, query_file, row_file, col_file, out_name = "out.txt";
    std::vector<std::string> query_ids;
    uint32_t top = 10, batch = 1000;
    bool to_file = false, show_all = false, to_screen = false, help = false;
    bool saw_query_file = false, saw_query_ids = false, saw_row_file = false;
};

enum class Takes { Word, Count, Words, Nothing };

struct Option {
    const char* flag;
    Takes takes;
    std::function<void(Settings&, const std::string&)> store;   // called once per value (Nothing: once, with "")
    const char* about;
};

const std::vector<Option>& option_table() {
    static const std::vector<Option> table = {
        {"--matrix", Takes::Word, [](Settings& s, const std::string& v) { s.matrix = v; },
         "Folder containing the pairwise matrix files"},
        {"--db", Takes::Word, [](Settings& s, const std::string& v) { s.db = v; },
         "Folder containing the matrix meta data"},
        {"--query_file", Takes::Word, [](Settings& s, const std::string& v) { s.query_file = v; s.saw_query_file = true; },
         "File containing query IDs (one per line)"},
        {"--query_ids", Takes::Words, [](Settings& s, const std::string& v) { s.query_ids.push_back(v); s.saw_query_ids = true; },
         "Query IDs as command line arguments (numeric indices or identifiers)"},
        {"--row_file", Takes::Word, [](Settings& s, const std::string& v) { s.row_file = v; s.saw_row_file = true; },
         "File containing query row IDs (one per line)"},
        {"--col_file", Takes::Word, [](Settings& s, const std::string& v) { s.col_file = v; },
         "File containing query col IDs (one per line)"},
        {"--top", Takes::Count, [](Settings& s, const std::string& v) { s.top = (uint32_t)std::stoul(v); },
         "Number of top jaccard values to show [default 10]"},
        {"--batch_size", Takes::Count, [](Settings& s, const std::string& v) { s.batch = (uint32_t)std::stoul(v); },
         "Number of queries to process per batch [default 1000]"},
        {"--write_to_file", Takes::Word, [](Settings& s, const std::string& v) { s.out_name = v; s.to_file = true; },
         "Where to save the output (expected format: *.csv/*.tsv/*.npy/*npz for row-col query. *.csv/*tsv/*txt for "
         "regular query)."},
        {"--show_all", Takes::Nothing, [](Settings& s, const std::string&) { s.show_all = true; },
         "Whether to show all neighbors instead of top N"},
        {"--print", Takes::Nothing, [](Settings& s, const std::string&) { s.to_screen = true; },
         "Whether to print the outputs to screen"},
        {"--help", Takes::Nothing, [](Settings& s, const std::string&) { s.help = true; }, "Show this help message"},
    };
    return table;
}

bool all_digits(const std::string& s) {
    return !s.empty() && s.size() <= 9 && std::all_of(s.begin(), s.end(), [](char c) { return c >= '0' && c <= '9'; });
}

bool looks_like_flag(const char* s) { return s[0] == '-' && s[1] == '-'; }

// false: the command line does not fit the table (the caller prints the help and fails)
bool read_command_line(int argc, char** argv, Settings& s) {
    for (int at = 1; at < argc;) {
        const Option* opt = nullptr;
        for (const Option& o : option_table())
            if (std::string(argv[at]) == o.flag) opt = &o;
        if (!opt) return false;
        ++at;
        switch (opt->takes) {
            case Takes::Nothing:
                opt->store(s, "");
                break;
            case Takes::Word:
            case Takes::Count:
                if (at >= argc) return false;
                if (opt->takes == Takes::Count && !all_digits(argv[at])) return false;
                opt->store(s, argv[at++]);
                break;
            case Takes::Words: {
                int taken = 0;
                for (; at < argc && !looks_like_flag(argv[at]); ++at, ++taken) opt->store(s, argv[at]);
                if (!taken) return false;
                break;
            }
        }
    }
    // the three ways of naming queries exclude one another; a row file needs its column file
    if ((int)s.saw_query_file + (int)s.saw_query_ids + (int)s.saw_row_file > 1) return false;
    if (s.saw_row_file && s.col_file.empty()) return false;
    return true;
}

void print_help(const char* program) {
    std::cout << "Query Pairwise Comparison Matrix\n\nUsage:\n        " << program
              << " [--matrix <folder>] [--db <folder>] [(--query_file <file> | --query_ids <ids>... | --row_file <row>"
                 " --col_file <col>)] [--top <int>] [--batch_size <int>] [--write_to_file <file>] [--show_all] [--print]"
                 " [--help]\n\nOptions:\n";
    for (const Option& o : option_table()) std::cout << "  " << o.flag << "\t " << o.about << "\n";
    std::cout << "\n";
}

// ---------------------------------------------------------------------------------------------------
// timing lines:  "--------- Completed\t<n>\t<noun> in\t<t>\t<unit> ---------"  /  "Query completed in <t>\t<unit>"
// Only the library calls are on the clock, not the printing.
// ---------------------------------------------------------------------------------------------------
class Clock {
public:
    template <typename Fn>
    auto timed(Fn&& fn) {
        const auto t0 = std::chrono::steady_clock::now();
        auto result = fn();
        spent_ += std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        return result;
    }
    void report_batch(uint64_t done, const char* noun) const {
        // the two-decimal fixed notation stays switched on for std::cout afterwards, exactly as in the reference:
        // values printed by later batches of the same run appear with two decimals
        std::cout << "--------- Completed\t" << done << "\t" << noun << " in\t" << std::fixed << std::setprecision(2)
                  << scaled() << "\t" << unit() << " ---------\n";
    }
    void report_total() const {
        std::cout << "Query completed in " << std::fixed << std::setprecision(2) << scaled() << "\t" << unit() << "\n"
                  << std::endl;
    }

private:
    double scaled() const { return spent_ < 60 ? spent_ : spent_ < 3600 ? spent_ / 60 : spent_ / 3600; }
    const char* unit() const { return spent_ < 60 ? "seconds" : spent_ < 3600 ? "minutes" : "hours"; }
    double spent_ = 0;
};

// calls work(first, last) for consecutive index ranges of at most `batch` items, reporting after each
void in_batches(size_t total, size_t batch, const char* noun, Clock& clock, const std::function<void(size_t, size_t)>& work) {
    for (size_t first = 0; first < total; first += batch) {
        const size_t last = std::min(total, first + batch);
        work(first, last);
        clock.report_batch(last, noun);
    }
    clock.report_total();
}

std::string extension_of(const std::string& name) {
    const size_t dot = name.rfind('.');
    return dot == std::string::npos ? std::string() : name.substr(dot + 1);
}

// ---------------------------------------------------------------------------------------------------
// samples of the DB folder + the rows a request names
// ---------------------------------------------------------------------------------------------------
struct Catalogue {
    std::vector<std::string> names;
    std::vector<float> norms;
    std::unordered_map<std::string, int> index_of;
    explicit Catalogue(const std::string& db_folder) {
        index_of = pc_mat::load_vector_identifiers(db_folder, names);
    }
    void finish_loading(const std::string& db_folder) {   // after the queries were resolved, as the reference orders it
        pc_mat::load_vector_norms(db_folder, norms);
        std::cout << "Total vectors loaded: " << names.size() << std::endl << std::endl;
        if (names.empty()) abort_with("Error: Could not determine total number of vectors");
    }
};

// ---------------------------------------------------------------------------------------------------
// mode 1: neighbours of each query
// ---------------------------------------------------------------------------------------------------
void neighbour_mode(const Settings& s, const std::string& separator) {
    Catalogue cat(s.db);
    std::vector<int> rows;
    if (!s.query_file.empty()) {
        std::vector<std::string> labels;
        rows = pc_mat::read_queries_from_file(s.query_file, cat.index_of, labels);
    } else if (!s.query_ids.empty()) {
        for (const std::string& q : s.query_ids) {
            const int row = pc_mat::parse_query_to_index(q, cat.index_of);
            if (row >= 0) rows.push_back(row);
        }
    } else {
        abort_with("Error: No queries specified. Use --query_file, --query_ids");
    }
    if (rows.empty()) abort_with("Error: No valid queries found");
    cat.finish_loading(s.db);

    // per-query files land next to the requested name: <dir>/<query id>_<file name>
    const size_t cut = s.out_name.find_last_of("/\\");
    const std::string out_dir = cut == std::string::npos ? std::string("./") : s.out_name.substr(0, cut);
    const std::string out_leaf = cut == std::string::npos ? s.out_name : s.out_name.substr(cut + 1);

    Clock clock;
    in_batches(rows.size(), s.batch, "queries", clock, [&](size_t first, size_t last) {
        std::vector<int> part(rows.begin() + (std::ptrdiff_t)first, rows.begin() + (std::ptrdiff_t)last);
        const std::vector<pc_mat::Result> found =
            clock.timed([&] { return pc_mat::query(s.matrix, part, cat.norms, cat.names); });
        for (const pc_mat::Result& r : found) {
            const size_t have = r.neighbor_ids.size();
            const size_t shown = s.show_all ? have : std::min<size_t>(s.top, have);
            if (s.to_screen) std::cout << "Query: " << r.self_id << " #Neighbors: " << have << std::endl;
            std::ofstream file;
            if (s.to_file) {
                const std::string path = out_dir + "/" + r.self_id + "_" + out_leaf;
                std::cout << "Writing in file: " << path << std::endl << std::endl;
                file.open(path);
                file << "ID" << separator << "Jaccard\n";
            }
            if (s.to_screen) std::cout << "Top " << shown << " neighbors:\n";
            for (size_t k = 0; k < shown; ++k) {
                if (s.to_screen)
                    std::cout << k + 1 << ". Neighbor: " << r.neighbor_ids[k]
                              << " Jaccard Similarity: " << r.jaccard_similarities[k] << std::endl;
                if (s.to_file) file << r.neighbor_ids[k] << separator << r.jaccard_similarities[k] << std::endl;
            }
            if (s.to_screen) std::cout << std::endl;
        }
    });
}

// ---------------------------------------------------------------------------------------------------
// mode 2: rows x columns slice, through sinks
// ---------------------------------------------------------------------------------------------------
class SliceSink {
public:
    virtual ~SliceSink() = default;
    virtual void header(const std::vector<std::string>& columns) = 0;
    virtual void line(const std::string& row_label, const std::vector<float>& values) = 0;
    virtual void close() {}
};

class DelimitedSink : public SliceSink {   // screen (tab separated on std::cout) or a csv / tsv file
public:
    DelimitedSink(std::ostream& os, std::string sep, bool flush_lines)
        : os_(os), sep_(std::move(sep)), flush_(flush_lines) {}
    void header(const std::vector<std::string>& columns) override {
        os_ << "Accession" << sep_;
        for (const std::string& c : columns) os_ << c << sep_;
        os_ << "\n";
    }
    void line(const std::string& row_label, const std::vector<float>& values) override {
        os_ << row_label << sep_;
        for (const float v : values) os_ << v << sep_;
        if (flush_) os_ << std::endl;
        else os_ << "\n";
    }

private:
    std::ostream& os_;
    std::string sep_;
    bool flush_;
};

// float32 array of shape (rows, columns) in NumPy's .npy v1.0 container; the header is a fixed 128 bytes, so the row
// count can be filled in when the last row has been written
class NpySink : public SliceSink {
public:
    explicit NpySink(const std::string& path) : path_(path), f_(path, std::ios::binary | std::ios::trunc) {
        if (!f_) abort_with("Error: could not write " + path);
    }
    void header(const std::vector<std::string>& columns) override {
        cols_ = columns.size();
        write_header();
    }
    void line(const std::string&, const std::vector<float>& values) override {
        f_.write(reinterpret_cast<const char*>(values.data()), (std::streamsize)(values.size() * sizeof(float)));
        ++rows_;
    }
    void close() override {
        write_header();
        f_.flush();
        if (!f_) abort_with("Error: could not write " + path_);
    }

private:
    void write_header() {
        char text[119];
        const int used = std::snprintf(text, sizeof text, "{'descr': '<f4', 'fortran_order': False, 'shape': (%zu, %zu), }",
                                       rows_, cols_);
        std::string block("\x93NUMPY\x01\x00\x76\x00", 10);   // magic, version 1.0, header length 118
        block.append(text, (size_t)used);
        block.resize(127, ' ');
        block.push_back('\n');
        const auto back = f_.tellp();
        f_.seekp(0);
        f_.write(block.data(), 128);
        if (back > std::streampos(128)) f_.seekp(back);
    }
    std::string path_;
    std::ofstream f_;
    size_t rows_ = 0, cols_ = 0;
};

void slice_mode(const Settings& s, const std::string& ext) {
    Catalogue cat(s.db);
    std::vector<std::string> row_labels, col_labels;
    std::vector<int32_t> rows = pc_mat::read_queries_from_file(s.row_file, cat.index_of, row_labels);
    std::vector<int32_t> cols = pc_mat::read_queries_from_file(s.col_file, cat.index_of, col_labels);
    if (rows.empty() || cols.empty()) abort_with("Empty row or col accessions.");
    cat.finish_loading(s.db);

    std::vector<std::unique_ptr<SliceSink>> sinks;
    std::ofstream text_file;
    if (s.to_file && (ext == "csv" || ext == "tsv")) {
        std::cout << "Writing in file: " << s.out_name << std::endl << std::endl;
        text_file.open(s.out_name);
        sinks.push_back(std::make_unique<DelimitedSink>(text_file, ext == "csv" ? "," : "\t", false));
    } else if (s.to_file) {   // npy / npz: both get the .npy container, one array
        sinks.push_back(std::make_unique<NpySink>(s.out_name));
    }
    if (s.to_screen) sinks.push_back(std::make_unique<DelimitedSink>(std::cout, "\t", true));
    for (auto& sink : sinks) sink->header(col_labels);

    Clock clock;
    in_batches(rows.size(), s.batch, "rows", clock, [&](size_t first, size_t last) {
        std::vector<int32_t> part(rows.begin() + (std::ptrdiff_t)first, rows.begin() + (std::ptrdiff_t)last);
        const std::vector<std::vector<float>> values = clock.timed(
            [&] { return pc_mat::query_sliced(s.matrix, part, cols, (int32_t)cat.names.size(), cat.norms); });
        for (size_t i = 0; i < values.size(); ++i)
            for (auto& sink : sinks) sink->line(row_labels[first + i], values[i]);
    });
    for (auto& sink : sinks) sink->close();
}

}  // namespace

int main(int argc, char* argv[]) {
    Settings s;
    const bool fits = read_command_line(argc, argv, s);
    if (!fits || s.help) {
        print_help(argv[0]);
        return s.help ? 0 : 1;
    }
    if (s.matrix.empty()) abort_with("Error: matrix folder is required.");
    const bool neighbour_query = s.saw_query_file || s.saw_query_ids;
    if (!neighbour_query && !s.saw_row_file) abort_with("No query files given.");
    if (!std::filesystem::exists(s.matrix)) abort_with("Error: Matrix folder does not exist.");
    for (std::string* folder : {&s.matrix, &s.db})
        if (!folder->empty() && folder->back() != '/' && folder->back() != '\\') folder->push_back('/');
    if (s.to_file && s.out_name.empty()) abort_with("No output filename given.");
    if (s.batch == 0) s.batch = 1;
    if (!s.to_file) s.to_screen = true;   // with nowhere else to go the output is printed

    const std::string ext = extension_of(s.out_name);
    if (neighbour_query) {
        if (s.to_file && ext != "csv" && ext != "tsv" && ext != "txt")
            abort_with("Output file extension is: " + ext + ". Expected: csv, tsv or txt.");
        neighbour_mode(s, ext == "csv" ? "," : "\t");
    } else {
        if (s.row_file.empty() || s.col_file.empty()) abort_with("Either row or col file is not specified.");
        if (s.to_file && ext != "csv" && ext != "tsv" && ext != "npy" && ext != "npz")
            abort_with("Output file extension is: " + ext + ". Expected: csv, tsv, npy or npz.");
        slice_mode(s, ext);
    }
    return 0;
}
