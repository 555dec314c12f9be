// project_everything -- drop-in for the reference executable of the same name
// (src/project_everything.cpp).  `sketch` runs the projection on the MI355X through libmvs_hip.so;
// the command line, stdout lines and the four DB files are the reference's.
//
//   project_everything sketch  <hash_file> <index_folder> [-t/--threads N] [-d/--dimension D] [--int16]
//   project_everything convert <signature_folder> <hash_file> [-t/--threads N]      (host only, zlib)
#include <atomic>
#include <chrono>
#include <thread>
#include <mutex>

#include "mvs_host.hpp"
#include "mvs_ingest.hpp"

namespace fs = std::filesystem;
using namespace mvs_host;

static void usage(const char* argv0) {   // src/project_everything.cpp:394-407, verbatim layout
    std::cerr << "Usage:\n";
    std::cerr << "  Convert mode:\n";
    std::cerr << "    " << argv0 << " convert <signature_folder> <hash_file> [-t threads]\n";
    std::cerr << "      signature_folder : Path to folder containing signature files\n";
    std::cerr << "      hash_file        : Output hash file path\n";
    std::cerr << "      -t, --threads    : Number of threads (default: 1)\n\n";
    std::cerr << "  Sketch mode:\n";
    std::cerr << "    " << argv0 << " sketch <hash_file> <index_folder> [-t threads] [-d dimension] [--int16]\n";
    std::cerr << "      hash_file        : Input hash file path\n";
    std::cerr << "      index_folder     : Output folder for index files\n";
    std::cerr << "      -t, --threads    : Number of threads (default: 1)\n";
    std::cerr << "      -d, --dimension  : Vector dimension (default: 2048)\n";
    std::cerr << "      --int16          : Use int16 instead of int32 for vector storage\n";
}

// src/project_everything.cpp:181-235.  Differences: archives are read in-process (no unzip/gunzip, no
// /tmp/signature_extract*), hashes are written sorted, and the reference's side file ./all_hashes.txt
// (appended in the current directory, :160-176) is not produced.
static int convert(const std::string& folder_name, const std::string& output_file, int num_threads) {
    auto start = std::chrono::high_resolution_clock::now();
    std::vector<std::string> sig_files;
    std::error_code ec;
    for (const auto& entry : fs::directory_iterator(folder_name, ec)) sig_files.push_back(entry.path().string());
    if (ec) {
        std::cerr << "Error reading folder " << folder_name << std::endl;
        return 0;
    }
    std::ofstream hash_out(output_file);
    if (!hash_out) {
        std::cerr << "Error opening " << output_file << " for writing." << std::endl;
        return 0;
    }
    std::vector<std::pair<std::string, std::vector<uint64_t>>> results(sig_files.size());
    std::mutex io;
    const unsigned nt = (unsigned)std::max(1, std::min<int>(num_threads, (int)std::max<size_t>(1, sig_files.size())));
    std::vector<std::thread> pool;
    for (unsigned t = 0; t < nt; ++t)
        pool.emplace_back([&, t]() {
            for (size_t i = t; i < sig_files.size(); i += nt) {
                std::vector<uint64_t> hashes;
                if (!mvs_ingest::load_signatures(sig_files[i], hashes)) {
                    std::lock_guard<std::mutex> g(io);
                    std::cerr << "Failed to unzip: " << sig_files[i] << std::endl;
                }
                const std::string stem = fs::path(sig_files[i]).stem().string();
                results[i] = {stem.substr(0, stem.find('.')), std::move(hashes)};
                std::lock_guard<std::mutex> g(io);
                std::cout << "Processed " << sig_files[i] << ", hashes size " << results[i].second.size()
                          << ", file number " << i << std::endl;
            }
        });
    for (auto& th : pool) th.join();
    std::string line;
    for (const auto& r : results) {
        line = r.first + ":";
        for (uint64_t h : r.second) {
            line += ' ';
            line += std::to_string(h);
        }
        line += '\n';
        hash_out << line;
    }
    hash_out.close();
    // the binary form `sketch` would otherwise have to parse back out of the text (hash_file.csr, see mvs_host.hpp)
    if (hash_out && !getenv("MVS_NO_CSR_CACHE")) {
        HashSets sets;
        bool plain_names = true;
        size_t total = 0;
        for (auto& r : results) {
            std::sort(r.second.begin(), r.second.end());
            r.second.erase(std::unique(r.second.begin(), r.second.end()), r.second.end());
            total += r.second.size();
            plain_names = plain_names && r.first.find_first_of(":\n") == std::string::npos;
        }
        if (plain_names && sets.hashes.reset(total)) {
            sets.offsets.assign(1, 0);
            for (const auto& r : results) {
                sets.names.push_back(r.first);
                if (!r.second.empty()) memcpy(sets.hashes.data() + sets.offsets.back(), r.second.data(), r.second.size() * 8);
                sets.offsets.push_back(sets.offsets.back() + (int64_t)r.second.size());
            }
            (void)write_csr_cache(output_file, sets);
        }
    }
    auto end = std::chrono::high_resolution_clock::now();
    std::chrono::duration<double> elapsed = end - start;
    std::cout << "Time to convert all signatures: " << elapsed.count() << " seconds" << std::endl;
    return 0;
}

static bool parse_int(const char* s, int& out) {
    char* end = nullptr;
    const long v = strtol(s, &end, 10);
    if (end == s || *end != '\0') return false;
    out = (int)v;
    return true;
}

// src/project_everything.cpp:238-362
static int sketch(const std::string& hash_file, std::string index_folder, int dimension, bool use_int16) {
    if (index_folder.empty() || index_folder[index_folder.size() - 1] != '/') index_folder += '/';   // :239-241
    if (fs::exists(index_folder)) {                                                                    // :244-252
        for (const auto& entry : fs::directory_iterator(index_folder)) fs::remove_all(entry.path());
    } else {
        fs::create_directories(index_folder);
    }
    auto start = std::chrono::high_resolution_clock::now();                                            // :255
    // MVS_STAGE_TIMING=1: per-stage wall times on stderr (not part of the reference's output)
    const bool stage_timing = getenv("MVS_STAGE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t = std::chrono::steady_clock::now();
        if (stage_timing)
            std::cerr << "[stage] " << what << " " << std::chrono::duration<double>(t - lap_t).count() << " s" << std::endl;
        lap_t = t;
    };

    // The device contexts (0.1-0.2 s of runtime start-up each) come up while the text is parsed: one per visible GPU --
    // the reference's OpenMP loop over samples (:289-298) becomes one host thread per GPU, each projecting a contiguous
    // range of samples with about the same number of hashes.  MVS_DEVICE pins the work to one device;
    // MVS_SKETCH_CONTEXTS=k asks for k contexts (cycling over the devices: k = 2 on a one-GPU box runs the multi-context
    // path on device 0).
    std::vector<int> devices;
    {
        int ndev = 0;
        if (getenv("MVS_DEVICE") || mvs_device_count(&ndev) != MVS_OK || ndev <= 0) {
            devices.push_back(pick_device());
        } else {
            int want = ndev;
            if (const char* e = getenv("MVS_SKETCH_CONTEXTS")) want = std::max(1, std::min(64, atoi(e)));
            for (int i = 0; i < want; ++i) devices.push_back(i % ndev);
        }
    }
    const size_t n_ctx = devices.size();
    std::vector<mvs_ctx*> ctxs(n_ctx, nullptr);
    std::vector<int> ctx_rcs(n_ctx, MVS_OK);
    std::vector<std::string> ctx_errs(n_ctx);
    auto destroy_all = [&]() {
        for (mvs_ctx*& c : ctxs) {
            if (c) mvs_ctx_destroy(c);
            c = nullptr;
        }
    };
    std::thread ctx_thread([&]() {
        std::vector<std::thread> pool;
        for (size_t g = 0; g < n_ctx; ++g)
            pool.emplace_back([&, g]() {
                ctx_rcs[g] = mvs_ctx_create(devices[g], &ctxs[g]);
                if (ctx_rcs[g] != MVS_OK) ctx_errs[g] = mvs_last_error();   // the message is per thread
            });
        for (auto& th : pool) th.join();
    });
    // the parsed form of an unchanged hash file is kept next to it (<hash_file>.csr): mapped instead of parsed again
    HashSets sets;
    struct Joiner {                       // declared after `sets`: joined before `sets` goes away
        std::thread th;
        ~Joiner() { if (th.joinable()) th.join(); }
    } cache_writer;
    bool parsed = load_csr_cache(hash_file, sets);
    if (parsed && sets.hashes.size() > (1u << 24)) {
        // page the mapping in on all host threads now (the context needs ~0.1 s anyway) instead of one fault at a
        // time inside the upload
        const size_t bytes = sets.hashes.size() * 8;
        const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
        std::vector<std::thread> pool;
        std::atomic<uint64_t> sink{0};
        for (unsigned t = 0; t < nt; ++t)
            pool.emplace_back([&, t]() {
                const volatile char* b = reinterpret_cast<const volatile char*>(sets.hashes.data());
                uint64_t acc = 0;
                for (size_t o = bytes / nt * t, e = t + 1 == nt ? bytes : bytes / nt * (t + 1); o < e; o += 4096) acc += (uint64_t)b[o];
                sink += acc;
            });
        for (auto& th : pool) th.join();
    }
    if (!parsed) {
        try {
            // <hash_file>.csr is written while the text is parsed (mvs_host.hpp; 4 GB for 10k x 50k hashes: started after
            // the parse, that copy into the page cache was what a first run waited for at its end); what is left of it
            // runs beside the projection, `cache_writer` is joined on every way out
            parsed = read_hash_file(hash_file, true, sets, 0, getenv("MVS_NO_CSR_CACHE") ? std::string() : hash_file,
                                    &cache_writer.th);
        } catch (const std::exception& e) {
            ctx_thread.join();
            std::cerr << "project_everything: reading " << hash_file << ": " << e.what() << std::endl;
            destroy_all();
            return 2;
        }
    }
    ctx_thread.join();
    if (!parsed) {                                                                                     // :258-262
        std::cerr << "Error opening " << hash_file << " for reading." << std::endl;
        destroy_all();
        return 0;   // the reference returns from sketch() and exits 0
    }
    lap("hash sets (text or cache) + device context");
    const int64_t n = (int64_t)sets.names.size();
    std::cout << "Loaded " << n << " hash sets from " << hash_file << std::endl;                       // :284
    for (size_t g = 0; g < n_ctx; ++g)
        if (ctx_rcs[g] != MVS_OK) {
            std::cerr << "project_everything: " << ctx_errs[g] << std::endl;
            destroy_all();
            return 2;
        }
    mvs_ctx* ctx = ctxs[0];
    std::vector<int32_t> vectors((size_t)n * (size_t)dimension);
    std::vector<int64_t> sumsq((size_t)n);
    // sample ranges with about the same number of hashes each: range g ends where the running hash count passes g + 1 shares
    std::vector<int64_t> cut(n_ctx + 1, n);
    cut[0] = 0;
    {
        const int64_t total_hashes = sets.offsets[(size_t)n];
        for (size_t g = 1; g < n_ctx; ++g) {
            const int64_t target = (int64_t)((double)total_hashes * (double)g / (double)n_ctx);
            cut[g] = std::lower_bound(sets.offsets.begin(), sets.offsets.begin() + n, target) - sets.offsets.begin();
            cut[g] = std::max(cut[g], cut[g - 1]);
        }
    }
    // batches bound the device footprint (hash lists of ~1M-hash samples x thousands of samples)
    const int64_t kMaxBatchHashes = 1LL << 28;   // 2 GiB of hashes per launch
    std::vector<std::string> errors(n_ctx);
    auto project_range = [&](size_t g) {
        for (int64_t s0 = cut[g]; s0 < cut[g + 1];) {
            int64_t s1 = s0 + 1;
            while (s1 < cut[g + 1] && sets.offsets[s1 + 1] - sets.offsets[s0] <= kMaxBatchHashes) ++s1;
            std::vector<int64_t> offs((size_t)(s1 - s0 + 1));
            for (int64_t s = s0; s <= s1; ++s) offs[(size_t)(s - s0)] = sets.offsets[s] - sets.offsets[s0];
            int64_t max_abs = 0;   // not needed here; the statistics come out of the projection kernel for free
            const int rc = mvs_project_csr_stats(ctxs[g], sets.hashes.data() + sets.offsets[s0], MVS_MEM_HOST, offs.data(), s1 - s0,
                                                 dimension, vectors.data() + (size_t)s0 * dimension, MVS_MEM_HOST,
                                                 sumsq.data() + s0, &max_abs);
            if (rc != MVS_OK) {
                errors[g] = mvs_last_error();
                return;
            }
            s0 = s1;
        }
    };
    if (n_ctx == 1) {
        project_range(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t g = 0; g < n_ctx; ++g) pool.emplace_back(project_range, g);
        for (auto& th : pool) th.join();
    }
    for (size_t g = 0; g < n_ctx; ++g)
        if (!errors[g].empty()) {
            std::cerr << "project_everything: " << errors[g] << std::endl;
            destroy_all();
            return 2;
        }
    if (stage_timing && n_ctx > 1) {
        std::cerr << "[stage] " << n_ctx << " contexts, samples per context:";
        for (size_t g = 0; g < n_ctx; ++g) std::cerr << " " << (cut[g + 1] - cut[g]) << " (device " << devices[g] << ")";
        std::cerr << std::endl;
    }
    lap("projection (upload, kernels, download)");
    for (int64_t i = 0; i < n; ++i)                                                                    // :294-297
        std::cout << "Projected " << sets.names[(size_t)i] << ", vector dimension " << dimension << ", index " << i
                  << "\n";
    std::cout.flush();

    auto end = std::chrono::high_resolution_clock::now();                                              // :301-303
    std::chrono::duration<double> elapsed = end - start;
    std::cout << "Time to compute all projected vectors: " << elapsed.count() << " seconds" << std::endl;

    std::ofstream norm_out(index_folder + "vector_norms.txt");                                         // :306-309
    std::ofstream dim_out(index_folder + "dimension.txt");
    std::ofstream dtype_out(index_folder + "dtype.txt");
    std::ofstream bin_out(index_folder + "vectors.bin", std::ios::binary);
    if (!norm_out) std::cerr << "Error opening vector_norms.txt for writing." << std::endl;
    if (!bin_out) std::cerr << "Error opening vectors.bin for writing." << std::endl;
    int status = 0;
    if (norm_out && bin_out && dim_out && dtype_out) {
        dim_out << dimension << "\n";                                                                  // :319
        dtype_out << (use_int16 ? "int16" : "int32") << "\n";                                          // :320
        // MVS_NORM_FLOAT32=1: the reference's float32 evaluation (`vec.cast<float>() / sqrt(dimension)`, then Eigen's
        // float norm, :328-329) in sequential order instead of this build's sqrt(double(sum v^2) / d).  The
        // reference's own last digit depends on its vectorised summation order and -ffast-math, so this
        // is the closest defined stand-in, not a bit-exact reproduction.
        const char* nf = getenv("MVS_NORM_FLOAT32");
        const bool norm_f32 = nf && nf[0] == '1';
        for (int64_t i = 0; i < n; ++i) {                                                              // :328-330
            double norm = norm_from_sumsq(sumsq[(size_t)i], dimension);
            if (norm_f32) norm = (double)norm_float32_path(vectors.data() + (size_t)i * (size_t)dimension, dimension);
            norm_out << sets.names[(size_t)i] << " " << format_g(norm) << "\n";
        }
        if (use_int16) {                                                                               // :332-347
            std::vector<int16_t> v16(vectors.size());
            if (mvs_sketch_saturate_i16(ctx, vectors.data(), MVS_MEM_HOST, (int64_t)vectors.size(), v16.data(),
                                        MVS_MEM_HOST) != MVS_OK) {
                std::cerr << "project_everything: " << mvs_last_error() << std::endl;
                status = 2;
            } else {
                bin_out.write(reinterpret_cast<const char*>(v16.data()), (std::streamsize)(v16.size() * 2));
            }
        } else {                                                                                       // :348-354
            bin_out.write(reinterpret_cast<const char*>(vectors.data()), (std::streamsize)(vectors.size() * 4));
        }
    }
    lap("progress lines + DB files");
    destroy_all();
    lap("context teardown");
    return status;
}

int main(int argc, char* argv[]) {
    bool is_convert = false, is_sketch = false, use_int16 = false, ok = argc >= 2;
    std::string input_path, output_path;
    int t = 1, d = 2048;
    if (ok) {
        const std::string cmd = argv[1];
        is_convert = cmd == "convert";
        is_sketch = cmd == "sketch";
        ok = is_convert || is_sketch;
    }
    int positional = 0;
    for (int i = 2; ok && i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "-t" || a == "--threads") {
            ok = i + 1 < argc && parse_int(argv[++i], t);
        } else if (is_sketch && (a == "-d" || a == "--dimension")) {
            ok = i + 1 < argc && parse_int(argv[++i], d);
        } else if (is_sketch && a == "--int16") {
            use_int16 = true;
        } else if (positional == 0) {
            input_path = a;
            ++positional;
        } else if (positional == 1) {
            output_path = a;
            ++positional;
        } else {
            ok = false;
        }
    }
    if (!ok || positional != 2) {   // src/project_everything.cpp:393-408
        usage(argv[0]);
        return 1;
    }
    if (is_convert) return convert(input_path, output_path, t);
    (void)t;   // the reference parses -t for sketch but never applies it (:384 vs :238)
    if (d <= 0) {
        std::cerr << "dimension must be positive" << std::endl;
        return 1;
    }
    return sketch(input_path, output_path, d, use_int16);
}
