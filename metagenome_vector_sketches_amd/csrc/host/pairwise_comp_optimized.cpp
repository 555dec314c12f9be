// pairwise_comp_optimized -- drop-in for the reference executable of the same name
// (src/pairwise_comp_optimized.cpp + src/pairwise_comp_optimized_16bits.cpp): same six required flags,
// same DB folder in, same shard folder out; the all-vs-all comparison runs on the MI355X.
//
//   pairwise_comp_optimized --db D/ --max_memory_gb G --num_threads T --output_folder O
//                           --num_shards S --shard_idx k [--start_shard a] [--end_shard b] [--help]
//   (extension: --shard_idx -1 computes ALL S shards from this one process on all visible GPUs)
//
// Multi-GPU: the reference runs one process per shard and every process re-reads the whole vectors.bin and compares
// its rows against ALL columns (src/pairwise_comp_optimized.cpp:937-982).  Started the same way -- one process per shard,
// --shard_idx k on GPU k % device_count -- with MVS_COLLECTIVE=rccl in the environment, the shard processes run ONE
// strong-scaled step between them (mvs_step.hpp: ShardStep): every process loads only the rows of ITS shard, the filter's
// inputs and the low limbs travel over RCCL beside the comparison, every unordered pair of row blocks is compared by
// exactly one process and the mirrored cells are exchanged; the processes find each other through
// <output_folder>/.mvs_comm_<MVS_COLLECTIVE_TOKEN>.  MVS_COLLECTIVE=files does the exchange through files in the output
// folder instead (ranks sharing one GPU).  --shard_idx -1 runs the same step from ONE process: a host thread + device
// context per GPU, each rank owning --num_shards / G consecutive shards whose rows it compares once, as one block.
// MVS_STEP=0 keeps the round-1 scheme (limb planes gathered whole, rows x all columns per shard); a result too dense for
// cell lists falls back to it by itself (every rank reads that in the exchanged headers).
//
// Differences a user can observe (all listed in DESIGN.md): tiles are sized by the kernel, not by
// --max_memory_gb (the "Using chunks of size" line still prints the reference's formula; the flag bounds
// the kept-cell staging buffer instead); rows are written in ascending order; the codec bytes are this
// build's own (the reference's `bits` library is not available); the int16 DB path writes the active
// shard format instead of the legacy EF+zstd one.
#include <chrono>
#include <functional>
#include <memory>

#include "mvs_host.hpp"
#include "mvs_step.hpp"

namespace fs = std::filesystem;
using namespace mvs_host;

struct Options {
    std::string db_folder, output_folder;
    double max_memory_gb = 0.0;
    int num_threads = 1, num_shards = 1, shard_idx = 0, start_shard = 0, end_shard = 1;
    bool show_help = false;
};

static void print_usage(const char* argv0) {
    std::cout << "Usage:\n"
              << "        " << argv0
              << " --db <folder> --max_memory_gb <float> --num_threads <int> --output_folder <folder>"
                 " --num_shards <int> --shard_idx <int> [--start_shard <int>] [--end_shard <int>] [--help]"
              << std::endl;
}

static bool parse(int argc, char* argv[], Options& o) {
    bool have[6] = {false, false, false, false, false, false};
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        auto value = [&](std::string& dst) {
            if (i + 1 >= argc) return false;
            dst = argv[++i];
            return true;
        };
        std::string v;
        char* end = nullptr;
        if (a == "--help") {
            o.show_help = true;
        } else if (a == "--db") {
            if (!value(o.db_folder)) return false;
            have[0] = true;
        } else if (a == "--max_memory_gb") {
            if (!value(v)) return false;
            o.max_memory_gb = strtod(v.c_str(), &end);
            if (end == v.c_str() || *end) return false;
            have[1] = true;
        } else if (a == "--output_folder") {
            if (!value(o.output_folder)) return false;
            have[3] = true;
        } else if (a == "--num_threads" || a == "--num_shards" || a == "--shard_idx" || a == "--start_shard" ||
                   a == "--end_shard") {
            if (!value(v)) return false;
            const long x = strtol(v.c_str(), &end, 10);
            if (end == v.c_str() || *end) return false;
            if (a == "--num_threads") { o.num_threads = (int)x; have[2] = true; }
            else if (a == "--num_shards") { o.num_shards = (int)x; have[4] = true; }
            else if (a == "--shard_idx") { o.shard_idx = (int)x; have[5] = true; }
            else if (a == "--start_shard") o.start_shard = (int)x;
            else o.end_shard = (int)x;
        } else {
            return false;
        }
    }
    for (bool h : have)
        if (!h) return false;
    return true;
}

struct Gpu {
    mvs_ctx* ctx = nullptr;
    mvs_sketch_set* set = nullptr;
    mvs_comm* comm = nullptr;
    ~Gpu() {
        if (comm) mvs_comm_destroy(comm);
        if (set) mvs_sketch_set_destroy(set);
        if (ctx) mvs_ctx_destroy(ctx);
    }
};

// MVS_INT16_LEGACY_OUTPUT=1: an int16 DB's shard in the reference's legacy format (elias_fano columns +
// round(dot / d) values + zstd, _16bits.cpp:251-323) instead of the active, queryable one
static bool legacy16_output() {
    const char* e = getenv("MVS_INT16_LEGACY_OUTPUT");
    return e && e[0] == '1';
}

static int gpu_fail(const char* what) {
    std::cerr << "pairwise_comp_optimized: " << what << ": " << mvs_last_error() << std::endl;
    return 2;
}

// vectors.bin goes through the device in row chunks straight from the page cache (the file is mapped, the
// library copies from the mapping) and is re-coded into the limb planes
static int load_db(Gpu& g, const std::string& matrix_file, int elem_bytes, int64_t n, int d) {
    const int64_t row_bytes = (int64_t)d * elem_bytes;
    const int64_t chunk_rows = std::max<int64_t>(1, (1LL << 30) / row_bytes);
    const int fd = ::open(matrix_file.c_str(), O_RDONLY);
    if (fd < 0) {
        std::cerr << "Error opening file: " << matrix_file << std::endl;       // :35-38
        return 1;
    }
    const size_t bytes = (size_t)(n * row_bytes);
    const char* base = nullptr;
    if (bytes) {
        void* m = ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            ::close(fd);
            std::cerr << "Error reading file: " << matrix_file << std::endl;
            return 1;
        }
        ::madvise(m, bytes, MADV_SEQUENTIAL);
        base = (const char*)m;
    }
    ::close(fd);
    // One pass in the common case: the planes are allocated for two limbs (|v| <= 32639, i.e. samples of up to tens
    // of millions of hashes) and every chunk reports its largest |v| with the same upload; only if a chunk needs
    // more limbs than the set has does the load start over with the limb count the data seen so far asks for.
    int rc = 0;
    int limbs = 2;
    for (int attempt = 0; attempt < 4 && !rc; ++attempt) {
        if (g.set) {
            mvs_sketch_set_destroy(g.set);
            g.set = nullptr;
        }
        if (mvs_sketch_set_alloc(g.ctx, n, d, limbs, &g.set) != MVS_OK) {
            rc = gpu_fail("allocating sketch set");
            break;
        }
        int64_t max_abs = 0;
        bool restart = false;
        for (int64_t r0 = 0; r0 < n && !rc; r0 += chunk_rows) {
            const int64_t rows = std::min(chunk_rows, n - r0);
            int64_t m = 0;
            if (mvs_sketch_set_fill_stats(g.set, base + r0 * row_bytes, elem_bytes, MVS_MEM_HOST, r0, rows, &m) != MVS_OK)
                rc = gpu_fail("re-coding vectors.bin");
            max_abs = std::max(max_abs, m);
            if (mvs_limbs_for_max_abs(max_abs) > limbs) {
                restart = true;
                break;
            }
        }
        if (!restart) break;
        limbs = mvs_limbs_for_max_abs(max_abs);
    }
    if (bytes) ::munmap((void*)base, bytes);
    return rc;
}

// Rendezvous of the shard processes of one job: they meet under <output_folder>/.mvs_comm_<token> through the
// library's file handshake (a job nonce all ranks agree on; files an earlier job left there are never taken for this
// job's), and with MVS_COLLECTIVE=rccl rank 0's RCCL id travels as one verified block of that transport.
static std::string comm_base_path(const std::string& output_folder, const std::string& fallback_token) {
    const char* tok = getenv("MVS_COLLECTIVE_TOKEN");
    return output_folder + ".mvs_comm_" + (tok ? std::string(tok) : fallback_token);
}
static int open_communicator(Gpu& g, const std::string& kind, const std::string& base, int rank, int world) {
    if (kind == "files") {
        if (mvs_comm_create_files(g.ctx, base.c_str(), rank, world, &g.comm) != MVS_OK) return gpu_fail("file communicator");
        return 0;
    }
    if (mvs_comm_create_rendezvous(g.ctx, base.c_str(), rank, world, &g.comm) != MVS_OK) return gpu_fail("RCCL communicator");
    return 0;
}

// Ranks leave together: before a collective every rank contributes its status, and all of them learn the worst one
// (a rank that returned early on its own would leave the others inside RCCL for ever -- it has no timeout).  The
// status rides in the high bits of the value the ranks reduce anyway (max|v| < 2^40).
static int agree(Gpu& g, int my_rc, int64_t* value) {
    int64_t v = ((int64_t)(my_rc != 0) << 40) | (value ? (*value & ((1LL << 40) - 1)) : 0);
    if (mvs_allreduce_max_i64(g.ctx, g.comm, &v) != MVS_OK) return gpu_fail("all-reduce of status / max|v|");
    if (value) *value = v & ((1LL << 40) - 1);
    if ((v >> 40) != 0) {
        if (!my_rc) std::cerr << "pairwise_comp_optimized: another shard process failed; leaving with it" << std::endl;
        return my_rc ? my_rc : 3;
    }
    return 0;
}

// Collective load: this process brings rows [b, e) of vectors.bin; the all-gather brings the rest.
static int load_db_collective(Gpu& g, const std::string& matrix_file, int elem_bytes, int64_t n, int d, int64_t b, int64_t e,
                              int world) {
    const int64_t row_bytes = (int64_t)d * elem_bytes;
    const int64_t rps = (n + world - 1) / world;
    int rc = 0, limbs = 2;              // a failure before the first collective is carried into it (agree), not returned
    const int fd = ::open(matrix_file.c_str(), O_RDONLY);
    if (fd < 0) {
        std::cerr << "Error opening file: " << matrix_file << std::endl;
        rc = 1;
    }
    const size_t bytes = (size_t)((e - b) * row_bytes);
    const char* base = nullptr;
    const size_t map_off = (size_t)(b * row_bytes) & ~(size_t)4095, map_len = (size_t)(b * row_bytes) - map_off + bytes;
    void* m = nullptr;
    if (bytes && !rc) {
        m = ::mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, (off_t)map_off);
        if (m == MAP_FAILED) {
            m = nullptr;
            std::cerr << "Error reading file: " << matrix_file << std::endl;
            rc = 1;
        } else {
            ::madvise(m, map_len, MADV_SEQUENTIAL);
            base = (const char*)m + ((size_t)(b * row_bytes) - map_off);
        }
    }
    if (fd >= 0) ::close(fd);
    const int64_t chunk_rows = std::max<int64_t>(1, (1LL << 30) / row_bytes);
    for (int attempt = 0; attempt < 4; ++attempt) {
        if (g.set) {
            mvs_sketch_set_destroy(g.set);
            g.set = nullptr;
        }
        int64_t n_alloc = 0, max_abs = 0;
        int d_pad = 0;
        if (rc) {
            // failed before the loop: only the agreement below is left to do
        } else if (mvs_sketch_set_alloc(g.ctx, n, d, limbs, &g.set) != MVS_OK) {
            rc = gpu_fail("allocating sketch set");
        } else {
            mvs_sketch_set_info(g.set, nullptr, nullptr, nullptr, &n_alloc, &d_pad);
            if (rps * world > n_alloc) {
                std::cerr << "pairwise_comp_optimized: too many shards for the plane padding" << std::endl;
                rc = 1;
            }
        }
        for (int64_t r0 = b; r0 < e && !rc; r0 += chunk_rows) {
            const int64_t rows = std::min(chunk_rows, e - r0);
            int64_t mx = 0;
            if (mvs_sketch_set_fill_stats(g.set, base + (r0 - b) * row_bytes, elem_bytes, MVS_MEM_HOST, r0, rows, &mx) != MVS_OK)
                rc = gpu_fail("re-coding vectors.bin");
            max_abs = std::max(max_abs, mx);
        }
        // one limb code for everybody: the largest |v| over all shards decides (every rank takes the same branch),
        // and a rank that failed above takes everybody out with it
        rc = agree(g, rc, &max_abs);
        if (rc) break;
        if (mvs_limbs_for_max_abs(max_abs) > limbs) {
            limbs = mvs_limbs_for_max_abs(max_abs);
            continue;
        }
        int8_t* planes = nullptr;
        if (mvs_sketch_set_planes(g.set, &planes) != MVS_OK || mvs_allgather_planes(g.ctx, g.comm, planes, rps, limbs, d_pad) != MVS_OK ||
            mvs_sketch_set_touch(g.set) != MVS_OK || mvs_ctx_synchronize(g.ctx) != MVS_OK)
            rc = gpu_fail("all-gather of the limb planes");
        break;
    }
    if (m) ::munmap(m, map_len);
    return rc;
}

// Kept-cell staging: grows to what a call reports it needs, up to `limit` cells; never value-initialised
// (the budget can be gigabytes).
struct Staging {
    std::unique_ptr<mvs_cell[]> p;
    size_t cap = 0, limit = 0;
    void reserve(size_t want) {
        if (want <= cap) return;
        p.reset();
        p.reset(new mvs_cell[want]);
        cap = want;
    }
};

// Legacy int16 output only: rows [b, e) against all columns as a list of cells WITH their dot products, in row blocks
// whose worst case (every cell kept) fits the staging buffer -- a call can then never report MVS_E_CAPACITY, so nothing
// is ever compared twice; the price is the symmetric schedule across blocks (each block still uses it inside its own
// square).
static int compare_rows(Gpu& g, const std::vector<double>& n2, int keep_mode, int64_t b, int64_t e, int64_t n_total,
                        Staging& st, std::vector<mvs_cell>& all) {
    const int64_t block = std::max<int64_t>(1, (int64_t)(st.limit / (size_t)std::max<int64_t>(n_total, 1)));
    st.reserve((size_t)(std::min(block, std::max<int64_t>(e - b, 1)) * n_total));
    for (int64_t rb = b; rb < e; rb += block) {
        const int64_t re = std::min(e, rb + block);
        int64_t cnt = 0;
        if (mvs_pairwise_rows(g.ctx, g.set, n2.data(), MVS_MEM_HOST, keep_mode, rb, re, st.p.get(), (int64_t)st.cap,
                              MVS_MEM_HOST, &cnt) != MVS_OK)
            return gpu_fail("pairwise comparison");
        all.insert(all.end(), st.p.get(), st.p.get() + cnt);
    }
    return 0;
}

// what the streamed pieces -- of a comparison (mvs_pairwise_stream*) or of a step's sorted cells (mvs_cells_stream*) -- go into
struct ShardSink {
    ShardWriter writer;
    std::string error;
    explicit ShardSink(const std::string& folder) : writer(folder) {}
    static int on_block(void* user, const mvs_row_block* b) {
        ShardSink* self = static_cast<ShardSink*>(user);
        try {
            self->writer.add(*b);
            return 0;
        } catch (const std::exception& e) {
            self->error = e.what();
            return 1;
        }
    }
    static int on_encoded(void* user, const mvs_encoded_rows* b) {
        ShardSink* self = static_cast<ShardSink*>(user);
        try {
            self->writer.add_encoded(*b);
            return 0;
        } catch (const std::exception& e) {
            self->error = e.what();
            return 1;
        }
    }
};
static bool host_encoder_wanted() {
    const char* enc = getenv("MVS_SHARD_ENCODER");
    return enc && std::string(enc) == "host";
}

// One shard: rows [begin_row, end_row) against all columns, streamed into <shard_folder>.  The comparison hands over CSR
// pieces of whole rows (ascending) -- by default with the rows already encoded in the shard codec ON THE DEVICE (1.4 bytes
// per kept cell on the link instead of 5, no host thread touches a cell); MVS_SHARD_ENCODER=host takes (column, q) pieces
// and encodes them on the host threads instead: the same files, byte for byte.  Where the reference keeps every kept cell
// of the shard in RAM (`all_results`, :974-980) this holds two pinned pieces and the per-row directory.
static int stream_shard(Gpu& g, const std::vector<double>& norms_sq, int keep_mode, int64_t begin_row, int64_t end_row,
                        const std::string& shard_folder, bool stage_timing, int64_t* n_kept, ShardStats* stats,
                        const std::function<void(const char*)>& lap) {
    ShardSink sink(shard_folder);
    const bool host_encoder = host_encoder_wanted();
    const int stream_rc =
        host_encoder ? mvs_pairwise_stream(g.ctx, g.set, norms_sq.data(), MVS_MEM_HOST, keep_mode, begin_row, end_row, 0,
                                           &ShardSink::on_block, &sink, n_kept)
                     : mvs_pairwise_stream_encoded(g.ctx, g.set, norms_sq.data(), MVS_MEM_HOST, keep_mode, begin_row, end_row, 0,
                                                   &ShardSink::on_encoded, &sink, n_kept);
    if (stream_rc != MVS_OK) {
        if (!sink.error.empty()) std::cerr << "pairwise_comp_optimized: " << sink.error << std::endl;
        return gpu_fail("pairwise comparison");
    }
    lap("compare + shard rows (streamed)");
    if (stage_timing) {
        double kms = 0.0;
        int64_t bytes = 0, blocks = 0, pieces = 0;
        int two = 0;
        mvs_ctx_stream_stats(g.ctx, &kms, &bytes, &blocks, &pieces, &two);
        std::cerr << "[stream] comparison kernels " << kms << " ms in " << blocks << " row block(s) ("
                  << (two ? "two-stage" : "exact kernel") << "), " << *n_kept << " kept cells = " << bytes << " bytes in "
                  << pieces << " piece(s), rows encoded on the " << (host_encoder ? "host" : "device") << std::endl;
    }
    try {
        *stats = sink.writer.finish();                                                              // :990
    } catch (const std::exception& e) {
        std::cerr << "pairwise_comp_optimized: " << e.what() << std::endl;
        return 2;
    }
    lap("shard index files");
    return 0;
}


// -------------------------------------------------------------------------------------------------------------------
// the strong-scaled step under the reference's command line (mvs_step.hpp)
// -------------------------------------------------------------------------------------------------------------------
// One rank of a job of `world` ranks: it owns shards [shard_begin, shard_end) -- consecutive, the same number on every rank --,
// loads exactly their rows of vectors.bin, takes part in ONE step and writes its shard folders.
struct StepRank {
    int device = 0, rank = 0, world = 1, shard_begin = 0, shard_end = 1;
};
struct StepJob {
    const Options* o = nullptr;
    std::string output_folder, matrix_file, comm_kind, comm_base;
    int elem_bytes = 4, dimension = 0, keep_mode = MVS_KEEP_INT32;
    int64_t total_vectors = 0;
    const std::vector<double>* norms_sq = nullptr;
    bool stage_timing = false;
    std::mutex* out_mu = nullptr;
};

// rows [b, e) of vectors.bin -> device memory (chunks of the mapping; the file's pages are read once, by this rank only)
static int upload_rows(mvs_ctx* ctx, const std::string& matrix_file, int elem_bytes, int d, int64_t b, int64_t e, mvs_step::DevMem& out,
                       int64_t* max_abs) {
    *max_abs = 0;
    const int64_t row_bytes = (int64_t)d * elem_bytes;
    const size_t bytes = (size_t)((e - b) * row_bytes);
    if (!bytes) return 0;
    const int fd = ::open(matrix_file.c_str(), O_RDONLY);
    if (fd < 0) {
        std::cerr << "Error opening file: " << matrix_file << std::endl;       // :35-38
        return 1;
    }
    const size_t map_off = (size_t)(b * row_bytes) & ~(size_t)4095, map_len = (size_t)(b * row_bytes) - map_off + bytes;
    void* m = ::mmap(nullptr, map_len, PROT_READ, MAP_PRIVATE, fd, (off_t)map_off);
    ::close(fd);
    if (m == MAP_FAILED) {
        std::cerr << "Error reading file: " << matrix_file << std::endl;
        return 1;
    }
    ::madvise(m, map_len, MADV_SEQUENTIAL);
    const char* base = (const char*)m + ((size_t)(b * row_bytes) - map_off);
    int rc = 0;
    try {
        out.ensure(ctx, bytes, false);
        const size_t chunk = (size_t)1 << 28;
        for (size_t at = 0; at < bytes && !rc; at += chunk)
            if (mvs_device_copy(ctx, out.as<char>() + at, MVS_MEM_DEVICE, base + at, MVS_MEM_HOST, std::min(chunk, bytes - at)) != MVS_OK)
                rc = gpu_fail("uploading vectors.bin");
        if (!rc && mvs_sketch_max_abs(ctx, out.p, elem_bytes, MVS_MEM_DEVICE, (e - b) * (int64_t)d, max_abs) != MVS_OK)
            rc = gpu_fail("largest |v| of the rank's rows");
    } catch (const mvs_step::StepError& err) {
        std::cerr << "pairwise_comp_optimized: " << err.what() << std::endl;
        rc = 2;
    }
    ::munmap(m, map_len);
    return rc;
}

// the round-1 scheme for this rank's shards, on the context the communicator lives on: limb planes gathered whole (or the
// whole file read), rows x all columns per shard through the streamed comparison -- what a dense result needs
static int legacy_rank(const StepJob& job, const StepRank& r, Gpu& x, bool collective) {
    int rc = 0;
    if (collective) {
        int64_t b = 0, e = 0;
        mvs_shard_rows(job.total_vectors, job.o->num_shards, r.shard_begin, &b, &e);
        rc = load_db_collective(x, job.matrix_file, job.elem_bytes, job.total_vectors, job.dimension, b, e, job.o->num_shards);
    } else {
        rc = load_db(x, job.matrix_file, job.elem_bytes, job.total_vectors, job.dimension);
    }
    for (int shard = r.shard_begin; shard < r.shard_end && !rc; ++shard) {
        const std::string folder = job.output_folder + "shard_" + std::to_string(shard) + "/";
        int64_t b = 0, e = 0, kept = 0;
        mvs_shard_rows(job.total_vectors, job.o->num_shards, shard, &b, &e);
        ShardStats st;
        rc = stream_shard(x, *job.norms_sq, job.keep_mode, b, e, folder, job.stage_timing, &kept, &st, [](const char*) {});
        std::lock_guard<std::mutex> lk(*job.out_mu);
        if (!rc) std::cout << "Jac space: " << st.jac_space << " ngh space: " << st.ngh_space << std::endl;   // :808
    }
    return rc;
}

static int run_step_rank(const StepJob& job, const StepRank& r) {
    const Options& o = *job.o;
    const int64_t N = job.total_vectors;
    const int64_t rps = (N + o.num_shards - 1) / std::max(1, o.num_shards);                       // :938
    const int64_t block_rows = (int64_t)(r.shard_end - r.shard_begin) * rps;                      // samples per rank block
    const auto own = mvs_step::rank_rows(N, block_rows, r.rank);
    auto t_lap = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t = std::chrono::steady_clock::now();
        if (job.stage_timing) {
            std::lock_guard<std::mutex> lk(*job.out_mu);
            std::cerr << "[stage] rank " << r.rank << ": " << what << " " << std::chrono::duration<double>(t - t_lap).count() << " s" << std::endl;
        }
        t_lap = t;
    };
    for (int shard = r.shard_begin; shard < r.shard_end; ++shard) {
        int64_t b = 0, e = 0;
        mvs_shard_rows(N, o.num_shards, shard, &b, &e);
        std::lock_guard<std::mutex> lk(*job.out_mu);
        std::cout << "Shard " << shard << " processing rows " << b << " to " << e << std::endl;   // :941
    }
    Gpu g, x;                       // the comparison's context; the exchange's (its own stream) with the communicator
    if (mvs_ctx_create(r.device, &g.ctx) != MVS_OK) return gpu_fail("creating context");
    if (job.stage_timing) mvs_ctx_set_timing(g.ctx, 1);
    if (r.world > 1) {
        if (mvs_ctx_create(r.device, &x.ctx) != MVS_OK) return gpu_fail("creating the exchange's context");
        const int rc = open_communicator(x, job.comm_kind, job.comm_base, r.rank, r.world);
        if (rc) return rc;
    }
    lap(r.world > 1 ? "contexts + communicator" : "context");
    // own rows; a failure before the first collective is carried into it (agree), not returned
    mvs_step::DevMem sketches;
    int64_t max_abs = 0;
    int rc = upload_rows(g.ctx, job.matrix_file, job.elem_bytes, job.dimension, own.first, own.second, sketches, &max_abs);
    int64_t max_abs_all = max_abs;
    if (r.world > 1) rc = agree(x, rc, &max_abs_all);
    if (rc) return rc;
    lap("own rows of vectors.bin on the device");
    bool too_dense = false;
    try {
        mvs_step::StepOptions so = mvs_step::StepOptions::from_env();
        so.timing = job.stage_timing;
        std::unique_ptr<mvs_step::Exchange> ex;
        if (r.world > 1) ex.reset(new mvs_step::Exchange(x.ctx, x.comm));
        mvs_step::ShardStep step(g.ctx, ex.get(), r.rank, r.world, so);
        step.run(N, block_rows, job.dimension, sketches.p, job.elem_bytes, own.second - own.first, max_abs, job.norms_sq->data(), nullptr,
                 job.keep_mode, mvs_limbs_for_max_abs(max_abs_all));
        too_dense = step.info.too_dense;
        lap("step: own rows re-coded, exchange, block plan, cells routed and sorted");
        if (job.stage_timing) {
            const mvs_step::StepInfo& si = step.info;
            std::lock_guard<std::mutex> lk(*job.out_mu);
            std::cerr << "[step] rank " << r.rank << "/" << r.world << ": prepare_own_rows_ms " << si.prepare_own_rows_ms << " plan_span_ms "
                      << si.plan_span_ms << " (filter_ms " << si.filter_ms << " recheck_ms " << si.recheck_ms << " flagged_tiles_ms "
                      << si.flagged_tiles_ms << "; " << si.filter_launches << " filter launch(es), " << si.filter_tiles << " tiles, "
                      << si.candidates << " candidates, " << si.flagged_tiles << " flagged tiles) cells_route_exchange_sort_ms "
                      << si.cells_route_exchange_sort_ms << " | " << step.n_cells() << " cells in this rank's rows, " << si.exchanged_cells
                      << " mirrored to other ranks, " << si.allgather_bytes_per_rank << " bytes gathered per rank ("
                      << (si.wire ? "coarse plane + low limbs" : "coarse plane + limb planes") << "), limbs " << si.limbs << ", " << si.blocks
                      << " block(s), " << si.attempts << " attempt(s)" << (si.note.empty() ? "" : "; ") << si.note << std::endl;
        }
        if (getenv("MVS_STEP_CHECKSUM")) {
            // bench.py's order-independent checksum of a shard (--config 3: `cells_checksum`): sum and sum of squares, mod 2^64, of
            // a 64-bit mix of (row, col, dot, q); summed over the ranks it is the same whatever the rank count
            std::vector<mvs_cell> host((size_t)step.n_cells());
            if (!host.empty() &&
                mvs_device_copy(g.ctx, host.data(), MVS_MEM_HOST, step.cells(), MVS_MEM_DEVICE, host.size() * sizeof(mvs_cell)) != MVS_OK)
                return gpu_fail("downloading the rank's cells");
            uint64_t s1 = 0, s2 = 0;
            for (const mvs_cell& c : host) {
                const uint64_t mix = ((uint64_t)(int64_t)c.row * 1000003ull + (uint64_t)(int64_t)c.col) * 2654435761ull +
                                     (uint64_t)(int64_t)c.dot * 40503ull + (uint64_t)(int64_t)c.q;
                s1 += mix;
                s2 += mix * mix;
            }
            char buf[160];
            snprintf(buf, sizeof buf, "[checksum] rank %d kept %lld sum %016llx sum2 %016llx", r.rank, (long long)step.n_cells(),
                     (unsigned long long)s1, (unsigned long long)s2);
            std::lock_guard<std::mutex> lk(*job.out_mu);
            std::cerr << buf << std::endl;
        }
        // the rank's sorted cells, shard by shard, through the device encoder into the shard folders
        for (int shard = r.shard_begin; shard < r.shard_end && !too_dense; ++shard) {
            const std::string folder = job.output_folder + "shard_" + std::to_string(shard) + "/";
            int64_t b = 0, e = 0, delivered = 0;
            mvs_shard_rows(N, o.num_shards, shard, &b, &e);
            ShardSink sink(folder);
            const int src = host_encoder_wanted()
                                ? mvs_cells_stream(g.ctx, step.cells(), step.n_cells(), b, e, &ShardSink::on_block, &sink, &delivered)
                                : mvs_cells_stream_encoded(g.ctx, step.cells(), step.n_cells(), b, e, &ShardSink::on_encoded, &sink, &delivered);
            if (src != MVS_OK) {
                if (!sink.error.empty()) std::cerr << "pairwise_comp_optimized: " << sink.error << std::endl;
                return gpu_fail("writing a shard's rows");
            }
            const ShardStats st = sink.writer.finish();                                           // :990
            std::lock_guard<std::mutex> lk(*job.out_mu);
            std::cout << "Jac space: " << st.jac_space << " ngh space: " << st.ngh_space << std::endl;   // :808
        }
        if (!too_dense) lap("shard files");
    } catch (const mvs_step::StepError& e) {
        std::lock_guard<std::mutex> lk(*job.out_mu);
        std::cerr << "pairwise_comp_optimized: rank " << r.rank << ": " << e.what() << std::endl;
        return 2;
    } catch (const std::exception& e) {
        std::lock_guard<std::mutex> lk(*job.out_mu);
        std::cerr << "pairwise_comp_optimized: rank " << r.rank << ": " << e.what() << std::endl;
        return 2;
    }
    if (too_dense) {
        // every rank read the same headers and is here: the streamed dense path, per shard (the step's buffers are gone)
        sketches.release();
        if (job.stage_timing) {
            std::lock_guard<std::mutex> lk(*job.out_mu);
            std::cerr << "[step] rank " << r.rank << ": the result is too dense for cell lists: streamed comparison per shard" << std::endl;
        }
        const bool per_shard_ranks = r.world > 1 && r.shard_end - r.shard_begin == 1 && r.world == o.num_shards;
        rc = legacy_rank(job, r, r.world > 1 ? x : g, per_shard_ranks);
        if (rc) return rc;
        lap("streamed comparison + shard files");
    }
    return 0;
}

// how many ranks a single process splits --num_shards over: the largest divisor of the shard count that the contexts allow
// (every rank owns the same number of consecutive shards: the blocks of the exchange have one size)
static int step_world(int num_shards, int contexts) {
    int w = 1;
    for (int k = 1; k <= std::min(num_shards, contexts); ++k)
        if (num_shards % k == 0) w = k;
    return w;
}

// --shard_idx -1 (an extension: the reference needs one process per shard): ALL shards from this one process, on all
// visible GPUs -- one device context and host thread per GPU, the whole vectors.bin resident on each (what each of the
// reference's shard processes reads too, :953-962), shard s computed by GPU s mod G.  MVS_DEVICE pins the work to one
// device; MVS_PAIRWISE_CONTEXTS=k asks for k contexts (cycling over the devices: k = 2 on a one-GPU box runs the
// multi-context path on device 0).
static int run_all_shards(const Options& o, const std::string& output_folder, const std::string& matrix_file, int elem_bytes,
                          int64_t total_vectors, int dimension, const std::vector<double>& norms_sq, int keep_mode, bool stage_timing) {
    std::vector<int> devices;
    {
        int ndev = 0;
        if (getenv("MVS_DEVICE") || mvs_device_count(&ndev) != MVS_OK || ndev <= 0) {
            devices.push_back(pick_device());
        } else {
            int want = std::min(ndev, std::max(1, o.num_shards));
            if (const char* e = getenv("MVS_PAIRWISE_CONTEXTS")) want = std::max(1, std::min(64, atoi(e)));
            for (int i = 0; i < want; ++i) devices.push_back(i % ndev);
        }
    }
    std::mutex out_mu;
    const char* step_env = getenv("MVS_STEP");
    if (!(step_env && step_env[0] == '0')) {
        // ONE step over all shards: W ranks (a host thread + contexts each), S / W consecutive shards per rank, every
        // unordered pair of row blocks compared once -- on one card: the whole frame once, where the round-1 scheme below
        // compares every shard's rows against all columns (S x the filter tiles of the symmetric schedule / 2)
        const int world = step_world(o.num_shards, (int)devices.size());
        bool distinct = true;
        for (int a = 0; a < world; ++a)
            for (int b = a + 1; b < world; ++b)
                if (devices[(size_t)a] == devices[(size_t)b]) distinct = false;
        StepJob job;
        job.o = &o;
        job.output_folder = output_folder;
        job.matrix_file = matrix_file;
        const char* coll = getenv("MVS_COLLECTIVE");
        job.comm_kind = (coll && *coll) ? coll : (distinct ? "rccl" : "files");      // (RCCL refuses two ranks on one device)
        if (job.comm_kind != "rccl" && job.comm_kind != "files") {
            std::cerr << "pairwise_comp_optimized: MVS_COLLECTIVE must be rccl or files" << std::endl;
            return 1;
        }
        job.comm_base = comm_base_path(output_folder, "p" + std::to_string((long)::getpid()));
        job.elem_bytes = elem_bytes;
        job.dimension = dimension;
        job.keep_mode = keep_mode;
        job.total_vectors = total_vectors;
        job.norms_sq = &norms_sq;
        job.stage_timing = stage_timing;
        job.out_mu = &out_mu;
        for (int shard = 0; shard < o.num_shards; ++shard) {
            const std::string folder = output_folder + "shard_" + std::to_string(shard) + "/";
            if (!fs::exists(folder)) fs::create_directories(folder);
        }
        std::vector<int> status((size_t)world, 0);
        auto rank_work = [&](int rk) {
            StepRank r;
            r.device = devices[(size_t)rk];
            r.rank = rk;
            r.world = world;
            r.shard_begin = rk * (o.num_shards / world);
            r.shard_end = (rk + 1) * (o.num_shards / world);
            status[(size_t)rk] = run_step_rank(job, r);
        };
        if (world == 1) {
            rank_work(0);
        } else {
            std::vector<std::thread> pool;
            for (int rk = 0; rk < world; ++rk) pool.emplace_back(rank_work, rk);
            for (auto& th : pool) th.join();
        }
        if (stage_timing)
            std::cerr << "[stage] " << o.num_shards << " shards in one step of " << world << " rank(s) (" << (world > 1 ? job.comm_kind : "no exchange")
                      << ")" << std::endl;
        for (int rc : status)
            if (rc) return rc;
        return 0;
    }
    const size_t n_ctx = devices.size();
    std::vector<int> status(n_ctx, 0);
    auto work = [&](size_t gi) {
        Gpu g;
        if (mvs_ctx_create(devices[gi], &g.ctx) != MVS_OK) {
            std::lock_guard<std::mutex> lk(out_mu);
            status[gi] = gpu_fail("creating context");
            return;
        }
        if (stage_timing) mvs_ctx_set_timing(g.ctx, 1);
        int rc = load_db(g, matrix_file, elem_bytes, total_vectors, dimension);
        for (int shard = (int)gi; shard < o.num_shards && !rc; shard += (int)n_ctx) {
            const std::string shard_folder = output_folder + "shard_" + std::to_string(shard) + "/";
            int64_t b = 0, e = 0, kept = 0;
            mvs_shard_rows(total_vectors, o.num_shards, shard, &b, &e);
            {
                std::lock_guard<std::mutex> lk(out_mu);
                std::cout << "Shard " << shard << " processing rows " << b << " to " << e << std::endl;   // :941
            }
            ShardStats st;
            rc = stream_shard(g, norms_sq, keep_mode, b, e, shard_folder, stage_timing, &kept, &st, [](const char*) {});
            std::lock_guard<std::mutex> lk(out_mu);
            if (!rc) std::cout << "Jac space: " << st.jac_space << " ngh space: " << st.ngh_space << std::endl;   // :808
        }
        status[gi] = rc;
    };
    if (n_ctx == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (size_t gi = 0; gi < n_ctx; ++gi) pool.emplace_back(work, gi);
        for (auto& th : pool) th.join();
    }
    if (stage_timing) std::cerr << "[stage] " << o.num_shards << " shards on " << n_ctx << " context(s)" << std::endl;
    for (int rc : status)
        if (rc) return rc;
    return 0;
}

int main(int argc, char* argv[]) {
    Options o;
    if (!parse(argc, argv, o) || o.show_help) {                                   // :846-850
        print_usage(argv[0]);
        return o.show_help ? 0 : 1;
    }
    std::string db_folder = o.db_folder, output_folder = o.output_folder;
    std::string dtype = "int32";
    const std::string dtype_file = db_folder + "dtype.txt";                       // :853 raw concatenation
    std::string norms_file = db_folder + "vector_norms.txt";
    if (!fs::exists(norms_file)) {                                                // :855-858
        std::cerr << "Error: Required file 'vector_norms.txt' not found in output folder: " << db_folder << std::endl;
        return 1;
    }
    {
        std::ifstream dtype_in(dtype_file);                                       // :859-865
        if (dtype_in) std::getline(dtype_in, dtype);
    }
    int dimension = 0;
    {
        std::ifstream dim_in(db_folder + "dimension.txt");                        // :866-873
        if (dim_in) dim_in >> dimension;
    }
    std::cout << "dtypeqs: " << dtype << std::endl;                               // :874
    const bool int16 = dtype == "int16";
    if (int16) std::cout << "dtyeom" << std::endl;                                // :877
    if (dimension <= 0) {
        std::cerr << "Error: could not read a positive dimension from " << db_folder << "dimension.txt" << std::endl;
        return 1;
    }
    if (!output_folder.empty() && output_folder.back() != '/' && output_folder.back() != '\\') output_folder += '/';
    if (int16) {
        // _16bits.cpp:334 reads the norms from the OUTPUT folder; fall back to the DB folder, where
        // sketch() actually writes them
        const std::string alt = output_folder + "vector_norms.txt";
        if (fs::exists(alt)) norms_file = alt;
    }
    const std::string matrix_file = db_folder + "vectors.bin";                    // :891
    DbInfo db;
    read_norms(norms_file, db);                                                   // :893-901
    const int elem_bytes = int16 ? 2 : 4;
    const int64_t bytes_per_vector = (int64_t)dimension * elem_bytes;
    if (!int16) {
        const int64_t max_bytes = (int64_t)(o.max_memory_gb * 1024 * 1024 * 1024);   // :904-908
        std::cout << "max bytes " << max_bytes << " " << o.max_memory_gb << std::endl;
        std::cout << "Using chunks of size " << mvs_chunk_size(o.max_memory_gb, dimension) << std::endl;
    }
    int64_t file_size = 0;
    {
        std::ifstream file(matrix_file, std::ios::ate | std::ios::binary);        // :911-914
        file_size = file ? (int64_t)file.tellg() : 0;
    }
    const int64_t total_vectors = file_size / bytes_per_vector;
    std::cout << "Total vectors: " << total_vectors << std::endl;                 // :916
    if ((int64_t)db.norms_sq.size() < total_vectors) {
        std::cerr << "Error: vector_norms.txt has " << db.norms_sq.size() << " entries for " << total_vectors
                  << " vectors" << std::endl;
        return 1;
    }
    auto start_time = std::chrono::high_resolution_clock::now();                  // :918

    if (o.shard_idx == -1) {
        if (int16 && legacy16_output()) {
            std::cerr << "pairwise_comp_optimized: --shard_idx -1 (all shards) writes the active shard format only" << std::endl;
            return 1;
        }
        db.norms_sq.resize((size_t)total_vectors);
        const int rc_all = run_all_shards(o, output_folder, matrix_file, elem_bytes, total_vectors, dimension, db.norms_sq,
                                          int16 ? MVS_KEEP_INT16 : MVS_KEEP_INT32, getenv("MVS_STAGE_TIMING") != nullptr);
        if (rc_all) return rc_all;
        auto end_time = std::chrono::high_resolution_clock::now();
        auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(end_time - start_time);
        std::cout << "Total computation time: " << duration.count() << " ms" << std::endl;
        return 0;
    }
    const std::string shard_folder = output_folder + "shard_" + std::to_string(o.shard_idx) + "/";   // :932-935
    if (!fs::exists(shard_folder)) fs::create_directories(shard_folder);
    int64_t begin_row = 0, end_row = 0;
    mvs_shard_rows(total_vectors, o.num_shards, o.shard_idx, &begin_row, &end_row);                   // :938-940
    std::cout << "Shard " << o.shard_idx << " processing rows " << begin_row << " to " << end_row << std::endl;

    // MVS_STAGE_TIMING=1: per-stage wall times on stderr (not part of the reference's output)
    const bool stage_timing = getenv("MVS_STAGE_TIMING") != nullptr;
    auto lap_t = std::chrono::steady_clock::now();
    auto lap = [&](const char* what) {
        const auto t = std::chrono::steady_clock::now();
        if (stage_timing)
            std::cerr << "[stage] " << what << " " << std::chrono::duration<double>(t - lap_t).count() << " s" << std::endl;
        lap_t = t;
    };
    Gpu g;
    int device = 0, ndev = 0;
    if (getenv("MVS_DEVICE")) device = pick_device();
    else if (mvs_device_count(&ndev) == MVS_OK && ndev > 0) device = o.shard_idx % ndev;
    if (mvs_ctx_create(device, &g.ctx) != MVS_OK) return gpu_fail("creating context");
    if (stage_timing) mvs_ctx_set_timing(g.ctx, 1);
    lap("context");
    // MVS_COLLECTIVE=rccl|files: the shard processes of the job exchange their row blocks instead of each reading
    // the whole file
    const char* coll = getenv("MVS_COLLECTIVE");
    int rc = 0;
    if (coll && *coll && o.num_shards > 1) {
        if (std::string(coll) != "rccl" && std::string(coll) != "files") {
            std::cerr << "pairwise_comp_optimized: MVS_COLLECTIVE must be rccl or files" << std::endl;
            return 1;
        }
        if (o.shard_idx < 0 || o.shard_idx >= o.num_shards) {
            std::cerr << "pairwise_comp_optimized: collective mode needs 0 <= shard_idx < num_shards" << std::endl;
            return 1;
        }
        const char* step_env = getenv("MVS_STEP");
        if (!(step_env && step_env[0] == '0') && !(int16 && legacy16_output())) {
            // the strong-scaled step: this process is rank shard_idx of num_shards (mvs_step.hpp)
            std::mutex out_mu;
            StepJob job;
            job.o = &o;
            job.output_folder = output_folder;
            job.matrix_file = matrix_file;
            job.comm_kind = coll;
            job.comm_base = comm_base_path(output_folder, "job");
            job.elem_bytes = elem_bytes;
            job.dimension = dimension;
            job.keep_mode = int16 ? MVS_KEEP_INT16 : MVS_KEEP_INT32;
            job.total_vectors = total_vectors;
            db.norms_sq.resize((size_t)total_vectors);
            job.norms_sq = &db.norms_sq;
            job.stage_timing = stage_timing;
            job.out_mu = &out_mu;
            StepRank r;
            r.device = device;
            r.rank = o.shard_idx;
            r.world = o.num_shards;
            r.shard_begin = o.shard_idx;
            r.shard_end = o.shard_idx + 1;
            mvs_ctx_destroy(g.ctx);                  // (the rank creates its own pair of contexts)
            g.ctx = nullptr;
            rc = run_step_rank(job, r);
            if (rc) return rc;
            auto end_time = std::chrono::high_resolution_clock::now();
            auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(end_time - start_time);
            std::cout << "Total computation time: " << duration.count() << " ms" << std::endl;     // :993-996
            return 0;
        }
        rc = open_communicator(g, coll, comm_base_path(output_folder, "job"), o.shard_idx, o.num_shards);
        if (rc) return rc;
        lap("communicator");
        rc = load_db_collective(g, matrix_file, elem_bytes, total_vectors, dimension, begin_row, end_row, o.num_shards);
        if (rc) return rc;
        lap("load own rows + all-gather");
    } else {
        rc = load_db(g, matrix_file, elem_bytes, total_vectors, dimension);
        if (rc) return rc;
        lap("load vectors.bin");
    }

    db.norms_sq.resize((size_t)total_vectors);
    const int keep_mode = int16 ? MVS_KEEP_INT16 : MVS_KEEP_INT32;
    if (int16 && legacy16_output()) {
        // The reference's own output for an int16 DB (_16bits.cpp:251-323, :426) stores round(dot / d) per cell: the one
        // consumer that needs the dot products, so this path keeps the cell list (mvs_pairwise_rows) instead of the
        // streamed (column, q) pieces.  --max_memory_gb bounds the staging buffer (16 bytes per cell, a quarter of it).
        double budget = o.max_memory_gb > 0 ? o.max_memory_gb : 1.0;
        Staging staging;
        staging.limit = (size_t)std::min<double>(budget * 1024.0 * 1024.0 * 1024.0 / 16.0 / 4.0, 256e6);
        staging.limit = std::max<size_t>(staging.limit, 1u << 20);
        std::vector<mvs_cell> all_results;
        rc = compare_rows(g, db.norms_sq, keep_mode, begin_row, end_row, total_vectors, staging, all_results);
        if (rc) return rc;
        lap("compare");
        auto end_time = std::chrono::high_resolution_clock::now();                  // _16bits.cpp:419-423
        auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(end_time - start_time);
        std::cout << "Total computation time: " << duration.count() << " ms" << std::endl;
        std::cout << "Total results: " << all_results.size() << std::endl;
        write_shard_legacy16(shard_folder, all_results.data(), all_results.size(), dimension);
        lap("write shard (legacy int16 format)");
        return 0;
    }
    int64_t n_kept = 0;
    ShardStats st;
    rc = stream_shard(g, db.norms_sq, keep_mode, begin_row, end_row, shard_folder, stage_timing, &n_kept, &st, lap);
    if (rc) return rc;
    if (int16) {                                                                  // _16bits.cpp:419-423
        auto end_time = std::chrono::high_resolution_clock::now();
        auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(end_time - start_time);
        std::cout << "Total computation time: " << duration.count() << " ms" << std::endl;
        std::cout << "Total results: " << n_kept << std::endl;
    }
    std::cout << "Jac space: " << st.jac_space << " ngh space: " << st.ngh_space << std::endl;        // :808
    if (!int16) {                                                                 // :993-996
        auto end_time = std::chrono::high_resolution_clock::now();
        auto duration = std::chrono::duration_cast<std::chrono::milliseconds>(end_time - start_time);
        std::cout << "Total computation time: " << duration.count() << " ms" << std::endl;
    }
    return 0;
}
