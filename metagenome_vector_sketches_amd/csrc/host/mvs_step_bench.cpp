// mvs_step_bench -- one rank's strong-scaled step (mvs_step.hpp: ShardStep, the code pairwise_comp_optimized runs) timed on ONE
// card at the per-rank problem size of a G-way split, with every byte the exchange would deliver already in place: what
// tools/strong_model.py measures for the Python step, for the C++ host, and for EVERY rank of the split (the slowest one
// bounds a step).
//
//   mvs_step_bench --db <folder>/ --ranks G [--rank r | --rank -1 (all, default)] [--steps K] [--warmup W] [--probe P]
//
// Prints one JSON object: per rank the wall of a step without instrumentation (mean / median / min over K steps, host clock
// around run() + a stream synchronisation) and the stage spans of P further steps with events on the stream, under the names of
// bench.py's `strong` record.  The exchange itself is modelled elsewhere (tools/strong_model.py --from-cpp).
#include <chrono>
#include <cstdio>
#include <numeric>

#include "mvs_host.hpp"
#include "mvs_step.hpp"

using namespace mvs_host;

static int fail(const char* what) {
    fprintf(stderr, "mvs_step_bench: %s: %s\n", what, mvs_last_error());
    return 2;
}

int main(int argc, char** argv) {
    std::string db;
    int G = 1, only = -1, steps = 20, warmup = 5, probe = 3, report_spin = -1;
    for (int i = 1; i + 1 < argc; i += 2) {
        const std::string a = argv[i];
        if (a == "--db") db = argv[i + 1];
        else if (a == "--ranks") G = atoi(argv[i + 1]);
        else if (a == "--rank") only = atoi(argv[i + 1]);
        else if (a == "--steps") steps = atoi(argv[i + 1]);
        else if (a == "--warmup") warmup = atoi(argv[i + 1]);
        else if (a == "--probe") probe = atoi(argv[i + 1]);
        else if (a == "--report-spin") report_spin = atoi(argv[i + 1]);      // microseconds mvs_cells_report polls before it blocks
        else {
            fprintf(stderr, "unknown flag %s\n", a.c_str());
            return 1;
        }
    }
    if (db.empty() || G < 1) {
        fprintf(stderr, "usage: %s --db <folder>/ --ranks G [--rank r] [--steps K] [--warmup W] [--probe P]\n", argv[0]);
        return 1;
    }
    int d = 0;
    {
        std::ifstream in(db + "dimension.txt");
        if (in) in >> d;
    }
    DbInfo info;
    if (d <= 0 || !read_norms(db + "vector_norms.txt", info)) {
        fprintf(stderr, "mvs_step_bench: %s is not a DB folder\n", db.c_str());
        return 1;
    }
    const std::string matrix = db + "vectors.bin";
    const int64_t N = (int64_t)fs::file_size(matrix) / ((int64_t)d * 4);
    info.norms_sq.resize((size_t)N);
    mvs_ctx* ctx = nullptr;
    if (mvs_ctx_create(pick_device(), &ctx) != MVS_OK) return fail("creating context");
    if (report_spin >= 0 && mvs_ctx_set_option(ctx, "report_spin", report_spin) != MVS_OK) return fail("report_spin");
    // the whole DB on the device: a rank's own rows and -- once -- what the exchange would have delivered of the others'
    mvs_step::DevMem all;
    {
        const int fd = ::open(matrix.c_str(), O_RDONLY);
        const size_t bytes = (size_t)N * (size_t)d * 4;
        void* m = fd >= 0 ? ::mmap(nullptr, bytes, PROT_READ, MAP_PRIVATE, fd, 0) : MAP_FAILED;
        if (fd >= 0) ::close(fd);
        if (m == MAP_FAILED) {
            fprintf(stderr, "mvs_step_bench: cannot map %s\n", matrix.c_str());
            return 1;
        }
        all.ensure(ctx, bytes, false);
        for (size_t at = 0; at < bytes; at += (size_t)1 << 28)
            if (mvs_device_copy(ctx, all.as<char>() + at, MVS_MEM_DEVICE, (const char*)m + at, MVS_MEM_HOST, std::min((size_t)1 << 28, bytes - at)) != MVS_OK)
                return fail("upload");
        ::munmap(m, bytes);
    }
    const int64_t block_rows = (N + G - 1) / G;
    printf("{\"n\": %lld, \"d\": %d, \"ranks\": %d, \"steps\": %d, \"host\": \"csrc/host/mvs_step.hpp (C++)\", \"per_rank\": [", (long long)N, d, G, steps);
    bool first_out = true;
    double worst_median = 0.0;
    int worst_rank = 0;
    try {
        for (int r = 0; r < G; ++r) {
            if (only >= 0 && r != only) continue;
            mvs_step::StepOptions so = mvs_step::StepOptions::from_env();
            so.exchange_in_place = G > 1;
            so.timing = false;
            mvs_step::ShardStep step(ctx, nullptr, r, G, so);
            const auto own = mvs_step::rank_rows(N, block_rows, r);
            const void* mine = all.as<char>() + (size_t)own.first * (size_t)d * 4;
            int64_t max_abs = 0;
            if (own.second > own.first &&
                mvs_sketch_max_abs(ctx, mine, 4, MVS_MEM_DEVICE, (own.second - own.first) * (int64_t)d, &max_abs) != MVS_OK)
                return fail("max |v|");
            auto run_once = [&]() {
                step.run(N, block_rows, d, mine, 4, own.second - own.first, max_abs, info.norms_sq.data(), nullptr, MVS_KEEP_INT32, 2);
            };
            mvs_ctx_set_timing(ctx, 0);
            run_once();                                     // buffers exist now
            for (int p = 0; p < G; ++p) {
                const auto rows = mvs_step::rank_rows(N, block_rows, p);
                step.model_fill_peer(p, all.as<char>() + (size_t)rows.first * (size_t)d * 4, 4, rows.second - rows.first);
            }
            for (int k = 0; k < warmup; ++k) run_once();
            mvs_ctx_synchronize(ctx);
            std::vector<double> walls;
            for (int k = 0; k < steps; ++k) {
                const auto t0 = std::chrono::steady_clock::now();
                run_once();
                mvs_ctx_synchronize(ctx);
                walls.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            const mvs_step::StepInfo plain = step.info;
            // stage spans: `probe` more steps with events on the stream (the last one's spans)
            step.set_timing(true);
            mvs_ctx_set_timing(ctx, 1);
            std::vector<double> inst;
            for (int k = 0; k < probe; ++k) {
                const auto t0 = std::chrono::steady_clock::now();
                run_once();
                mvs_ctx_synchronize(ctx);
                inst.push_back(std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
            }
            const mvs_step::StepInfo si = step.info;
            std::vector<double> sorted = walls;
            std::sort(sorted.begin(), sorted.end());
            const double mean = std::accumulate(walls.begin(), walls.end(), 0.0) / (double)walls.size();
            const double median = sorted[sorted.size() / 2];
            if (median > worst_median) {
                worst_median = median;
                worst_rank = r;
            }
            printf("%s{\"rank\": %d, \"rows\": [%lld, %lld], \"rows_per_rank_padded\": %lld, \"wall_ms\": %.4f, \"wall_ms_median\": %.4f, \"wall_ms_min\": %.4f, "
                   "\"wall_instrumented_ms\": %.4f, \"prepare_own_rows_ms\": %.4f, \"plan_span_ms\": %.4f, \"cells_route_exchange_sort_ms\": %.4f, "
                   "\"diag_filter_ms\": %.4f, \"peer_filters_ms\": %.4f, \"finish_ms\": %.4f, \"filter_ms\": %.4f, \"recheck_ms\": %.4f, \"flagged_tiles_ms\": %.4f, \"filter_launches\": %lld, \"filter_tiles\": %lld, "
                   "\"candidates\": %lld, \"flagged_tiles\": %lld, \"own_cells\": %lld, \"foreign_cells\": %lld, \"plan_blocks\": %d, "
                   "\"attempts_per_step\": %d, \"sorted_ahead\": %s, \"wire\": %s, \"walls_ms\": [",
                   first_out ? "" : ", ", r, (long long)own.first, (long long)own.second, (long long)step.block_pad(), mean, median, sorted.front(),
                   inst.empty() ? 0.0 : inst.back(), si.prepare_own_rows_ms, si.plan_span_ms, si.cells_route_exchange_sort_ms, si.diag_filter_ms, si.peer_filters_ms, si.finish_ms, si.filter_ms, si.recheck_ms,
                   si.flagged_tiles_ms, (long long)si.filter_launches, (long long)si.filter_tiles, (long long)si.candidates, (long long)si.flagged_tiles,
                   (long long)step.n_cells(), (long long)si.exchanged_cells, si.blocks, plain.attempts, plain.sorted_ahead ? "true" : "false",
                   si.wire ? "true" : "false");
            for (size_t k = 0; k < walls.size(); ++k) printf("%s%.4f", k ? ", " : "", walls[k]);
            printf("]}");
            first_out = false;
            fflush(stdout);
        }
    } catch (const std::exception& e) {
        fprintf(stderr, "mvs_step_bench: %s\n", e.what());
        return 2;
    }
    printf("], \"slowest_rank\": %d, \"slowest_rank_wall_ms_median\": %.4f}\n", worst_rank, worst_median);
    all.release();
    mvs_ctx_destroy(ctx);
    return 0;
}
