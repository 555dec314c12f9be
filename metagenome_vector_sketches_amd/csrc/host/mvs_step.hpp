// mvs_step.hpp -- the strong-scaled comparison step of one rank, in C++17 above the C ABI (include/mvs_hip.h).
//
// The reference shards the all-vs-all comparison by rows and lets every shard process compare its rows against ALL
// columns, each process re-reading the whole vectors.bin (src/pairwise_comp_optimized.cpp:937-982).  Here the G ranks of a
// job (one process per GPU, or one host thread per GPU of one process) run ONE step between them:
//
//   own rows  -> limb planes + the filter's inputs (coarse plane, row statistics) of the rank's OWN rows (k_recode_rows)
//   exchange  -> on the communicator's context (its own stream), in this order on every rank: row statistics (+ norms),
//                the coarse plane in row chunks (1 : 2), the LOW limbs (2 bytes per entry on the links; the plan rebuilds
//                the rows its re-check reads) -- or the limb planes themselves once some rank's |v| exceeds MVS_WIRE_MAX_ABS
//   compare   -> the rank's share of the symmetric schedule as one block plan (mvs_plan_*): the diagonal block at once, the
//                peers' blocks per arrived chunk; every unordered pair of row blocks is compared by exactly one rank
//   cells     -> routed on the device into this rank's rows and mirror images for the others; ONE fixed-size all-gather whose
//                64-byte headers carry every rank's status, largest |v| and overflow counters: every rank reads the same
//                headers and takes the same decision (redo with other limbs, regrow, leave together); collect; row-bucket
//                sort; one host synchronisation.
//
// This is metagenome_vector_sketches_amd/parallel.py: ShardedComparison in C++ (same partition, same exchange order, same
// header protocol; tests compare the two), driving libmvs_hip.so through nothing but the C ABI.  A rank may own SEVERAL
// consecutive shards of the reference's --num_shards (block_rows = shards per rank x ceil(N / S)): their rows are compared
// once, as one block, and the sorted cells are split by shard afterwards (mvs_cells_stream_encoded per shard folder).
#ifndef MVS_STEP_HPP
#define MVS_STEP_HPP

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../../include/mvs_hip.h"

namespace mvs_step {

struct StepError : std::runtime_error {
    int code;
    StepError(int c, const std::string& what) : std::runtime_error(what), code(c) {}
};
inline void check(int rc, const char* what) {
    if (rc != MVS_OK) throw StepError(rc, std::string(what) + ": " + mvs_last_error());
}

// -------------------------------------------------------------------------------------------------------------------
// partition (parallel.py: shard_rows / half_split / block_plan / chunk_bounds / clip_blocks; pure host arithmetic)
// -------------------------------------------------------------------------------------------------------------------
// rank r owns samples [r * block_rows, min((r + 1) * block_rows, n_total)): src/pairwise_comp_optimized.cpp:938-940 with
// block_rows = ceil(N / G), or a whole number of such shards per rank
inline std::pair<int64_t, int64_t> rank_rows(int64_t n_total, int64_t block_rows, int rank) {
    const int64_t b = std::min<int64_t>((int64_t)rank * block_rows, n_total);
    return {b, std::min<int64_t>(b + block_rows, n_total)};
}
inline int64_t pad256(int64_t rows) { return std::max<int64_t>(256, (rows + 255) / 256 * 256); }
// where the opposite block of an even world is cut: a multiple of 256 rows near the middle
inline int64_t half_split(int64_t block_pad) { return ((block_pad / 256 + 1) / 2) * 256; }

// Rectangles in STORAGE rows (rank p at [p * P, (p + 1) * P)) that `rank` compares: its diagonal block first, then the blocks
// whose kept cells are mirrored into other ranks' rows; over all ranks every unordered pair of rows is covered exactly once.
inline std::vector<mvs_plan_block> block_plan(int world, int rank, int64_t P, bool symmetric) {
    const int64_t rb = (int64_t)rank * P, re = rb + P;
    std::vector<mvs_plan_block> plan{{rb, re, rb, re}};
    if (!symmetric) {
        for (int p = 0; p < world; ++p)
            if (p != rank) plan.push_back({rb, re, (int64_t)p * P, (int64_t)(p + 1) * P});
        return plan;
    }
    for (int k = 1; k <= (world - 1) / 2; ++k) {
        const int64_t p = (rank + k) % world;
        plan.push_back({rb, re, p * P, (p + 1) * P});
    }
    if (world > 1 && world % 2 == 0) {
        const int64_t p = (rank + world / 2) % world, h = half_split(P);
        if (rank < p) {                    // the lower rank takes the first half of ITS rows against all of p's
            if (h > 0) plan.push_back({rb, rb + h, p * P, (p + 1) * P});
        } else if (h < P) {                // the higher rank takes all of its rows against the second half of p's
            plan.push_back({rb, re, p * P + h, (p + 1) * P});
        }
    }
    return plan;
}

// [(c0, c1)] cutting a block of P rows into at most `chunks` pieces on multiples of 256 rows; first in (0, 1): the share of
// the first piece, the others split the rest evenly (first <= 0: even pieces)
inline std::vector<std::pair<int64_t, int64_t>> chunk_bounds(int64_t P, int chunks, double first) {
    const int64_t tiles = P / 256;
    chunks = (int)std::max<int64_t>(1, std::min<int64_t>(chunks, tiles));
    std::vector<int64_t> cuts;
    if (first > 0.0 && chunks > 1) {
        // Python's round(): half to even
        const double x = (double)tiles * first;
        double r = std::floor(x + 0.5);
        if (x + 0.5 == r && std::fmod(r, 2.0) != 0.0) r -= 1.0;
        const int64_t t0 = std::max<int64_t>(1, std::min<int64_t>(tiles - (chunks - 1), (int64_t)r));
        const int64_t rest = tiles - t0;
        cuts.push_back(0);
        for (int k = 0; k < chunks; ++k) cuts.push_back((t0 + rest * k / (chunks - 1)) * 256);
    } else {
        for (int k = 0; k <= chunks; ++k) cuts.push_back((tiles * k / chunks) * 256);
    }
    std::vector<std::pair<int64_t, int64_t>> out;
    for (int k = 0; k < chunks; ++k)
        if (cuts[(size_t)k + 1] > cuts[(size_t)k]) out.emplace_back(cuts[(size_t)k], cuts[(size_t)k + 1]);
    return out;
}

// the parts of the rectangles whose columns lie at offsets [c0, c1) of their rank block
inline std::vector<mvs_plan_block> clip_blocks(const std::vector<mvs_plan_block>& blocks, size_t from, int64_t P, int64_t c0, int64_t c1) {
    std::vector<mvs_plan_block> out;
    for (size_t k = from; k < blocks.size(); ++k) {
        const mvs_plan_block& b = blocks[k];
        const int64_t base = (b.col_begin / P) * P;
        const int64_t lo = std::max(b.col_begin, base + c0), hi = std::min(b.col_end, base + c1);
        if (hi > lo) out.push_back({b.row_begin, b.row_end, lo, hi});
    }
    return out;
}

// -------------------------------------------------------------------------------------------------------------------
// device buffers (mvs_device_alloc / _free)
// -------------------------------------------------------------------------------------------------------------------
struct DevMem {
    mvs_ctx* ctx = nullptr;
    void* p = nullptr;
    size_t bytes = 0;
    DevMem() = default;
    DevMem(const DevMem&) = delete;
    DevMem& operator=(const DevMem&) = delete;
    ~DevMem() { release(); }
    void release() {
        if (p) (void)mvs_device_free(ctx, p);
        p = nullptr;
        bytes = 0;
    }
    // at least `want` bytes; a buffer that has to grow loses its contents.  Returns true when it was (re)allocated.
    bool ensure(mvs_ctx* c, size_t want, bool zero) {
        if (p && bytes >= want) return false;
        release();
        ctx = c;
        check(mvs_device_alloc(c, want, zero ? 1 : 0, &p), "device allocation");
        bytes = want;
        return true;
    }
    template <typename T>
    T* as() const { return static_cast<T*>(p); }
};

// -------------------------------------------------------------------------------------------------------------------
// the exchange: a communicator on a context of its own (its own stream), so that collectives run beside the comparison
// -------------------------------------------------------------------------------------------------------------------
// submit(compute, fn): fn(xctx, comm) issues collective calls that see everything `compute` has queued so far; wait(compute,
// handle) orders compute's stream behind them.  RCCL calls are asynchronous on xctx's stream and are issued inline; the
// file transport (ranks sharing one card) blocks inside every call, so its calls run on ONE worker thread: the collectives
// of a rank keep their order (invariant 5).
class Exchange {
 public:
    struct Handle {
        int slot = -1;
        unsigned long long ticket = 0;
    };
    Exchange(mvs_ctx* xctx, mvs_comm* comm) : xctx_(xctx), comm_(comm) {
        int r = 0, w = 1, k = 0;
        check(mvs_comm_info(comm, &r, &w, &k), "mvs_comm_info");
        rank_ = r;
        world_ = w;
        rccl_ = k != 0;
        for (Slot& s : slots_) {
            check(mvs_event_create(xctx, 0, &s.ready), "event");
            check(mvs_event_create(xctx, 0, &s.done), "event");
        }
        if (!rccl_) worker_ = std::thread([this] { run(); });
    }
    ~Exchange() {
        {
            std::lock_guard<std::mutex> lk(mu_);
            closing_ = true;
        }
        cv_.notify_all();
        if (worker_.joinable()) worker_.join();
        for (Slot& s : slots_) {
            (void)mvs_event_destroy(s.ready);
            (void)mvs_event_destroy(s.done);
        }
    }
    Exchange(const Exchange&) = delete;
    Exchange& operator=(const Exchange&) = delete;
    int rank() const { return rank_; }
    int world() const { return world_; }
    bool is_rccl() const { return rccl_; }
    mvs_ctx* ctx() const { return xctx_; }
    mvs_comm* comm() const { return comm_; }
    const char* kind() const { return rccl_ ? "libmvs_hip mvs_comm (RCCL)" : "libmvs_hip mvs_comm (file transport)"; }

    Handle submit(mvs_ctx* compute, std::function<void(mvs_ctx*, mvs_comm*)> fn) {
        const int idx = (int)(next_ % kSlots);
        Slot& s = slots_[(size_t)idx];
        {
            std::unique_lock<std::mutex> lk(mu_);
            cv_.wait(lk, [&] { return s.state != 1; });      // (64 exchanges in flight: never in practice)
            s.state = 1;
            s.error.clear();
            s.code = MVS_OK;
            s.ticket = ++next_;
        }
        check(mvs_event_record(compute, s.ready), "recording the exchange's start");
        Handle h{idx, s.ticket};
        if (rccl_) {
            execute(s, fn);
            std::lock_guard<std::mutex> lk(mu_);
            s.state = 2;
        } else {
            {
                std::lock_guard<std::mutex> lk(mu_);
                queue_.push_back({idx, std::move(fn)});
            }
            cv_.notify_all();
        }
        return h;
    }
    // the compute context's stream waits (on the device) for the exchange; the HOST waits only for a blocking transport
    void wait(mvs_ctx* compute, const Handle& h) {
        if (h.slot < 0) return;
        Slot& s = slots_[(size_t)h.slot];
        {
            std::unique_lock<std::mutex> lk(mu_);
            if (s.ticket != h.ticket) return;                // long done: the slot has been reused since
            cv_.wait(lk, [&] { return s.state == 2; });
            if (s.code != MVS_OK) throw StepError(s.code, s.error);
        }
        check(mvs_ctx_wait_event(compute, s.done), "ordering the comparison behind the exchange");
    }

 private:
    static constexpr int kSlots = 64;
    struct Slot {
        mvs_event* ready = nullptr;
        mvs_event* done = nullptr;
        int state = 0;              // 0 free, 1 submitted, 2 issued (RCCL) / completed (file transport)
        unsigned long long ticket = 0;
        int code = MVS_OK;
        std::string error;
    };
    struct Job {
        int slot;
        std::function<void(mvs_ctx*, mvs_comm*)> fn;
    };
    void execute(Slot& s, const std::function<void(mvs_ctx*, mvs_comm*)>& fn) {
        try {
            check(mvs_ctx_wait_event(xctx_, s.ready), "ordering the exchange behind the comparison");
            fn(xctx_, comm_);
            check(mvs_event_record(xctx_, s.done), "recording the exchange's end");
        } catch (const StepError& e) {
            s.code = e.code;
            s.error = e.what();
        } catch (const std::exception& e) {
            s.code = MVS_E_HIP;
            s.error = e.what();
        }
    }
    void run() {
        for (;;) {
            Job job;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [&] { return closing_ || !queue_.empty(); });
                if (queue_.empty()) return;
                job = std::move(queue_.front());
                queue_.pop_front();
            }
            Slot& s = slots_[(size_t)job.slot];
            execute(s, job.fn);
            {
                std::lock_guard<std::mutex> lk(mu_);
                s.state = 2;
            }
            cv_.notify_all();
        }
    }
    mvs_ctx* xctx_;
    mvs_comm* comm_;
    int rank_ = 0, world_ = 1;
    bool rccl_ = false;
    Slot slots_[kSlots];
    unsigned long long next_ = 0;
    std::mutex mu_;
    std::condition_variable cv_;
    std::deque<Job> queue_;
    bool closing_ = false;
    std::thread worker_;
};

// -------------------------------------------------------------------------------------------------------------------
// the step
// -------------------------------------------------------------------------------------------------------------------
struct StepOptions {
    bool symmetric = true;          // MVS_SHARDED_SYMMETRIC=0: the plain rows x all-columns schedule (no mirroring, no cell exchange)
    int gather_chunks = 2;          // MVS_GATHER_CHUNKS: pieces the peers' coarse rows arrive in
    double gather_first = 0.33;     // MVS_GATHER_FIRST: share of the first piece (transfer is about twice as fast as filtering per row)
    bool wire = true;               // MVS_WIRE_LOW_LIMB=0: limb planes on the wire instead of low limbs
    bool speculate = true;          // MVS_PLAN_SPECULATE=0: every plan waits for its own counts
    bool timing = false;            // stage spans from events on the step's streams (an event costs the stream ~6 us)
    bool exchange_in_place = false; // MEASUREMENT ONLY (mvs_step_bench, as tools/strong_model.py): a rank of `world` > 1 without a
                                    // communicator -- every byte the exchange would deliver has been put in place
                                    // (ShardStep::model_fill_peer), the collectives are skipped, the mirrored cells stay home
    int64_t dense_limit_cells = (int64_t)1 << 27;   // a rank whose blocks keep more than this (2 GiB of cells): the step reports
                                                    // too_dense on every rank and the caller takes the streamed dense path
    static StepOptions from_env() {
        StepOptions o;
        auto flag = [](const char* name, bool def) {
            const char* e = getenv(name);
            return e && *e ? std::string(e) != "0" : def;
        };
        o.symmetric = flag("MVS_SHARDED_SYMMETRIC", true);
        o.wire = flag("MVS_WIRE_LOW_LIMB", true);
        o.speculate = flag("MVS_PLAN_SPECULATE", true);
        if (const char* e = getenv("MVS_GATHER_CHUNKS")) o.gather_chunks = std::max(1, atoi(e));
        if (const char* e = getenv("MVS_GATHER_FIRST")) o.gather_first = atof(e);
        if (const char* e = getenv("MVS_STEP_DENSE_LIMIT")) o.dense_limit_cells = std::max<long long>(1, atoll(e));
        return o;
    }
};

struct StepInfo {
    int limbs = 2, blocks = 0, attempts = 0, own_regrown = 0, plan_respeculated = 0, redone = 0;
    bool wire = false, sorted_ahead = false, too_dense = false;
    int64_t exchanged_cells = 0, allgather_bytes_per_rank = 0, max_abs_all = 0;
    std::string note;
    // stage spans of the step (StepOptions::timing), as bench.py's `strong` record names them
    double prepare_own_rows_ms = 0, plan_span_ms = 0, cells_route_exchange_sort_ms = 0;
    // the plan's span cut where the exchange's arrivals matter (tools/strong_model.py walks these): own rows ready -> diagonal
    // block's filter queued and run; the peers' filter launches; low-limb rebuild + re-check + flagged tiles
    double diag_filter_ms = 0, peer_filters_ms = 0, finish_ms = 0;
    double filter_ms = 0, recheck_ms = 0, flagged_tiles_ms = 0;
    int64_t filter_launches = 0, filter_tiles = 0, candidates = 0, flagged_tiles = 0;
};

class ShardStep {
 public:
    ShardStep(mvs_ctx* ctx, Exchange* exchange, int rank, int world, StepOptions opt = StepOptions::from_env())
        : ctx_(ctx), ex_(exchange), rank_(rank), world_(world), opt_(opt) {
        if (world > 1 && !exchange && !opt.exchange_in_place) throw StepError(MVS_E_INVALID, "world > 1 needs a communicator");
        if (exchange && opt.exchange_in_place) throw StepError(MVS_E_INVALID, "exchange_in_place is for a step without a communicator");
        if (exchange && (exchange->rank() != rank || exchange->world() != world))
            throw StepError(MVS_E_INVALID, "the communicator's rank / world differ from the step's");
        wire_ok_ = opt.wire;
        set_timing(opt.timing);
    }
    // stage spans from events on the step's stream (an event between two kernels costs the stream ~6 us: off for timed steps)
    void set_timing(bool on) {
        opt_.timing = on;
        if (on && !ev_[0])
            for (mvs_event*& e : ev_) check(mvs_event_create(ctx_, 1, &e), "event");
    }
    ~ShardStep() {
        if (set_) (void)mvs_sketch_set_destroy(set_);
        for (mvs_event* e : ev_)
            if (e) (void)mvs_event_destroy(e);
    }
    ShardStep(const ShardStep&) = delete;
    ShardStep& operator=(const ShardStep&) = delete;

    // One step.  d_sketches: this rank's rows (DEVICE, n_local x d, elem_bytes 4 or 2; must stay valid until the call returns:
    // a redo with another limb code reads them again), max_abs_local their largest |v|; norms_sq_all: the squared parsed norms
    // of ALL n_total samples on the HOST (src/pairwise_comp_optimized.cpp:893-901: every process reads the same
    // vector_norms.txt) -- or NULL with d_norms_sq_local (DEVICE, n_local doubles): then the norms are exchanged with the
    // row statistics.  Afterwards cells() / n_cells() = this rank's rows x all columns, ordered by (row, col), on the device.
    void run(int64_t n_total, int64_t block_rows, int d, const void* d_sketches, int elem_bytes, int64_t n_local, int64_t max_abs_local,
             const double* norms_sq_all, const double* d_norms_sq_local, int keep_mode, int limbs_guess = 2) {
        info = StepInfo();
        const auto rows = rank_rows(n_total, block_rows, rank_);
        if (rows.second - rows.first != n_local) throw StepError(MVS_E_INVALID, "the rank's rows do not match its share of the samples");
        if ((int64_t)world_ * block_rows < n_total) throw StepError(MVS_E_INVALID, "the ranks' blocks do not cover the samples");
        src_ = {n_total, block_rows, d, d_sketches, elem_bytes, n_local, max_abs_local, norms_sq_all, d_norms_sq_local, keep_mode};
        int limbs = limbs_guess;
        for (int round = 0; round < 4; ++round) {
            begin(limbs);
            feed();
            const int verdict = finish(&limbs);
            if (verdict == 0) return;
            ++info.redone;
        }
        throw StepError(MVS_E_RANGE, "the step kept being redone");
    }
    const mvs_cell* cells() const { return sorted_.as<mvs_cell>(); }
    int64_t n_cells() const { return n_out_; }
    std::pair<int64_t, int64_t> rows() const { return rank_rows(src_.n_total, src_.block_rows, rank_); }
    // Measurement aid (tools/strong_model.py, csrc/host/mvs_step_bench.cpp): what the exchange WOULD have delivered for rank p's
    // block -- planes, filter inputs, low limbs -- built here from p's sketches, so that one rank's step can be timed on one
    // card with every byte of the exchange in place.  Call after a first run() (the buffers exist then); the data survives
    // later steps (a step rewrites its own block only).
    void model_fill_peer(int p, const void* d_sketches_p, int elem_bytes, int64_t n_rows_p) {
        if (!set_ || p == rank_) return;
        check(mvs_sketch_set_recode_rows(ctx_, set_, n_rows_p ? d_sketches_p : nullptr, elem_bytes, n_rows_p, (int64_t)p * P_, P_),
              "mvs_sketch_set_recode_rows (peer block)");
        if (lo_.p) check(mvs_sketch_set_wire_rows(ctx_, set_, lo_.as<int8_t>(), (int64_t)p * P_, P_), "mvs_sketch_set_wire_rows (peer block)");
    }
    int64_t block_pad() const { return P_; }
    const double* norms_sq_storage() const { return n2_.as<double>(); }     // DEVICE, indexed by storage row
    StepInfo info;

 private:
    struct Source {
        int64_t n_total = 0, block_rows = 0;
        int d = 0;
        const void* d_sketches = nullptr;
        int elem_bytes = 4;
        int64_t n_local = 0, max_abs = 0;
        const double* n2_all = nullptr;
        const double* d_n2_local = nullptr;
        int keep_mode = MVS_KEEP_INT32;
    };
    void mark(int k) {
        if (opt_.timing) check(mvs_event_record(ctx_, ev_[k]), "event");
    }

    // ---- buffers: reused while their geometry holds ----
    void begin(int limbs) {
        const Source& s = src_;
        limbs_ = limbs;
        P_ = pad256(s.block_rows);
        const int64_t n_st = P_ * world_;                    // storage rows; the rows behind a rank's samples stay zero
        int64_t n_alloc = 0;
        int d_pad = 0;
        size_t nbytes = 0;
        check(mvs_limb_geometry(n_st, s.d, limbs, &n_alloc, &d_pad, &nbytes), "mvs_limb_geometry");
        const bool same = set_ && key_limbs_ == limbs && n_alloc_ == n_alloc && d_pad_ == d_pad && key_n_st_ == n_st && key_d_ == s.d;
        if (!same) {
            if (set_) {
                check(mvs_sketch_set_destroy(set_), "mvs_sketch_set_destroy");
                set_ = nullptr;
            }
            planes_.release();
            coarse_.release();
            stats_.release();
            n2_.release();
            lo_.release();
            planes_.ensure(ctx_, nbytes, true);
            coarse_.ensure(ctx_, (size_t)n_alloc * (size_t)d_pad, true);     // fragment-major coarse plane (two-limb sets)
            stats_.ensure(ctx_, (size_t)n_alloc * 16, true);                 // 16 bytes of row statistics
            n2_.ensure(ctx_, (size_t)n_alloc * 8, true);
            if (limbs == 2 && world_ > 1 && wire_ok_) lo_.ensure(ctx_, (size_t)n_alloc * (size_t)d_pad, true);
            check(mvs_sketch_set_from_planes(ctx_, planes_.as<int8_t>(), n_st, n_alloc, s.d, d_pad, limbs, &set_), "mvs_sketch_set_from_planes");
            check(mvs_sketch_set_attach_derived(set_, coarse_.as<int8_t>(), stats_.p), "mvs_sketch_set_attach_derived");
            n_alloc_ = n_alloc;
            d_pad_ = d_pad;
            key_limbs_ = limbs;
            key_n_st_ = n_st;
            key_d_ = s.d;
            norms_in_place_ = false;
        }
        check(mvs_sketch_set_touch(set_), "mvs_sketch_set_touch");         // the buffers are about to be rewritten
        wire_ = wire_ok_ && limbs == 2 && world_ > 1 && lo_.p != nullptr;
        small_ = Exchange::Handle();
        planes_h_ = Exchange::Handle();
        coarse_h_.clear();
        rebuilt_ = false;
        mark(0);
    }

    // ---- own rows -> planes + filter inputs; the exchange of every rank's block starts ----
    void feed() {
        const Source& s = src_;
        const int64_t base = (int64_t)rank_ * P_;
        // norms: everybody has the file -> one upload into storage order (block p at p * P), once per geometry; else own norms
        // into this rank's block and the all-gather below brings the others'
        if (s.n2_all) {
            if (!norms_in_place_ || norms_src_ != s.n2_all) {     // (the same array again: its contents are taken as unchanged)
                std::vector<double> st((size_t)(P_ * world_), 0.0);
                for (int p = 0; p < world_; ++p) {
                    const auto r = rank_rows(s.n_total, s.block_rows, p);
                    if (r.second > r.first) std::memcpy(st.data() + (size_t)p * (size_t)P_, s.n2_all + r.first, (size_t)(r.second - r.first) * 8);
                }
                check(mvs_device_copy(ctx_, n2_.p, MVS_MEM_DEVICE, st.data(), MVS_MEM_HOST, st.size() * 8), "uploading the norms");
                norms_in_place_ = true;
                norms_src_ = s.n2_all;
            }
        } else if (s.n_local) {
            check(mvs_device_copy(ctx_, n2_.as<double>() + base, MVS_MEM_DEVICE, s.d_n2_local, MVS_MEM_DEVICE, (size_t)s.n_local * 8),
                  "placing the rank's norms");
        }
        check(mvs_sketch_set_recode_rows(ctx_, set_, s.n_local ? s.d_sketches : nullptr, s.elem_bytes, s.n_local, base, P_),
              "mvs_sketch_set_recode_rows");
        mark(1);
        if (world_ == 1) return;
        if (!ex_) {                         // exchange_in_place: only what this rank itself contributes to the wire buffer
            if (wire_) copy_low_limbs(base, P_);
            for (const auto& ch : chunk_bounds(P_, opt_.gather_chunks, opt_.gather_first)) coarse_h_.push_back({ch.first, ch.second, Exchange::Handle()});
            return;
        }
        const int64_t P = P_;
        const int d_pad = d_pad_, nl = limbs_ & 0xff;
        const bool gather_norms = s.n2_all == nullptr;
        // bytes per row: what every peer filter launch needs first
        small_ = ex_->submit(ctx_, [this, P, gather_norms](mvs_ctx* x, mvs_comm* cm) {
            check(mvs_allgather_bytes(x, cm, stats_.p, P * 16), "all-gather of the row statistics");
            if (gather_norms) check(mvs_allgather_f64(x, cm, n2_.as<double>(), P), "all-gather of the norms");
        });
        // the coarse plane in row chunks (units of 16 rows = 16 * d_pad contiguous bytes): the first piece is the small one --
        // it has to land while the diagonal block's filter runs
        for (const auto& ch : chunk_bounds(P, opt_.gather_chunks, opt_.gather_first)) {
            const int64_t a = ch.first, b = ch.second;
            coarse_h_.push_back({a, b, ex_->submit(ctx_, [this, P, a, b, d_pad](mvs_ctx* x, mvs_comm* cm) {
                                     check(mvs_allgather_rows(x, cm, coarse_.as<int8_t>(), P / 16, a / 16, (b - a) / 16, 1, 16 * d_pad),
                                           "all-gather of the coarse plane");
                                 })});
        }
        if (wire_) {
            // the LOW limbs of the rank's rows into the wire buffer (row r at r * d_pad): one strided device copy per row would be
            // thousands of calls -- the planes interleave limbs per row, so the library's row exchange moves "1 limb of 2"
            copy_low_limbs(base, P);
            planes_h_ = ex_->submit(ctx_, [this, P, d_pad](mvs_ctx* x, mvs_comm* cm) {
                check(mvs_allgather_rows(x, cm, lo_.as<int8_t>(), P, 0, P, 1, d_pad), "all-gather of the low limbs");
            });
        } else {
            planes_h_ = ex_->submit(ctx_, [this, P, nl, d_pad](mvs_ctx* x, mvs_comm* cm) {
                check(mvs_allgather_rows(x, cm, planes_.as<int8_t>(), P, 0, P, nl, d_pad), "all-gather of the limb planes");
            });
        }
    }
    // planes[(row * 2 + limb) * d_pad + k] -> lo[row * d_pad + k] for the rank's rows
    void copy_low_limbs(int64_t first, int64_t count) {
        check(mvs_sketch_set_wire_rows(ctx_, set_, lo_.as<int8_t>(), first, count), "mvs_sketch_set_wire_rows");
    }

    // ---- the rank's block plan: diagonal block at once, the other blocks per arrived chunk ----
    const uint64_t* compare(const std::vector<mvs_plan_block>& plan, bool mirror, bool first) {
        const int64_t P = P_;
        if (opt_.speculate) check(mvs_ctx_set_option(ctx_, "plan_speculate", 1), "plan_speculate");
        const int rc_begin = mvs_plan_begin(ctx_, set_, n2_.as<double>(), src_.keep_mode, (int64_t)rank_ * P, (int64_t)(rank_ + 1) * P,
                                            mirror ? MVS_PLAN_MIRROR_OUTSIDE : 0, raw_.as<mvs_cell>(), (int64_t)(raw_.bytes / sizeof(mvs_cell)));
        if (opt_.speculate) (void)mvs_ctx_set_option(ctx_, "plan_speculate", 0);
        check(rc_begin, "mvs_plan_begin");
        mark(2);
        check(mvs_plan_filter(ctx_, plan.data(), 1), "mvs_plan_filter (diagonal block)");     // nothing of it comes from another rank
        mark(5);
        const bool others = plan.size() > 1;
        if (first && ex_) ex_->wait(ctx_, small_);               // row statistics (+ norms) of every rank
        if (others) check(mvs_plan_rows_ready(ctx_, 0, (int64_t)world_ * P), "mvs_plan_rows_ready");
        int64_t counts[6] = {0, 0, 0, 0, 0, 0};
        bool exact_mode = false;
        if (first && others) {
            // (reading the plan's mode waits for nothing new: the previous plan's counts are on the host since its report)
            check(mvs_plan_stats(ctx_, nullptr, counts), "mvs_plan_stats");
            exact_mode = (counts[4] & 1) != 0;
        }
        if (first && others && exact_mode) {
            // no filter in this plan (filter switched off, another limb code): mvs_plan_filter runs the exact kernel on a block at
            // once, and that reads the other ranks' LIMB planes -- they have to be there (and rebuilt) before the call
            if (ex_) {
                for (auto& c : coarse_h_) ex_->wait(ctx_, c.h);
                ex_->wait(ctx_, planes_h_);
            }
            if (wire_ && !rebuilt_) rebuild_all();
            check(mvs_plan_filter(ctx_, plan.data() + 1, (int)plan.size() - 1), "mvs_plan_filter (peers' blocks, exact kernel)");
        } else if (first && !coarse_h_.empty()) {
            for (auto& c : coarse_h_) {
                if (ex_) ex_->wait(ctx_, c.h);
                const std::vector<mvs_plan_block> blocks = clip_blocks(plan, 1, P, c.a, c.b);
                if (!blocks.empty()) check(mvs_plan_filter(ctx_, blocks.data(), (int)blocks.size()), "mvs_plan_filter (peers' rows)");
            }
        } else if (others) {
            check(mvs_plan_filter(ctx_, plan.data() + 1, (int)plan.size() - 1), "mvs_plan_filter (peers' blocks)");
        }
        mark(6);
        // the limb planes: the re-check reads them.  Every exchange of the step is joined here even if this rank's plan needs
        // nothing from anybody (rank 1 of 2): the next step rewrites the buffers the collectives read
        if (first && ex_) ex_->wait(ctx_, planes_h_);
        // the plan rebuilds the rows its re-check and flagged tiles read -- the columns of its candidates -- and no others
        if (wire_ && !rebuilt_ && others) check(mvs_plan_wire(ctx_, lo_.as<int8_t>()), "mvs_plan_wire");
        const uint64_t* d_cnt = nullptr;
        check(mvs_plan_finish(ctx_, &d_cnt), "mvs_plan_finish");
        mark(3);
        return d_cnt;
    }
    void rebuild_all() {
        for (int p = 0; p < world_; ++p)
            if (p != rank_)
                check(mvs_sketch_set_planes_from_wire(ctx_, set_, lo_.as<int8_t>(), (int64_t)p * P_, P_), "mvs_sketch_set_planes_from_wire");
        rebuilt_ = true;
    }

    // ---- cells: route, exchange, collect, sort, report.  Returns 0 (done) or 1 (redo the step; *limbs says how) ----
    int finish(int* limbs_next) {
        const Source& s = src_;
        const auto own = rank_rows(s.n_total, s.block_rows, rank_);
        const int64_t rb = own.first, re = own.second;
        const bool mirror = opt_.symmetric && world_ > 1;
        const std::vector<mvs_plan_block> plan = block_plan(world_, rank_, P_, opt_.symmetric);
        info.limbs = limbs_;
        info.wire = wire_;
        info.blocks = (int)plan.size();
        info.allgather_bytes_per_rank = world_ > 1 ? P_ * (int64_t)(wire_ ? 2 : (limbs_ & 0xff) + 1) * d_pad_ + P_ * 24 : 0;
        // two capacities (parallel.py: finish): the shard itself, and what the rank's blocks may produce per direction.  A raw
        // list that overflowed is a COLLECTIVE matter (every rank reads it in the headers), a shard that overflowed a LOCAL one.
        int64_t cap_own = std::max<int64_t>(std::max<int64_t>(cap_own_, (int64_t)1 << 16), 64 * std::max<int64_t>(re - rb, 1));
        int64_t cap_raw = std::max(cap_raw_, cap_own);
        int status = MVS_OK;
        std::string err;
        const uint64_t* d_cnt = nullptr;
        bool need_compute = true, local_only = false;
        std::vector<int64_t> rep((size_t)(2 + 5 * world_), 0);
        for (int attempt = 0; attempt < 8; ++attempt) {
            ++info.attempts;
            const int64_t want_raw = (mirror ? 2 : 1) * cap_raw + 1024;
            // the raw buffer is only ever replaced in front of a comparison: between a plan and the routing it IS the result
            if (need_compute) raw_.ensure(ctx_, (size_t)want_raw * sizeof(mvs_cell), false);
            const bool own_new = own_.ensure(ctx_, (size_t)cap_own * sizeof(mvs_cell), false);
            if (sorted_.ensure(ctx_, (size_t)cap_own * sizeof(mvs_cell), false) || own_new) sort_ahead_ = false;
            state_.ensure(ctx_, (size_t)(16 + 4 * (re - rb + 2) + 8), true);
            if (!zero_.p) zero_.ensure(ctx_, 64, true);
            if (need_compute && status == MVS_OK) {
                try {
                    d_cnt = compare(plan, mirror, attempt == 0);
                } catch (const StepError& e) {        // the others learn about it from the header of the cell exchange
                    status = e.code ? e.code : MVS_E_HIP;
                    err = e.what();
                    d_cnt = nullptr;
                }
            }
            // every rank's send buffer = a 64-byte header {foreign cells, status, max |v|, raw cells, raw capacity} + room for
            // cap_f mirror images; ONE all-gather of them tells every rank how every other rank fared
            const int64_t cap_f = mirror ? cap_f_ : 0;
            const int64_t stride = MVS_CELLS_HEADER_BYTES + 16 * cap_f;
            if (xbuf_.bytes < (size_t)(world_ * stride)) {
                if (local_only) throw StepError(MVS_E_INVALID, "internal: the exchange buffer of a local repeat was resized");
                xbuf_.ensure(ctx_, (size_t)(world_ * stride), true);
            }
            char* send = xbuf_.as<char>() + (size_t)rank_ * (size_t)stride;
            check(mvs_cells_route(ctx_, raw_.as<mvs_cell>(), d_cnt ? d_cnt : zero_.as<uint64_t>(), (int64_t)(raw_.bytes / sizeof(mvs_cell)), P_,
                                  s.block_rows, s.n_total, rb, re, own_.as<mvs_cell>(), (int64_t)(own_.bytes / sizeof(mvs_cell)),
                                  state_.as<uint64_t>(), send, cap_f, status, s.max_abs),
                  "mvs_cells_route");
            if (world_ > 1) {
                if (!local_only && ex_) {          // (a local repeat rewrote this rank's own block with the same cells)
                    const Exchange::Handle h = ex_->submit(ctx_, [this, stride](mvs_ctx* x, mvs_comm* cm) {
                        check(mvs_allgather_bytes(x, cm, xbuf_.p, stride), "all-gather of the mirrored cells");
                    });
                    ex_->wait(ctx_, h);
                }
                if (mirror)
                    check(mvs_cells_collect(ctx_, xbuf_.p, world_, rank_, cap_f, rb, re, own_.as<mvs_cell>(), (int64_t)(own_.bytes / sizeof(mvs_cell)),
                                            state_.as<uint64_t>()),
                          "mvs_cells_collect");
            }
            // the sort, queued in front of the step's host synchronisation when the previous step says the row buckets will do
            const bool ahead = sort_ahead_ && d_cnt != nullptr;
            if (ahead) {
                check(mvs_cells_sort_rows_ahead(ctx_, own_.as<mvs_cell>(), (int64_t)(own_.bytes / sizeof(mvs_cell)), rb, re, state_.as<uint64_t>(),
                                                sorted_.as<mvs_cell>(), (int64_t)(sorted_.bytes / sizeof(mvs_cell))),
                      "mvs_cells_sort_rows_ahead");
                mark(4);
            }
            check(mvs_cells_report(ctx_, xbuf_.p, world_, cap_f, re - rb, state_.as<uint64_t>(), rep.data()), "mvs_cells_report");   // the host sync
            local_only = false;
            const int64_t n_out = rep[0], max_row = rep[(size_t)(1 + 5 * world_)];
            auto head = [&](int r, int k) { return rep[(size_t)(1 + 5 * r + k)]; };
            int64_t worst = 0, max_abs_all = 0, max_foreign = 0, max_raw = 0;
            bool any_stale = false, any_raw_over = false;
            for (int r = 0; r < world_; ++r) {
                worst = std::max(worst, head(r, 1));
                max_abs_all = std::max(max_abs_all, head(r, 2));
                max_foreign = std::max(max_foreign, head(r, 0));
                const uint64_t raw_n = (uint64_t)head(r, 3);
                if (raw_n >= MVS_PLAN_STALE) any_stale = true;
                else {
                    max_raw = std::max<int64_t>(max_raw, (int64_t)raw_n);
                    if ((int64_t)raw_n > head(r, 4)) any_raw_over = true;
                }
            }
            info.max_abs_all = max_abs_all;
            if (worst) throw StepError((int)worst, !err.empty() ? err : std::string("another rank failed in its block comparisons"));
            if (wire_ && max_abs_all > MVS_WIRE_MAX_ABS) {
                // some rank's values are beyond what the low limb pins: every rank sees the same headers, the step is redone
                // with the limb planes themselves on the wire, from now on
                wire_ok_ = false;
                *limbs_next = limbs_;
                info.note = "|v| beyond " + std::to_string(MVS_WIRE_MAX_ABS) + ": step redone with the limb planes on the wire";
                return 1;
            }
            const int need = mvs_limbs_for_max_abs(max_abs_all);
            const bool plain = need <= 4 && limbs_ <= 4;
            if (plain ? need > limbs_ : need != limbs_) {
                *limbs_next = need;
                info.note = "limb guess " + std::to_string(limbs_) + " did not hold: step redone with " + std::to_string(need);
                return 1;
            }
            if (any_stale) {
                // a rank's plan ran its second half on the previous step's counts and they did not hold: that rank compares
                // again -- the library will not speculate this time --, the others exchange again with it
                need_compute = (uint64_t)head(rank_, 3) >= MVS_PLAN_STALE;
                ++info.plan_respeculated;
                continue;
            }
            if (max_raw > opt_.dense_limit_cells) {      // read from the headers: every rank takes this exit together
                info.too_dense = true;
                info.note = "a rank's blocks keep " + std::to_string(max_raw) + " cells: too dense for cell lists";
                n_out_ = 0;
                return 0;
            }
            // ---- decisions every rank takes alike: they read nothing but the exchanged headers ----
            bool redo = false;
            need_compute = head(rank_, 3) > head(rank_, 4);          // this rank's raw list overflowed: its blocks again, with room
            if (any_raw_over) {                                      // ... and the others exchange again with it
                redo = true;
                if (need_compute) cap_raw = std::max(cap_raw, head(rank_, 3) / (mirror ? 2 : 1) + 1);
            }
            if (mirror && max_foreign > cap_f) {
                cap_f_ = max_foreign * 5 / 4 + 1024;
                redo = true;
            }
            if (redo) {
                if (n_out > cap_own) cap_own = n_out + n_out / 4;     // (a lower bound while lists overflow: the repeat will tell)
                continue;
            }
            // ---- this rank's own business: its shard did not fit.  No other rank knows, and none needs to ----
            if (n_out > (int64_t)(own_.bytes / sizeof(mvs_cell))) {
                cap_own = n_out + n_out / 4;
                need_compute = false;
                local_only = true;
                ++info.own_regrown;
                continue;
            }
            // ---- done: order the shard ----
            const bool sorted_ahead = ahead && max_row <= 64;
            if (n_out && !sorted_ahead) {
                if (max_row <= 64)
                    check(mvs_cells_sort_rows(ctx_, own_.as<mvs_cell>(), n_out, rb, re, state_.as<uint64_t>(), sorted_.as<mvs_cell>()), "mvs_cells_sort_rows");
                else
                    check(mvs_cells_sort(ctx_, own_.as<mvs_cell>(), n_out, sorted_.as<mvs_cell>()), "mvs_cells_sort");
            }
            if (!sorted_ahead) mark(4);
            sort_ahead_ = max_row <= 64;
            info.sorted_ahead = sorted_ahead;
            info.exchanged_cells = mirror ? head(rank_, 0) : 0;
            n_out_ = n_out;
            cap_own_ = cap_own;
            cap_raw_ = cap_raw;
            if (opt_.timing) read_spans();
            return 0;
        }
        throw StepError(MVS_E_CAPACITY, "the step's buffers kept overflowing");
    }
    void read_spans() {
        float ms = 0.0f;
        if (mvs_event_elapsed_ms(ev_[0], ev_[1], &ms) == MVS_OK) info.prepare_own_rows_ms = ms;
        if (mvs_event_elapsed_ms(ev_[2], ev_[3], &ms) == MVS_OK) info.plan_span_ms = ms;
        if (mvs_event_elapsed_ms(ev_[3], ev_[4], &ms) == MVS_OK) info.cells_route_exchange_sort_ms = ms;
        if (mvs_event_elapsed_ms(ev_[1], ev_[5], &ms) == MVS_OK) info.diag_filter_ms = ms;
        if (mvs_event_elapsed_ms(ev_[5], ev_[6], &ms) == MVS_OK) info.peer_filters_ms = ms;
        if (mvs_event_elapsed_ms(ev_[6], ev_[3], &ms) == MVS_OK) info.finish_ms = ms;
        double pms[4] = {0, 0, 0, 0};
        int64_t counts[6] = {0, 0, 0, 0, 0, 0};
        if (mvs_plan_stats(ctx_, pms, counts) == MVS_OK) {
            info.filter_ms = pms[0];
            info.recheck_ms = pms[1];
            info.flagged_tiles_ms = pms[2];
            info.candidates = counts[0];
            info.flagged_tiles = counts[1];
            info.filter_tiles = counts[2];
            info.filter_launches = counts[3];
        }
    }

    mvs_ctx* ctx_;
    Exchange* ex_;
    int rank_, world_;
    StepOptions opt_;
    Source src_;
    const double* norms_src_ = nullptr;      // the host norms that are in place on the device
    bool wire_ok_ = true, wire_ = false, rebuilt_ = false, norms_in_place_ = false, sort_ahead_ = false;
    int limbs_ = 2, d_pad_ = 0, key_limbs_ = 0, key_d_ = 0;
    int64_t P_ = 0, n_alloc_ = 0, key_n_st_ = 0, n_out_ = 0;
    int64_t cap_f_ = 1 << 14, cap_own_ = 0, cap_raw_ = 0;
    mvs_sketch_set* set_ = nullptr;
    DevMem planes_, coarse_, stats_, n2_, lo_, raw_, own_, sorted_, state_, xbuf_, zero_;
    Exchange::Handle small_, planes_h_;
    struct CoarseChunk {
        int64_t a, b;
        Exchange::Handle h;
    };
    std::vector<CoarseChunk> coarse_h_;
    mvs_event* ev_[7] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
};

}  // namespace mvs_step

#endif
