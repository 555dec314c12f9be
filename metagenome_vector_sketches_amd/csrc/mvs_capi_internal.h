// mvs_capi_internal.h -- what the translation units of the C ABI (mvs_capi*.hip) share: the context and sketch-set objects, the
// buffer / read-back helpers, and the stages of the comparison that the streamed output and the block plans drive themselves.
// Not installed.
#ifndef MVS_CAPI_INTERNAL_H
#define MVS_CAPI_INTERNAL_H

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <deque>
#include <functional>
#include <memory>
#include <dlfcn.h>
#include <mutex>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <chrono>
#include <thread>
#include <vector>

#include <array>

#include "mvs_encode.h"
#include "mvs_internal.h"

// a balanced tile order of a filter launch (mvs::plan_tile_order), cached by launch geometry: the comparisons of a job repeat
// the same few shapes, so a list is built and uploaded once
struct mvs_tile_order {
    std::vector<long long> key;
    unsigned* d = nullptr;            // NULL: the static map serves this shape
    unsigned per = 0;
};

struct mvs_ctx {
    int device = 0;
    std::vector<mvs_tile_order> tile_orders;
    mvs::Options opt;   // tuning switches: environment defaults read once at creation, then mvs_ctx_set_option
    hipStream_t own_stream = nullptr;
    hipStream_t stream = nullptr;
    // timing of the dominant kernels (optional)
    bool timing = false;
    // event pairs: 0 projection kernel, 1 whole comparison (filter + re-check, or the exact kernel),
    // 2 the filter kernel alone, 3 the re-check kernel alone
    hipEvent_t ev[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    bool ev_valid[5] = {false, false, false, false, false};   // [4]: exact kernel on the flagged tiles (ev[6]..ev[3])
    // reusable device scratch
    void* scratch = nullptr;
    size_t scratch_bytes = 0;
    unsigned long long* d_counter = nullptr;   // 8-byte slot for counters / max
    // grow-only device buffers of mvs_pairwise_rows (no hipMalloc/hipFree on the hot path)
    void* pw_thr = nullptr;   size_t pw_thr_bytes = 0;
    void* pw_tmp = nullptr;   size_t pw_tmp_bytes = 0;
    void* pw_sort = nullptr;  size_t pw_sort_bytes = 0;
    void* pw_out = nullptr;   size_t pw_out_bytes = 0;
    void* stage = nullptr;    size_t stage_bytes = 0;   // host sketches on their way to the limb planes
    // two-stage comparison: coarse plane + row statistics of the set `coarse_id` (generation `coarse_gen`),
    // per-call filter constants, candidate list
    void* pw_coarse = nullptr;  size_t pw_coarse_bytes = 0;
    void* pw_coarse_fm = nullptr;  size_t pw_coarse_fm_bytes = 0;   // fragment-major copy (streaming search filters), built on demand
    void* st_tlist = nullptr;  size_t st_tlist_bytes = 0;    // dense row passes: active tiles per tile row of the block, their counts,
    void* st_tlist_n = nullptr;  size_t st_tlist_n_bytes = 0;  // and every row's first / last kept column
    void* st_ends = nullptr;  size_t st_ends_bytes = 0;
    void* pw_need = nullptr;  size_t pw_need_bytes = 0;             // block plans: rows whose limb planes are to be rebuilt (mvs_plan_wire)
    void* pw_planes_fm = nullptr;  size_t pw_planes_fm_bytes = 0;   // fragment-major copy of the limb planes of set planes_fm_id
    unsigned long long planes_fm_id = 0, planes_fm_gen = 0;         // (generation planes_fm_gen), for the ping-pong exact kernel
    bool coarse_fm_valid = false;           // ... of the cached plane
    void* pw_rows = nullptr;    size_t pw_rows_bytes = 0;
    void* pw_fmeta = nullptr;   size_t pw_fmeta_bytes = 0;
    void* pw_cand = nullptr;    size_t pw_cand_bytes = 0;
    // streamed output (mvs_pairwise_stream): packed kept cells raw / sorted, their CSR form, the download side
    void* st_raw = nullptr;     size_t st_raw_bytes = 0;
    void* st_sorted = nullptr;  size_t st_sorted_bytes = 0;
    void* st_col[2] = {nullptr, nullptr};     size_t st_col_bytes[2] = {0, 0};   // CSR arrays of two row blocks: one is
    void* st_q[2] = {nullptr, nullptr};       size_t st_q_bytes[2] = {0, 0};     // downloaded while the next is built
    void* st_rowptr = nullptr;  size_t st_rowptr_bytes = 0;
    void* st_counts = nullptr;  size_t st_counts_bytes = 0;
    void* st_dense = nullptr;   size_t st_dense_bytes = 0;  // dense results: one byte per cell (mvs_internal.h)
    void* rb_pinned = nullptr;  size_t rb_bytes = 0;        // pinned landing zone of the small per-block read-backs (row index,
                                                            // record offsets, counters): a copy into pageable memory makes the
                                                            // runtime stage and block per copy -- 5 of them per row block
    size_t st_dense_zero = 0;   // the first st_dense_zero bytes of st_dense are zero once the work queued on `stream` is through:
                                // the tile-granular dense flow needs a cleared matrix, clears it again behind its last block --
                                // while the link still drains -- and so finds it clean the next time
    // rows encoded on the device (mvs_pairwise_stream_encoded): per-row sizes / offsets / directory, the records themselves
    void* en_size = nullptr;    size_t en_size_bytes = 0;
    void* en_off = nullptr;     size_t en_off_bytes = 0;
    void* en_jac = nullptr;     size_t en_jac_bytes = 0;
    void* en_first = nullptr;   size_t en_first_bytes = 0;
    void* en_par = nullptr;     size_t en_par_bytes = 0;
    void* st_enc[2] = {nullptr, nullptr};     size_t st_enc_bytes[2] = {0, 0};
    hipStream_t dl_stream = nullptr;
    void* dl_pinned[2] = {nullptr, nullptr};      // pinned host buffers, each allocated (and grown) when first needed:
    size_t dl_bytes[2] = {0, 0};                  // pinning costs ~0.3 ms per MiB, a one-piece result needs only one
    hipEvent_t dl_done[2] = {nullptr, nullptr};   // download into pinned buffer i has completed
    hipEvent_t dl_block[2] = {nullptr, nullptr};  // the downloads out of CSR array set i have completed
    hipEvent_t dl_ready[2] = {nullptr, nullptr};  // the arrays of the row block in set i are final on the compute stream
    void* dl_hsa = nullptr;                       // option stream_copy = 1: agents + completion signals of the DMA copies
    void (*dl_hsa_free)(void*) = nullptr;         // (mvs_capi_stream.hip owns the type)
    hipStream_t post_stream = nullptr;            // dense row blocks -> CSR / encoded rows beside the next block's comparison
    hipEvent_t cmp_done = nullptr;                // the comparison launch of the block about to be post-processed is through
    // what the last mvs_pairwise_stream did (mvs_ctx_stream_stats)
    double st_kernel_ms = 0.0;                    // comparison kernels, summed over the row blocks (timing enabled)
    long long st_bytes = 0, st_blocks = 0, st_pieces = 0, st_two_stage = 0;
    // tile-granular two-stage comparison: tile flags, flagged tiles per tile row, their row-major list (entry 0 = total),
    // the candidate list without the pairs of flagged tiles
    void* pw_tflag = nullptr;   size_t pw_tflag_bytes = 0;
    void* pw_trow = nullptr;    size_t pw_trow_bytes = 0;
    void* pw_tlist = nullptr;   size_t pw_tlist_bytes = 0;
    void* pw_cand2 = nullptr;   size_t pw_cand2_bytes = 0;
    void* pw_ttouch = nullptr;  size_t pw_ttouch_bytes = 0;   // dense byte matrix: tiles the re-check's cells were scattered into,
    void* pw_tnew = nullptr;    size_t pw_tnew_bytes = 0;     // and the list of those touched for the first time (to be cleared)
    long long last_flagged_tiles = 0, last_filter_tiles = 0;   // of the last two-stage comparison (mvs_ctx_pairwise_stats)
    void* pw_chdr = nullptr;    size_t pw_chdr_bytes = 0;   // candidate regions of the ping-pong filter: counts, entries
    void* pw_cent = nullptr;    size_t pw_cent_bytes = 0;
    unsigned long long coarse_id = 0, coarse_gen = 0;
    int coarse_mode = -1;                   // radix rule (option coarse_radix) the cached plane was built with
    unsigned long long few_rows_id = 0, few_rows_gen = 0;   // the set whose last comparison was a block of < 1024 rows done by the
                                                            // exact kernel because no coarse plane existed (pairwise_launch)
    unsigned long long filter_off_id = 0;   // (set, coefficient) for which the filter passed too many pairs
    double filter_off_coeff = 0.0;
    unsigned long long last_candidates = 0; // candidate pairs of the last two-stage comparison (0: exact kernel)
    unsigned long long h_start = 0;      // host copy of the starting cell count of an appending call
    // host hash lists on their way to the device: two pinned staging buffers + a copy stream, so that the host-side
    // copy into pinned memory, the DMA and the projection kernel of consecutive pieces overlap
    void* up_pinned[2] = {nullptr, nullptr};
    size_t up_bytes = 0;
    hipStream_t up_stream = nullptr;
    hipEvent_t up_done[2] = {nullptr, nullptr};   // DMA out of staging buffer i has completed
    // pinned host staging for small metadata uploads (projection unit lists)
    void* pinned = nullptr;
    size_t pinned_bytes = 0;
    hipEvent_t pinned_ev = nullptr;
    bool pinned_busy = false;
    // block plans (mvs_plan_*): state between begin / filter / finish, scratch of mvs_sketch_set_prepare_rows, events
    int plan_overlap = 0;                 // block plans: 1 = filter launches alternate between the stream and a side stream
    int report_spin = 0;                  // mvs_cells_report: microseconds to poll the stream before blocking on it (0: block at once)
                                          // (measured at the per-rank size of an 8-way split: filters 1.191 -> 1.164 ms, within
                                          // the box-to-box spread; off by default -- one more queue beside RCCL's for 2 %)
    struct PlanState* plan = nullptr;
    void* plan_tmp = nullptr;   size_t plan_tmp_bytes = 0;
    const void* rows_max_done = nullptr;   // state block whose widest row mvs_cells_sort_rows_ahead has already computed
};

struct mvs_sketch_set {
    mvs_ctx* ctx = nullptr;
    const int8_t* planes = nullptr;
    int8_t* owned = nullptr;
    int64_t n = 0, n_alloc = 0;
    int d = 0, d_pad = 0, limbs = 0;
    unsigned long long id = 0, gen = 0;   // identity of the plane contents (cache key of derived data)
    // mvs_sketch_set_attach_derived: the filter's inputs in caller buffers (block plans), NULL otherwise
    int8_t* ext_coarse_fm = nullptr;
    mvs::CoarseRow* ext_rows = nullptr;
    // rows rewritten (mvs_sketch_set_fill) since the context's derived data of this set was built: refreshed row by row on
    // the next comparison instead of rebuilding everything (a search appends a handful of query rows to a resident database)
    int64_t dirty_lo = 0, dirty_hi = 0;
};

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return mvs_capi::fail(MVS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace mvs_capi {

extern std::atomic<unsigned long long> g_set_ids;

// sets the calling thread's error message (mvs_last_error) and returns `code`
int fail(int code, const char* fmt, ...);
inline bool mem_ok(int m) { return m == MVS_MEM_HOST || m == MVS_MEM_DEVICE; }

// rocprofv3 --marker-trace range around an entry point (option `markers`)
struct Range {
    bool on = false;
    Range(const mvs_ctx* c, const char* name);
    ~Range();
};

// RAII device buffer used for staging host inputs / outputs
struct DevBuf {
    void* p = nullptr;
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

int ensure_buf(mvs_ctx* c, void** p, size_t* have, size_t bytes);
struct ReadBack {
    void* dst;
    const void* src;
    size_t bytes;
};
int ensure_read_back(mvs_ctx* c, size_t total);
int read_back(mvs_ctx* c, hipStream_t st, std::initializer_list<ReadBack> items);
int ensure_scratch(mvs_ctx* c, size_t bytes);
int acquire_pinned(mvs_ctx* c, size_t bytes);
void parallel_copy(void* dst, const void* src, size_t bytes);
constexpr size_t kUploadPiece = 32u << 20;    // bytes per staging buffer (pinning memory costs ~0.3 ms per MiB: keep them small)
int ensure_upload_pipeline(mvs_ctx* c);
int check_kernel(const char* what);
void plan_state_free(mvs_ctx* c);      // mvs_capi_plan.hip

// ---- mvs_capi_sketch.hip ----
void note_rows_rewritten(mvs_sketch_set* s, int64_t lo, int64_t hi);

// ---- mvs_capi_compare.hip ----
int refresh_derived(mvs_ctx* c, const mvs_sketch_set* cs);
// the balanced order of the launch `segs` describes with the kernel arguments `a` (option plan_order; *d == NULL: use the static map)
int tile_order_for(mvs_ctx* c, const mvs::PairwiseArgs& a, const mvs::PlanSegs& segs, const unsigned** d, unsigned* per);
struct PackedOut {
    void** buf;
    size_t* bytes;
    int64_t row0;              // rows are stored relative to this one
    int shift;                 // row field starts at this bit (16 bits of q, then the column)
    bool two_stage_only;       // do not fall back to the exact kernel: return kNeedExact and let the caller plan row blocks
};
constexpr int kNeedExact = 100;   // internal status of pairwise_launch (never leaves the library)

// Streamed output where the result is dense: the exact kernel writes one byte per cell (q or 0) into a row-major matrix
// instead of appending to a list; [sym_begin, sym_end) is the square the symmetric schedule works in -- larger than the
// launch's own rows when a caller walks a shard block by block and lets the mirror images land in later blocks' rows.
struct DenseOut {
    uint8_t* matrix;
    int64_t row0, ld;
    int64_t sym_begin, sym_end;
    unsigned int* flag;
};

// ---- the two-stage comparison, stage by stage ----
// What the filter stage leaves for the stages after it.  The stages are separate functions because the streamed output
// decides BETWEEN them how the kept cells leave the device (a list when they are few, the dense byte matrix when whole
// regions of the result are dense) and, for the matrix, launches the flagged tiles row block by row block.
struct TwoStage {
    mvs::PairwiseArgs a{};            // the filter launch's arguments: candidate list (pruned), tile grid, symmetric square
    bool tiles = false;               // the filter could flag tiles (tile-granular comparison)
    int n_tr = 0, n_tc = 0;           // its grid of 256 x 256 tiles
    unsigned long long n_cand = 0;    // listed candidates (an upper bound once the list has been pruned)
    int n_flagged = 0;                // flagged tiles
    std::vector<int> row_first;       // n_tr + 1 entries: where each tile row starts in the row-major list of flagged tiles
    const int* d_list = nullptr;      // that list on the device
    mvs::Options opt;                 // the options the filter stage ran with: the later stages use the same
    unsigned int* ext_flags = nullptr;   // in: tile flags live here (this launch's tile rows of a larger grid) instead of in
                                         // the context's own array -- the streamed pipeline keeps one array for the whole matrix
};

void fill_args(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, int keep_mode, int64_t rb, int64_t re, int64_t cb,
               int64_t ce, bool symmetric, bool mirror_all, double keep_coeff, mvs::PairwiseArgs& a, const mvs::Options* o = nullptr);
// the running cell count starts at `start` (appending calls); kKeepCount: it stays what the device counter holds
constexpr unsigned long long kKeepCount = ~0ULL;
int set_cell_count(mvs_ctx* c, unsigned long long start);
int two_stage_filter(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, double keep_coeff, int64_t capacity_hint,
                     bool hold_all, unsigned long long start, mvs::PairwiseArgs& a, TwoStage& ts, const mvs::Options* o = nullptr);
int two_stage_recheck(mvs_ctx* c, TwoStage& ts);
int two_stage_tiles(mvs_ctx* c, TwoStage& ts, int first, int count, bool timed);
bool two_stage_applies(mvs_ctx* c, const mvs_sketch_set* s, int64_t rb, int64_t re, int64_t cb, int64_t ce, double keep_coeff,
                       bool symmetric = true, const mvs::Options* o = nullptr);
int sort_on_device(mvs_ctx* c, mvs_cell* in, int64_t n, mvs_cell* out);
int pairwise_launch(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, int keep_mode, int64_t rb, int64_t re,
                    int64_t cb, int64_t ce, bool symmetric, bool mirror_all, mvs_cell* raw, int64_t capacity,
                    unsigned long long start, unsigned long long* count, double keep_coeff = 0.05,
                    const PackedOut* po = nullptr, const DenseOut* dn = nullptr, const mvs::Options* o = nullptr);
}  // namespace mvs_capi

#endif
