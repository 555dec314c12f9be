// mvs_capi.hip -- the C ABI of libmvs_hip.so (include/mvs_hip.h): contexts, buffer staging and
// kernel orchestration.  No compute happens on the host here and there is no CPU fallback.
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

struct mvs_ctx {
    int device = 0;
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

static void plan_state_free(mvs_ctx* c);   // defined with PlanState (block plans, near the end of this file)

namespace mvs {
int capi_fail(int code, const char* fmt, ...);
hipStream_t capi_stream(mvs_ctx* c) { return c->stream; }
int capi_device(mvs_ctx* c) { return c->device; }
const Options& capi_options(mvs_ctx* c) { return c->opt; }
}  // namespace mvs

namespace {

thread_local std::string g_err;
std::atomic<unsigned long long> g_set_ids{0};

int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

}  // namespace

int mvs::capi_fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

namespace {

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(MVS_E_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

bool mem_ok(int m) { return m == MVS_MEM_HOST || m == MVS_MEM_DEVICE; }

// rocprofv3 --marker-trace ranges around the ABI's main entry points (option `markers`, off by default).  The roctx
// library (librocprofiler-sdk-roctx / libroctx64) is bound at run time: without it the ranges are no-ops.
struct Roctx {
    int (*push)(const char*) = nullptr;
    int (*pop)() = nullptr;
    Roctx() {
        for (const char* name : {"librocprofiler-sdk-roctx.so.1", "librocprofiler-sdk-roctx.so", "libroctx64.so.4",
                                 "libroctx64.so", "/opt/rocm/lib/librocprofiler-sdk-roctx.so"}) {
            if (void* h = dlopen(name, RTLD_NOW | RTLD_LOCAL)) {
                push = reinterpret_cast<int (*)(const char*)>(dlsym(h, "roctxRangePushA"));
                pop = reinterpret_cast<int (*)()>(dlsym(h, "roctxRangePop"));
                if (push && pop) return;
                push = nullptr;
                pop = nullptr;
            }
        }
    }
};
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

int ensure_buf(mvs_ctx* c, void** p, size_t* have, size_t bytes) {
    if (*have >= bytes) return MVS_OK;
    if (*p) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(*p));
        *p = nullptr;
        *have = 0;
    }
    const size_t want = bytes + std::min<size_t>(bytes / 4, (size_t)256 << 20) + 4096;   // head room: avoid regrowing on small changes
    if (hipMalloc(p, want) != hipSuccess) return fail(MVS_E_NOMEM, "hipMalloc of %zu bytes failed", want);
    *have = want;
    return MVS_OK;
}

// Several small device -> host read-backs with ONE synchronisation: the copies land in a pinned buffer of the context (a copy
// into pageable memory is staged by the runtime and blocks the host once per copy), the stream is synchronised once, then
// the values are copied out.  Only the thread that drives the context's comparison calls this.
struct ReadBack {
    void* dst;
    const void* src;
    size_t bytes;
};
int ensure_read_back(mvs_ctx* c, size_t total) {
    if (c->rb_pinned && c->rb_bytes >= total) return MVS_OK;
    if (c->rb_pinned) HIP_TRY(hipHostFree(c->rb_pinned));
    c->rb_pinned = nullptr;
    c->rb_bytes = 0;
    const size_t want = std::max<size_t>(total * 2, (size_t)1 << 20);
    HIP_TRY(hipHostMalloc(&c->rb_pinned, want, hipHostMallocDefault));
    c->rb_bytes = want;
    return MVS_OK;
}
int read_back(mvs_ctx* c, hipStream_t st, std::initializer_list<ReadBack> items) {
    size_t total = 0;
    for (const ReadBack& it : items) total += (it.bytes + 63) & ~(size_t)63;
    const int rc_rb = ensure_read_back(c, total);
    if (rc_rb) return rc_rb;
    size_t at = 0;
    for (const ReadBack& it : items) {
        if (it.bytes) HIP_TRY(hipMemcpyAsync((char*)c->rb_pinned + at, it.src, it.bytes, hipMemcpyDeviceToHost, st));
        at += (it.bytes + 63) & ~(size_t)63;
    }
    HIP_TRY(hipStreamSynchronize(st));
    at = 0;
    for (const ReadBack& it : items) {
        if (it.bytes) memcpy(it.dst, (const char*)c->rb_pinned + at, it.bytes);
        at += (it.bytes + 63) & ~(size_t)63;
    }
    return MVS_OK;
}

int ensure_scratch(mvs_ctx* c, size_t bytes) {
    if (c->scratch_bytes >= bytes) return MVS_OK;
    if (c->scratch) {
        HIP_TRY(hipStreamSynchronize(c->stream));
        HIP_TRY(hipFree(c->scratch));
        c->scratch = nullptr;
        c->scratch_bytes = 0;
    }
    HIP_TRY(hipMalloc(&c->scratch, bytes));
    c->scratch_bytes = bytes;
    return MVS_OK;
}

// pinned host buffer whose previous upload has completed
int acquire_pinned(mvs_ctx* c, size_t bytes) {
    if (c->pinned_busy) {
        HIP_TRY(hipEventSynchronize(c->pinned_ev));
        c->pinned_busy = false;
    }
    if (c->pinned_bytes >= bytes) return MVS_OK;
    if (c->pinned) {
        HIP_TRY(hipHostFree(c->pinned));
        c->pinned = nullptr;
        c->pinned_bytes = 0;
    }
    const size_t want = bytes < (1u << 20) ? (1u << 20) : bytes;
    HIP_TRY(hipHostMalloc(&c->pinned, want, hipHostMallocDefault));
    c->pinned_bytes = want;
    return MVS_OK;
}

// ---- options: one table drives the environment defaults, the setter and the getter ----
struct OptionSpec {
    const char* name;      // mvs_ctx_set_option name; the environment variable is MVS_<NAME in upper case>
    int mvs::Options::*ifield;
    double mvs::Options::*dfield;
    long long lo, hi;
};
const OptionSpec kOptions[] = {
    {"pairwise_filter", &mvs::Options::pairwise_filter, nullptr, 0, 2},
    {"filter_variant", &mvs::Options::filter_variant, nullptr, -1, 99},
    {"exact_variant", &mvs::Options::exact_variant, nullptr, 0, 3},
    {"pairwise_variant", &mvs::Options::pairwise_variant, nullptr, 0, 9},
    {"pairwise_symmetric", &mvs::Options::pairwise_symmetric, nullptr, 0, 1},
    {"pairwise_debug", &mvs::Options::pairwise_debug, nullptr, 0, 15},
    {"sort", &mvs::Options::sort, nullptr, 0, 2},
    {"enable_k3", &mvs::Options::enable_k3, nullptr, 0, 1},
    {"markers", &mvs::Options::markers, nullptr, 0, 1},
    {"project_variant", &mvs::Options::project_variant, nullptr, 0, 14},
    {"comm_timeout_s", &mvs::Options::comm_timeout_s, nullptr, 1, 86400},
    {"pairwise_map", &mvs::Options::pairwise_map, nullptr, 0, 2},
    {"coarse_radix", &mvs::Options::coarse_radix, nullptr, 0, 1},
    {"cand_regions", &mvs::Options::cand_regions, nullptr, 0, 1},
    {"recheck_mode", &mvs::Options::recheck_mode, nullptr, 0, 3},
    {"recheck_blocks", &mvs::Options::recheck_blocks, nullptr, 1, 64},
    {"stream_dense", &mvs::Options::stream_dense, nullptr, 0, 3},
    {"encode_stage_words", &mvs::Options::encode_stage_words, nullptr, 1, 64},
    {"stream_block_rows", &mvs::Options::stream_block_rows, nullptr, 0, 1 << 30},
    {"tile_dense_thr", &mvs::Options::tile_dense_thr, nullptr, 0, 8192},
    {"stream_list_cells", &mvs::Options::stream_list_cells, nullptr, 0, 1 << 30},
    {"stream_pipeline", &mvs::Options::stream_pipeline, nullptr, 0, 1},
    {"stream_trace", &mvs::Options::stream_trace, nullptr, 0, 1},
    {"search_stream", &mvs::Options::search_stream, nullptr, 0, 1},
    {"search_depth", &mvs::Options::search_depth, nullptr, 3, 6},
    {"fragment_major", &mvs::Options::fragment_major, nullptr, 0, 1},
    {"pairwise_bdirect", &mvs::Options::pairwise_bdirect, nullptr, 0, 1},
    {"plan_strip_wgs", &mvs::Options::plan_strip_wgs, nullptr, 256, 1 << 22},
    {"recode_rows_wg", &mvs::Options::recode_rows_wg, nullptr, 8, 16},
    {"plan_speculate", &mvs::Options::plan_speculate, nullptr, 0, 1},
    {"pairwise_block_cells", nullptr, &mvs::Options::pairwise_block_cells, 1, (1LL << 62)},
};

int apply_option(mvs::Options& o, const OptionSpec& sp, long long v) {
#ifndef MVS_ABLATIONS
    if (sp.ifield == &mvs::Options::pairwise_debug && v != 0)
        return fail(MVS_E_INVALID, "pairwise_debug needs a library built with -DMVS_ABLATIONS");
    if (sp.ifield == &mvs::Options::filter_variant && v >= 11 && v <= 33)
        return fail(MVS_E_INVALID, "filter_variant %lld is a k-loop ablation: needs -DMVS_ABLATIONS", v);
#endif
    if (v < sp.lo || v > sp.hi) return fail(MVS_E_INVALID, "option %s: %lld outside [%lld, %lld]", sp.name, v, sp.lo, sp.hi);
    if (sp.ifield) o.*(sp.ifield) = (int)v;
    else o.*(sp.dfield) = (double)v;
    return MVS_OK;
}

// MVS_<NAME> environment variables give the initial values (MVS_SORT also takes "merge" / "radix");
// values the setter would reject are ignored
void options_from_env(mvs::Options& o) {
    for (const OptionSpec& sp : kOptions) {
        std::string env = "MVS_";
        for (const char* p = sp.name; *p; ++p) env += (char)toupper((unsigned char)*p);
        const char* e = getenv(env.c_str());
        if (!e || !*e) continue;
        long long v;
        if (sp.ifield == &mvs::Options::sort && (e[0] == 'm' || e[0] == 'r')) v = e[0] == 'r' ? 2 : 1;
        else v = sp.dfield ? (long long)atof(e) : atoll(e);
        const std::string keep = g_err;
        if (apply_option(o, sp, v) != MVS_OK) g_err = keep;
    }
}

// Copy `bytes` from pageable host memory into pinned memory on several threads: one thread moves ~10 GB/s, the PCIe
// link 50+ GB/s, so a single memcpy would be what bounds the upload.
void parallel_copy(void* dst, const void* src, size_t bytes) {
    const size_t kMin = 8u << 20;
    unsigned nt = std::thread::hardware_concurrency();
    nt = nt == 0 ? 4 : (nt > 16 ? 16 : nt);
    if (bytes < 2 * kMin) nt = 1;
    if (nt <= 1) {
        std::memcpy(dst, src, bytes);
        return;
    }
    std::vector<std::thread> pool;
    const size_t per = (bytes / nt + 4095) & ~(size_t)4095;
    for (unsigned t = 0; t < nt; ++t) {
        const size_t b = (size_t)t * per;
        if (b >= bytes) break;
        const size_t n = std::min(per, bytes - b);
        pool.emplace_back([=]() { std::memcpy((char*)dst + b, (const char*)src + b, n); });
    }
    for (auto& th : pool) th.join();
}

constexpr size_t kUploadPiece = 32u << 20;    // bytes per staging buffer (pinning memory costs ~0.3 ms per MiB: keep them small)

int ensure_upload_pipeline(mvs_ctx* c) {
    if (c->up_bytes) return MVS_OK;
    HIP_TRY(hipStreamCreateWithFlags(&c->up_stream, hipStreamNonBlocking));
    for (int i = 0; i < 2; ++i) {
        HIP_TRY(hipHostMalloc(&c->up_pinned[i], kUploadPiece, hipHostMallocDefault));
        HIP_TRY(hipEventCreateWithFlags(&c->up_done[i], hipEventDisableTiming));
    }
    c->up_bytes = kUploadPiece;
    return MVS_OK;
}

const Roctx& roctx() {
    static const Roctx r;
    return r;
}
Range::Range(const mvs_ctx* c, const char* name) {
    if (c && c->opt.markers && roctx().push) {
        roctx().push(name);
        on = true;
    }
}
Range::~Range() {
    if (on) roctx().pop();
}

int check_kernel(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(MVS_E_HIP, "%s launch: %s", what, hipGetErrorString(e));
    return MVS_OK;
}

}  // namespace

extern "C" {

const char* mvs_version(void) { return "mvs_hip 0.1 (gfx950)"; }
const char* mvs_last_error(void) { return g_err.c_str(); }

int mvs_device_count(int* count) {
    if (!count) return fail(MVS_E_INVALID, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) {
        *count = 0;
        return fail(MVS_E_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    }
    *count = n;
    return MVS_OK;
}

int mvs_ctx_create(int device, mvs_ctx** out) {
    if (!out) return fail(MVS_E_INVALID, "ctx out pointer is NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        return fail(MVS_E_HIP, "no HIP device available (%s); libmvs_hip has no CPU fallback",
                    e != hipSuccess ? hipGetErrorString(e) : "device count 0");
    if (device < 0 || device >= n) return fail(MVS_E_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_TRY(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(MVS_E_HIP, "device %d is %s; this library is built for gfx950 only", device, prop.gcnArchName);
    mvs_ctx* c = new (std::nothrow) mvs_ctx();
    if (!c) return fail(MVS_E_NOMEM, "out of host memory");
    c->device = device;
#ifdef MVS_ABLATIONS
    // The profiling build compiles the ping-pong kernels with extra code (time stamps, injected candidates), and with it
    // hipcc no longer keeps the fragment registers of the direct-B loop's hand-counted loads untouched: measured in round 5,
    // the ablation build's direct-B filter lost candidates at random (499 k +- 300 listed against 413 702, a few dozen kept
    // cells missing) while the shipped library -- which tools/check_isa.py gates -- is exact and deterministic.  So this build
    // runs both operands through LDS unless asked otherwise, and what it measures is the LDS-only kernel.
    c->opt.pairwise_bdirect = 0;
#endif
    options_from_env(c->opt);
    if (hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete c;
        return fail(MVS_E_HIP, "hipStreamCreate failed");
    }
    c->stream = c->own_stream;
    for (auto& ev : c->ev) {
        if (hipEventCreate(&ev) != hipSuccess) {
            mvs_ctx_destroy(c);
            return fail(MVS_E_HIP, "hipEventCreate failed");
        }
    }
    if (hipEventCreateWithFlags(&c->pinned_ev, hipEventDisableTiming) != hipSuccess) {
        mvs_ctx_destroy(c);
        return fail(MVS_E_HIP, "hipEventCreate failed");
    }
    // 8 counter slots; +256 B: the filter's stop flag; +1024 B: the re-check's eight round counters, 64 B apart
    if (hipMalloc((void**)&c->d_counter, 2048) != hipSuccess) {
        mvs_ctx_destroy(c);
        return fail(MVS_E_HIP, "hipMalloc failed");
    }
    *out = c;
    return MVS_OK;
}

int mvs_ctx_destroy(mvs_ctx* c) {
    if (!c) return MVS_OK;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->scratch) (void)hipFree(c->scratch);
    if (c->d_counter) (void)hipFree(c->d_counter);
    if (c->pw_thr) (void)hipFree(c->pw_thr);
    if (c->pw_tmp) (void)hipFree(c->pw_tmp);
    if (c->pw_sort) (void)hipFree(c->pw_sort);
    if (c->pw_out) (void)hipFree(c->pw_out);
    if (c->stage) (void)hipFree(c->stage);
    if (c->pw_coarse) (void)hipFree(c->pw_coarse);
    if (c->pw_coarse_fm) (void)hipFree(c->pw_coarse_fm);
    if (c->pw_planes_fm) (void)hipFree(c->pw_planes_fm);
    if (c->pw_need) (void)hipFree(c->pw_need);
    if (c->st_tlist) (void)hipFree(c->st_tlist);
    if (c->st_tlist_n) (void)hipFree(c->st_tlist_n);
    if (c->st_ends) (void)hipFree(c->st_ends);
    if (c->pw_rows) (void)hipFree(c->pw_rows);
    if (c->pw_fmeta) (void)hipFree(c->pw_fmeta);
    if (c->pw_cand) (void)hipFree(c->pw_cand);
    for (void* p : {c->st_raw, c->st_sorted, c->st_col[0], c->st_col[1], c->st_q[0], c->st_q[1], c->st_rowptr, c->st_counts,
                    c->st_dense, c->en_size, c->en_off, c->en_jac, c->en_first, c->en_par, c->st_enc[0], c->st_enc[1]})
        if (p) (void)hipFree(p);
    for (int i = 0; i < 2; ++i) {
        if (c->dl_pinned[i]) (void)hipHostFree(c->dl_pinned[i]);
        if (c->dl_done[i]) (void)hipEventDestroy(c->dl_done[i]);
        if (c->dl_block[i]) (void)hipEventDestroy(c->dl_block[i]);
    }
    for (hipEvent_t ev : c->dl_ready) if (ev) (void)hipEventDestroy(ev);
    if (c->dl_stream) (void)hipStreamDestroy(c->dl_stream);
    if (c->post_stream) (void)hipStreamDestroy(c->post_stream);
    if (c->cmp_done) (void)hipEventDestroy(c->cmp_done);
    if (c->pw_chdr) (void)hipFree(c->pw_chdr);
    if (c->pw_cent) (void)hipFree(c->pw_cent);
    for (void* p : {c->pw_tflag, c->pw_trow, c->pw_tlist, c->pw_cand2, c->pw_ttouch, c->pw_tnew})
        if (p) (void)hipFree(p);
    if (c->rb_pinned) (void)hipHostFree(c->rb_pinned);
    for (int i = 0; i < 2; ++i) {
        if (c->up_pinned[i]) (void)hipHostFree(c->up_pinned[i]);
        if (c->up_done[i]) (void)hipEventDestroy(c->up_done[i]);
    }
    if (c->up_stream) (void)hipStreamDestroy(c->up_stream);
    if (c->pinned) (void)hipHostFree(c->pinned);
    if (c->pinned_ev) (void)hipEventDestroy(c->pinned_ev);
    for (auto& ev : c->ev)
        if (ev) (void)hipEventDestroy(ev);
    plan_state_free(c);
    if (c->plan_tmp) (void)hipFree(c->plan_tmp);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
    return MVS_OK;
}

int mvs_ctx_set_option(mvs_ctx* c, const char* name, int64_t value) {
    if (!c || !name) return fail(MVS_E_INVALID, "NULL argument");
    if (std::strcmp(name, "plan_overlap") == 0) {            // host-side scheduling only: not one of the kernels' options
        if (value < 0 || value > 1) return fail(MVS_E_INVALID, "option plan_overlap: 0 or 1");
        c->plan_overlap = (int)value;
        return MVS_OK;
    }
    if (std::strcmp(name, "report_spin") == 0) {             // host-side waiting only
        if (value < 0 || value > 1000000) return fail(MVS_E_INVALID, "option report_spin: 0 .. 1000000 microseconds");
        c->report_spin = (int)value;
        return MVS_OK;
    }
    for (const OptionSpec& sp : kOptions)
        if (std::strcmp(sp.name, name) == 0) return apply_option(c->opt, sp, (long long)value);
    return fail(MVS_E_INVALID, "unknown option '%s'", name);
}

int mvs_ctx_get_option(const mvs_ctx* c, const char* name, int64_t* value) {
    if (!c || !name || !value) return fail(MVS_E_INVALID, "NULL argument");
    if (std::strcmp(name, "plan_overlap") == 0) {
        *value = c->plan_overlap;
        return MVS_OK;
    }
    if (std::strcmp(name, "report_spin") == 0) {
        *value = c->report_spin;
        return MVS_OK;
    }
    for (const OptionSpec& sp : kOptions)
        if (std::strcmp(sp.name, name) == 0) {
            *value = sp.ifield ? (int64_t)(c->opt.*(sp.ifield)) : (int64_t)(c->opt.*(sp.dfield));
            return MVS_OK;
        }
    return fail(MVS_E_INVALID, "unknown option '%s'", name);
}

int mvs_ctx_set_stream(mvs_ctx* c, void* hip_stream) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    c->stream = (hipStream_t)hip_stream;
    return MVS_OK;
}

int mvs_ctx_use_own_stream(mvs_ctx* c) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    c->stream = c->own_stream;
    return MVS_OK;
}

int mvs_ctx_synchronize(mvs_ctx* c) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    HIP_TRY(hipSetDevice(c->device));
    HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

int mvs_ctx_set_timing(mvs_ctx* c, int enabled) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    c->timing = enabled != 0;
    for (bool& v : c->ev_valid) v = false;
    return MVS_OK;
}

int mvs_ctx_pairwise_candidates(mvs_ctx* c, int64_t* candidates) {
    if (!c || !candidates) return fail(MVS_E_INVALID, "NULL argument");
    *candidates = (int64_t)c->last_candidates;
    return MVS_OK;
}

int mvs_ctx_pairwise_stats(mvs_ctx* c, int64_t* candidates, int64_t* flagged_tiles, int64_t* filter_tiles) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (candidates) *candidates = (int64_t)c->last_candidates;
    if (flagged_tiles) *flagged_tiles = (int64_t)c->last_flagged_tiles;
    if (filter_tiles) *filter_tiles = (int64_t)c->last_filter_tiles;
    return MVS_OK;
}

int mvs_ctx_kernel_ms(mvs_ctx* c, int which, float* ms) {
    if (!c || !ms || which < 0 || which > 4) return fail(MVS_E_INVALID, "bad argument");
    if (!c->ev_valid[which]) return fail(MVS_E_INVALID, "no timing recorded for kernel %d", which);
    // pairs 2, 3 and 4 share events with pair 1: filter = ev[2]..ev[5], re-check (with the gather / tile list / prune
    // passes in front of it) = ev[5]..ev[7], exact kernel on the flagged tiles (the last such launch) = ev[6]..ev[3]
    hipEvent_t b = which == 2 ? c->ev[2] : which == 3 ? c->ev[5] : which == 4 ? c->ev[6] : c->ev[2 * which];
    hipEvent_t e = which == 2 ? c->ev[5] : which == 3 ? c->ev[7] : which == 4 ? c->ev[3] : c->ev[2 * which + 1];
    HIP_TRY(hipEventSynchronize(e));
    HIP_TRY(hipEventElapsedTime(ms, b, e));
    return MVS_OK;
}

// -------------------------------------------------------------------------------------------------
// projection
// -------------------------------------------------------------------------------------------------
int mvs_project_csr(mvs_ctx* c, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                    int64_t n_samples, int d, int32_t* out, int mem_out) {
    return mvs_project_csr_stats(c, hashes, mem_hashes, offsets, n_samples, d, out, mem_out, nullptr, nullptr);
}

int mvs_project_csr_stats(mvs_ctx* c, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                          int64_t n_samples, int d, int32_t* out, int mem_out, int64_t* sumsq, int64_t* max_abs) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    const Range range(c, "mvs_project_csr");
    if ((sumsq == nullptr) != (max_abs == nullptr)) return fail(MVS_E_INVALID, "sumsq and max_abs go together");
    // sumsq lives where the sketches live: device array for device sketches, host array for host sketches
    int64_t* const sumsq_user = sumsq;
    DevBuf dsum;
    if (max_abs) *max_abs = 0;
    if (n_samples < 0 || d <= 0) return fail(MVS_E_INVALID, "n_samples=%lld d=%d", (long long)n_samples, d);
    if (!mem_ok(mem_hashes) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad mem flag");
    if (n_samples == 0) return MVS_OK;
    if (!offsets || !out) return fail(MVS_E_INVALID, "offsets/out is NULL");
    HIP_TRY(hipSetDevice(c->device));

    // units: runs of <= kProjUnitMax hashes, written straight into pinned memory
    if (n_samples >= (1LL << 31)) return fail(MVS_E_RANGE, "too many samples");
    size_t n_units = 0;
    bool all_single = true;
    for (int64_t s = 0; s < n_samples; ++s) {
        const int64_t b = offsets[s], e = offsets[s + 1];
        if (e < b) return fail(MVS_E_INVALID, "offsets not monotone at sample %lld", (long long)s);
        if (e - b >= (1LL << 31)) return fail(MVS_E_RANGE, "sample %lld has >= 2^31 hashes", (long long)s);
        // an empty sample gets one unit of zero hashes: the kernel then stores its row of zeros itself
        n_units += e == b ? 1 : (size_t)((e - b + mvs::kProjUnitMax - 1) / mvs::kProjUnitMax);
        all_single = all_single && (e - b) <= mvs::kProjUnitMax;
    }
    int rc = acquire_pinned(c, std::max<size_t>(n_units * sizeof(mvs::ProjUnit), 256));
    if (rc) return rc;
    mvs::ProjUnit* units = (mvs::ProjUnit*)c->pinned;
    {
        size_t w = 0;
        for (int64_t s = 0; s < n_samples; ++s) {
            const int64_t b = offsets[s], e = offsets[s + 1];
            const bool single = (e - b) <= mvs::kProjUnitMax;
            if (e == b) units[w++] = mvs::ProjUnit{b, 0, (int32_t)s, 1, 0};
            for (int64_t p = b; p < e; p += mvs::kProjUnitMax) {
                mvs::ProjUnit u;
                u.begin = p;
                u.count = (int32_t)std::min<int64_t>(mvs::kProjUnitMax, e - p);
                u.sample = (int32_t)s;
                u.single = single ? 1 : 0;
                u.pad = 0;
                units[w++] = u;
            }
        }
    }
    const int64_t total = offsets[n_samples];
    if (total > 0 && !hashes) return fail(MVS_E_INVALID, "hashes is NULL");

    DevBuf dh, dout;
    const uint64_t* d_hashes = hashes;
    // Host hash lists larger than one staging piece go up through the two-buffer pipeline: while piece k is on the
    // link, piece k+1 is being copied into pinned memory and the samples that piece k-1 completed are being projected.
    const bool pipelined = mem_hashes == MVS_MEM_HOST && (size_t)total * 8 > kUploadPiece;
    if (mem_hashes == MVS_MEM_HOST) {
        HIP_TRY(dh.alloc((size_t)total * 8));
        if (!pipelined) HIP_TRY(hipMemcpyAsync(dh.p, hashes, (size_t)total * 8, hipMemcpyHostToDevice, c->stream));
        d_hashes = (const uint64_t*)dh.p;
    }
    int32_t* d_out = out;
    const size_t out_bytes = (size_t)n_samples * (size_t)d * 4;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc(out_bytes));
        d_out = (int32_t*)dout.p;
        if (sumsq) {
            HIP_TRY(dsum.alloc((size_t)n_samples * 8));
            sumsq = (int64_t*)dsum.p;
        }
    }
    const size_t ubytes = n_units * sizeof(mvs::ProjUnit);
    rc = ensure_scratch(c, std::max<size_t>(ubytes, 256));
    if (rc) return rc;
    if (n_units) {
        HIP_TRY(hipMemcpyAsync(c->scratch, units, ubytes, hipMemcpyHostToDevice, c->stream));
        HIP_TRY(hipEventRecord(c->pinned_ev, c->stream));
        c->pinned_busy = true;
    }

    // samples cut into several units are combined with atomics and start from zero; single units store
    if (!all_single) HIP_TRY(hipMemsetAsync(d_out, 0, out_bytes, c->stream));
    const bool fused = sumsq != nullptr && all_single;         // statistics inside the projection kernel
    if (fused) {
        HIP_TRY(hipMemsetAsync(sumsq, 0, (size_t)n_samples * 8, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    }
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[0], c->stream));
    const int nblk = (d + 63) / 64;
    // kernel variant (launch_project): four blocks per wave sharing the first splitmix64 round where the dimension
    // fills them (8.97 vs 9.44 ms on 10k x 50k hashes, d = 2048), else two or one block per wave; option
    // project_variant forces one
    int bpw = (nblk % 4 == 0 && nblk >= 8) ? 14 : (nblk >= 2 ? 2 : 1);
    if (c->opt.project_variant == 14 && nblk >= 4) bpw = 14;
    if (c->opt.project_variant == 12 && nblk >= 2) bpw = 12;
    if (c->opt.project_variant == 2 && nblk >= 2) bpw = 2;
    if (c->opt.project_variant == 1) bpw = 1;
    if (!pipelined) {
        mvs::launch_project(c->stream, d_hashes, (const mvs::ProjUnit*)c->scratch, (int64_t)n_units, d, d_out, bpw,
                            fused ? (unsigned long long*)sumsq : nullptr, fused ? c->d_counter : nullptr);
        rc = check_kernel("k_project");
        if (rc) return rc;
    } else {
        rc = ensure_upload_pipeline(c);
        if (rc) return rc;
        // the unit list is in hash order: units [u_done, u_next) are those whose hashes the pieces sent so far cover
        size_t u_done = 0;
        const size_t total_bytes = (size_t)total * 8;
        int piece = 0;
        for (size_t off = 0; off < total_bytes; off += kUploadPiece, ++piece) {
            const int b = piece & 1;
            const size_t len = std::min(kUploadPiece, total_bytes - off);
            if (piece >= 2) HIP_TRY(hipEventSynchronize(c->up_done[b]));      // staging buffer b is free again
            parallel_copy(c->up_pinned[b], (const char*)hashes + off, len);
            HIP_TRY(hipMemcpyAsync((char*)dh.p + off, c->up_pinned[b], len, hipMemcpyHostToDevice, c->up_stream));
            HIP_TRY(hipEventRecord(c->up_done[b], c->up_stream));
            const int64_t covered = (int64_t)((off + len) / 8);
            size_t u_next = u_done;
            while (u_next < n_units && units[u_next].begin + units[u_next].count <= covered) ++u_next;
            if (u_next > u_done) {
                HIP_TRY(hipStreamWaitEvent(c->stream, c->up_done[b], 0));
                mvs::launch_project(c->stream, d_hashes, (const mvs::ProjUnit*)c->scratch + u_done, (int64_t)(u_next - u_done), d,
                                    d_out, bpw, fused ? (unsigned long long*)sumsq : nullptr, fused ? c->d_counter : nullptr);
                rc = check_kernel("k_project");
                if (rc) return rc;
                u_done = u_next;
            }
        }
        if (u_done != n_units) return fail(MVS_E_INVALID, "internal: %zu of %zu projection units launched", u_done, n_units);
    }
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[1], c->stream));
        c->ev_valid[0] = true;
    }
    if (sumsq) {
        if (fused) {
            unsigned long long m = 0;
            {
                const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
                if (rb_rc) return rb_rc;
            }
            *max_abs = (int64_t)m;
        } else {   // some sample spans several units: its entries are final only now
            rc = mvs_sketch_stats(c, d_out, MVS_MEM_DEVICE, n_samples, d, sumsq, MVS_MEM_DEVICE, max_abs);
            if (rc) return rc;
        }
    }
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, c->stream));
        if (sumsq_user) HIP_TRY(hipMemcpyAsync(sumsq_user, sumsq, (size_t)n_samples * 8, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    } else if (mem_hashes == MVS_MEM_HOST) {
        HIP_TRY(hipStreamSynchronize(c->stream));   // staging buffer is freed on return
    } else {
        // the unit list lives in ctx scratch, which stays valid; nothing to wait for
    }
    return MVS_OK;
}

int mvs_sketch_sumsq(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n, int d, int64_t* sumsq,
                     int mem_out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n < 0 || d <= 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sketches || !sumsq) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n * d * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int64_t* d_out = sumsq;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n * 8));
        d_out = (int64_t*)dout.p;
    }
    mvs::launch_sumsq(c->stream, d_in, n, d, d_out);
    int rc = check_kernel("k_sumsq");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(sumsq, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    if (mem_out == MVS_MEM_HOST || mem_in == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

namespace {

// "%g" keeps 6 significant digits: x -> the decimal r * 10^-j (r an integer of 6 digits, round-half-even on the exact
// binary value of x as printf does) -> the double nearest to that decimal (what strtod returns) -> squared.
// The product x * 10^j is rounded once; only when it lands exactly on k + 0.5 can the true product lie on either side,
// and the fma residual says which (rounding is monotonic, so a product off the tie is on the true side of it).  An
// exponent estimate that is off by one next to a power of ten yields the same decimal (r = 10^6 is renormalised).
__device__ double norm_sq_from_text(long long sumsq, int d) {
    if (sumsq <= 0) return 0.0;
    const double x = sqrt((double)sumsq / (double)d);
    constexpr double p10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                                1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};
    int e = 0;                                   // 10^e <= x < 10^(e+1), up to the off-by-one noted above
    if (x >= 1.0) {
        while (e < 21 && x >= p10[e + 1]) ++e;
    } else {
        double y = x;
        while (e > -16 && y < 1.0) {
            y *= 10.0;
            --e;
        }
    }
    int j = 5 - e;                               // x * 10^j has 6 digits before the point
    double m, err;
    if (j >= 0) {
        m = x * p10[j];
        err = fma(x, p10[j], -m);                // exact: true product = m + err
    } else {
        m = x / p10[-j];
        err = -fma(m, p10[-j], -x);              // sign of (true quotient - m)
    }
    double r = rint(m);                          // half-even
    const double fl = floor(m);
    if (m - fl == 0.5 && err != 0.0) r = err > 0.0 ? fl + 1.0 : fl;
    if (r >= 1e6) {
        r = 1e5;
        --j;
    }
    const double v = j >= 0 ? r / p10[j] : r * p10[-j];
    return v * v;
}

__global__ __launch_bounds__(256) void k_norms_sq_text(const int64_t* __restrict__ sumsq, int64_t n, int d,
                                                       double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = norm_sq_from_text(sumsq[i], d);
}

}  // namespace

int mvs_norms_sq_text(mvs_ctx* c, const int64_t* sumsq, int64_t n, int d, double* out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n < 0 || d <= 0) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sumsq || !out) return fail(MVS_E_INVALID, "NULL buffer");
    if ((n + 255) / 256 > 0x7fffffffLL) return fail(MVS_E_INVALID, "too many entries");
    HIP_TRY(hipSetDevice(c->device));
    hipLaunchKernelGGL(k_norms_sq_text, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream, sumsq, n, d, out);
    return check_kernel("k_norms_sq_text");
}

int mvs_sketch_stats(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n, int d, int64_t* sumsq, int mem_out,
                     int64_t* max_abs) {
    if (!c || !max_abs) return fail(MVS_E_INVALID, "NULL argument");
    *max_abs = 0;
    if (n < 0 || d <= 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!sketches || !sumsq) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n * d * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int64_t* d_out = sumsq;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n * 8));
        d_out = (int64_t*)dout.p;
    }
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_stats(c->stream, d_in, n, d, d_out, c->d_counter);
    int rc = check_kernel("k_stats");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(sumsq, d_out, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_sketch_saturate_i16(mvs_ctx* c, const int32_t* sketches, int mem_in, int64_t n_elems, int16_t* out,
                            int mem_out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n_elems < 0 || !mem_ok(mem_in) || !mem_ok(mem_out)) return fail(MVS_E_INVALID, "bad argument");
    if (n_elems == 0) return MVS_OK;
    if (!sketches || !out) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf din, dout;
    const int32_t* d_in = sketches;
    if (mem_in == MVS_MEM_HOST) {
        HIP_TRY(din.alloc((size_t)n_elems * 4));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n_elems * 4, hipMemcpyHostToDevice, c->stream));
        d_in = (const int32_t*)din.p;
    }
    int16_t* d_out = out;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc((size_t)n_elems * 2));
        d_out = (int16_t*)dout.p;
    }
    mvs::launch_saturate_i16(c->stream, d_in, n_elems, d_out);
    int rc = check_kernel("k_saturate_i16");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST)
        HIP_TRY(hipMemcpyAsync(out, d_out, (size_t)n_elems * 2, hipMemcpyDeviceToHost, c->stream));
    if (mem_out == MVS_MEM_HOST || mem_in == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

// -------------------------------------------------------------------------------------------------
// pairwise
// -------------------------------------------------------------------------------------------------
int mvs_sketch_max_abs(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n_elems,
                       int64_t* max_abs) {
    if (!c || !max_abs) return fail(MVS_E_INVALID, "NULL argument");
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n_elems < 0)
        return fail(MVS_E_INVALID, "bad argument");
    *max_abs = 0;
    if (n_elems == 0) return MVS_OK;
    if (!sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, (size_t)n_elems * elem_bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, (size_t)n_elems * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_max_abs(c->stream, d_in, elem_bytes, n_elems, c->d_counter);
    int rc = check_kernel("k_max_abs");
    if (rc) return rc;
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_limbs_for_max_abs(int64_t max_abs) {
    if (max_abs < 0) max_abs = -max_abs;
    if (max_abs <= 127) return 1;
    if (max_abs <= 32639) return 2;      // 127 * (1 + 256)
    if (max_abs <= 8355711) return 3;    // 127 * (1 + 256 + 65536)
    return 4;                            // exact mod 2^32 for every int32
}

int mvs_limb_geometry(int64_t n, int d, int limbs, int64_t* n_alloc, int* d_pad, size_t* bytes) {
    if (n < 0 || d <= 0 || !mvs::limb_code_ok(limbs)) return fail(MVS_E_INVALID, "bad argument");
    // tiles are up to 256 rows: pad to a multiple of 256 plus one spare tile
    const int64_t na = (n + 255) / 256 * 256 + 256;
    const int dp = (d + mvs::kBK - 1) / mvs::kBK * mvs::kBK;
    if (n_alloc) *n_alloc = na;
    if (d_pad) *d_pad = dp;
    if (bytes) *bytes = (size_t)na * (size_t)mvs::planes_of(limbs) * (size_t)dp;
    return MVS_OK;
}

int mvs_limb_split(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n_rows, int d, int limbs,
                   int8_t* planes, int d_pad, int64_t row_offset) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n_rows < 0 || d <= 0 || !mvs::limb_code_ok(limbs) ||
        d_pad < d || d_pad % mvs::kBK != 0 || row_offset < 0)
        return fail(MVS_E_INVALID, "bad argument");
    if (n_rows == 0) return MVS_OK;
    if (!sketches || !planes) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {   // grow-only staging buffer of the context (no allocation per chunk)
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, (size_t)n_rows * d * elem_bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, (size_t)n_rows * d * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    mvs::launch_limb_split(c->stream, d_in, elem_bytes, n_rows, d, limbs, planes, d_pad, row_offset);
    int rc = check_kernel("k_limb_split");
    if (rc) return rc;
    if (mem == MVS_MEM_HOST) HIP_TRY(hipStreamSynchronize(c->stream));
    return MVS_OK;
}

int mvs_sketch_set_create(mvs_ctx* c, const void* sketches, int elem_bytes, int mem, int64_t n, int d,
                          mvs_sketch_set** out) {
    if (!c || !out) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem) || n < 0 || d <= 0)
        return fail(MVS_E_INVALID, "bad argument");
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    if (n > 0 && !sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    HIP_TRY(hipSetDevice(c->device));
    // stage once if the input is on the host
    DevBuf din;
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST && n > 0) {
        HIP_TRY(din.alloc((size_t)n * d * elem_bytes));
        HIP_TRY(hipMemcpyAsync(din.p, sketches, (size_t)n * d * elem_bytes, hipMemcpyHostToDevice, c->stream));
        d_in = din.p;
    }
    int64_t max_abs = 0;
    int rc = mvs_sketch_max_abs(c, d_in, elem_bytes, MVS_MEM_DEVICE, n * d, &max_abs);
    if (rc) return rc;
    int limbs = mvs_limbs_for_max_abs(max_abs);
    // The 3-pass Karatsuba scheme (63 * (1 + 128): digits in [-64,63], their sum in int8) is exact and tested but
    // measures 9-19 % SLOWER than two base-256 limbs on MI355X (25 % fewer MFMAs, 1.5x the LDS traffic), so it is
    // opt-in: option enable_k3.
    if (c->opt.enable_k3 && max_abs > 127 && max_abs <= 8127) limbs = MVS_LIMBS_K3;
    int64_t n_alloc = 0;
    int d_pad = 0;
    size_t bytes = 0;
    mvs_limb_geometry(n, d, limbs, &n_alloc, &d_pad, &bytes);
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    if (hipMalloc((void**)&s->owned, bytes) != hipSuccess) {
        delete s;
        return fail(MVS_E_NOMEM, "hipMalloc of %zu bytes of limb planes failed", bytes);
    }
    s->ctx = c;
    s->planes = s->owned;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    hipError_t e = hipMemsetAsync(s->owned, 0, bytes, c->stream);
    if (e == hipSuccess) {
        rc = mvs_limb_split(c, d_in, elem_bytes, MVS_MEM_DEVICE, n, d, limbs, s->owned, d_pad, 0);
        if (rc == MVS_OK && hipStreamSynchronize(c->stream) != hipSuccess) rc = fail(MVS_E_HIP, "sync failed");
    } else {
        rc = fail(MVS_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e));
    }
    if (rc) {
        mvs_sketch_set_destroy(s);
        return rc;
    }
    *out = s;
    return MVS_OK;
}

int mvs_sketch_set_from_planes(mvs_ctx* c, const int8_t* planes, int64_t n, int64_t n_alloc, int d, int d_pad,
                               int limbs, mvs_sketch_set** out) {
    if (!c || !out || !planes) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    int64_t need_alloc = 0;
    int need_pad = 0;
    if (mvs_limb_geometry(n, d, limbs, &need_alloc, &need_pad, nullptr)) return MVS_E_INVALID;
    if (n_alloc < need_alloc || d_pad != need_pad)
        return fail(MVS_E_INVALID, "plane buffer geometry: need n_alloc >= %lld and d_pad == %d",
                    (long long)need_alloc, need_pad);
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    s->ctx = c;
    s->planes = planes;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    *out = s;
    return MVS_OK;
}

int mvs_sketch_set_alloc(mvs_ctx* c, int64_t n, int d, int limbs, mvs_sketch_set** out) {
    if (!c || !out) return fail(MVS_E_INVALID, "NULL argument");
    *out = nullptr;
    if (n < 0 || d <= 0 || !mvs::limb_code_ok(limbs)) return fail(MVS_E_INVALID, "bad argument");
    if (n >= (1LL << 31) - 256) return fail(MVS_E_RANGE, "n too large for int32 row/col indices");
    HIP_TRY(hipSetDevice(c->device));
    int64_t n_alloc = 0;
    int d_pad = 0;
    size_t bytes = 0;
    mvs_limb_geometry(n, d, limbs, &n_alloc, &d_pad, &bytes);
    mvs_sketch_set* s = new (std::nothrow) mvs_sketch_set();
    if (!s) return fail(MVS_E_NOMEM, "out of host memory");
    if (hipMalloc((void**)&s->owned, bytes) != hipSuccess) {
        delete s;
        return fail(MVS_E_NOMEM, "hipMalloc of %zu bytes of limb planes failed", bytes);
    }
    s->ctx = c;
    s->planes = s->owned;
    s->n = n;
    s->n_alloc = n_alloc;
    s->d = d;
    s->d_pad = d_pad;
    s->limbs = limbs;
    s->id = ++g_set_ids;
    if (hipMemsetAsync(s->owned, 0, bytes, c->stream) != hipSuccess) {
        mvs_sketch_set_destroy(s);
        return fail(MVS_E_HIP, "hipMemsetAsync failed");
    }
    *out = s;
    return MVS_OK;
}

namespace {
// Rows [lo, hi) of an owned set are about to be rewritten.  If the context holds data derived from the set's present
// contents (coarse plane, fragment-major copies) and the range is a small part of it, the set keeps its generation and
// remembers the range: refresh_derived() re-derives just those rows before the next comparison.  A search front end
// that appends its queries behind a resident database (search.py: SearchIndex) thus keeps the database's coarse plane --
// bumping the generation made every search rebuild it (6 ms per 10^6 sketches) or fall back to the exact kernels.
void note_rows_rewritten(mvs_sketch_set* s, int64_t lo, int64_t hi) {
    mvs_ctx* c = s->ctx;
    // (the "a block of few rows went to the exact kernel for lack of a coarse plane" marker counts as well: it is what makes
    // the SECOND such block build the plane, and it must survive the upload of that block's rows)
    const bool cached = (c->coarse_id == s->id && c->coarse_gen == s->gen) || (c->planes_fm_id == s->id && c->planes_fm_gen == s->gen) ||
                        (c->few_rows_id == s->id && c->few_rows_gen == s->gen);
    const int64_t u_lo = s->dirty_hi > s->dirty_lo ? std::min(s->dirty_lo, lo) : lo;
    const int64_t u_hi = s->dirty_hi > s->dirty_lo ? std::max(s->dirty_hi, hi) : hi;
    if (cached && (u_hi - u_lo) * 8 <= s->n) {
        s->dirty_lo = u_lo;
        s->dirty_hi = u_hi;
        return;
    }
    ++s->gen;   // derived data of the old contents is stale as a whole
    s->dirty_lo = s->dirty_hi = 0;
}
}  // namespace

int mvs_sketch_set_fill(mvs_sketch_set* s, const void* sketches, int elem_bytes, int mem, int64_t row_offset,
                        int64_t n_rows) {
    if (!s || !s->owned) return fail(MVS_E_INVALID, "set is NULL or not owned by the library");
    if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > s->n)
        return fail(MVS_E_INVALID, "rows [%lld,%lld) outside the set", (long long)row_offset,
                    (long long)(row_offset + n_rows));
    if (n_rows > 0) note_rows_rewritten(s, row_offset, row_offset + n_rows);
    return mvs_limb_split(s->ctx, sketches, elem_bytes, mem, n_rows, s->d, s->limbs, s->owned, s->d_pad, row_offset);
}

int mvs_sketch_set_fill_stats(mvs_sketch_set* s, const void* sketches, int elem_bytes, int mem, int64_t row_offset,
                              int64_t n_rows, int64_t* max_abs) {
    if (!s || !s->owned) return fail(MVS_E_INVALID, "set is NULL or not owned by the library");
    if (!max_abs) return fail(MVS_E_INVALID, "max_abs is NULL");
    *max_abs = 0;
    if ((elem_bytes != 4 && elem_bytes != 2) || !mem_ok(mem)) return fail(MVS_E_INVALID, "bad argument");
    if (row_offset < 0 || n_rows < 0 || row_offset + n_rows > s->n)
        return fail(MVS_E_INVALID, "rows [%lld,%lld) outside the set", (long long)row_offset,
                    (long long)(row_offset + n_rows));
    if (n_rows == 0) return MVS_OK;
    if (!sketches) return fail(MVS_E_INVALID, "sketches is NULL");
    mvs_ctx* c = s->ctx;
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)n_rows * s->d * elem_bytes;
    const void* d_in = sketches;
    if (mem == MVS_MEM_HOST) {   // one upload serves both kernels
        int rc0 = ensure_buf(c, &c->stage, &c->stage_bytes, bytes);
        if (rc0) return rc0;
        HIP_TRY(hipMemcpyAsync(c->stage, sketches, bytes, hipMemcpyHostToDevice, c->stream));
        d_in = c->stage;
    }
    note_rows_rewritten(s, row_offset, row_offset + n_rows);
    HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    mvs::launch_max_abs(c->stream, d_in, elem_bytes, n_rows * s->d, c->d_counter);
    int rc = check_kernel("k_max_abs");
    if (rc) return rc;
    mvs::launch_limb_split(c->stream, d_in, elem_bytes, n_rows, s->d, s->limbs, s->owned, s->d_pad, row_offset);
    rc = check_kernel("k_limb_split");
    if (rc) return rc;
    unsigned long long m = 0;
    {
        const int rb_rc = read_back(c, c->stream, {{&m, c->d_counter, 8}});
        if (rb_rc) return rb_rc;
    }
    *max_abs = (int64_t)m;
    return MVS_OK;
}

int mvs_sketch_set_info(const mvs_sketch_set* s, int64_t* n, int* d, int* limbs, int64_t* n_alloc, int* d_pad) {
    if (!s) return fail(MVS_E_INVALID, "set is NULL");
    if (n) *n = s->n;
    if (d) *d = s->d;
    if (limbs) *limbs = s->limbs;
    if (n_alloc) *n_alloc = s->n_alloc;
    if (d_pad) *d_pad = s->d_pad;
    return MVS_OK;
}

int mvs_sketch_set_planes(mvs_sketch_set* s, int8_t** planes) {
    if (!s || !planes) return fail(MVS_E_INVALID, "NULL argument");
    if (!s->owned) return fail(MVS_E_INVALID, "the set is a view of a caller-owned buffer");
    *planes = s->owned;
    return MVS_OK;
}

int mvs_sketch_set_touch(mvs_sketch_set* s) {
    if (!s) return fail(MVS_E_INVALID, "set is NULL");
    ++s->gen;
    s->dirty_lo = s->dirty_hi = 0;
    return MVS_OK;
}

int mvs_sketch_set_destroy(mvs_sketch_set* s) {
    if (!s) return MVS_OK;
    if (s->owned) {
        (void)hipSetDevice(s->ctx->device);
        (void)hipStreamSynchronize(s->ctx->stream);
        (void)hipFree(s->owned);
    }
    delete s;
    return MVS_OK;
}

namespace {

// the rows of `s` that were rewritten since the context's derived data was built (note_rows_rewritten): re-derive exactly
// those rows in every cache that belongs to the set's present generation, coarse plane first (its fragment-major copy is
// made from it).  The fragment-major layouts hold 16 rows per KiB, so the range is widened to whole groups of 16.
int refresh_derived(mvs_ctx* c, const mvs_sketch_set* cs) {
    mvs_sketch_set* s = const_cast<mvs_sketch_set*>(cs);
    if (s->dirty_hi <= s->dirty_lo) return MVS_OK;
    const int64_t lo = s->dirty_lo & ~(int64_t)15, hi = std::min<int64_t>(s->n_alloc, (s->dirty_hi + 15) & ~(int64_t)15);
    const int64_t count = hi - lo, dp = s->d_pad;
    s->dirty_lo = s->dirty_hi = 0;
    if (s->limbs == 2 && c->coarse_id == s->id && c->coarse_gen == s->gen) {
        if (c->coarse_mode != c->opt.coarse_radix) {
            c->coarse_id = 0;                                  // another radix rule was asked for: rebuilt as a whole anyway
        } else {
            const int64_t valid = std::max<int64_t>(0, std::min<int64_t>(count, s->n - lo));
            mvs::launch_coarse_build(c->stream, s->planes + lo * 2 * dp, valid, count, s->d_pad, (int8_t*)c->pw_coarse + lo * dp,
                                     (mvs::CoarseRow*)c->pw_rows + lo, c->opt.coarse_radix);
            int rc = check_kernel("k_coarse_build(rows)");
            if (rc) return rc;
            if (c->coarse_fm_valid) {
                mvs::launch_coarse_fm(c->stream, (const int8_t*)c->pw_coarse + lo * dp, count, s->d_pad, (int8_t*)c->pw_coarse_fm + lo * dp);
                rc = check_kernel("k_coarse_fm(rows)");
                if (rc) return rc;
            }
        }
    }
    if (s->limbs == 2 && c->planes_fm_id == s->id && c->planes_fm_gen == s->gen) {
        mvs::launch_coarse_fm(c->stream, s->planes + lo * 2 * dp, count, s->d_pad, (int8_t*)c->pw_planes_fm + lo * 2 * dp, 2);
        const int rc = check_kernel("k_coarse_fm(limb planes, rows)");
        if (rc) return rc;
    }
    return MVS_OK;
}

// coarse plane + row statistics of `s`, cached in the context until the set (or its contents) changes
int prepare_coarse(mvs_ctx* c, const mvs_sketch_set* s) {
    const int rr = refresh_derived(c, s);
    if (rr) return rr;
    if (c->coarse_id == s->id && c->coarse_gen == s->gen && c->coarse_mode == c->opt.coarse_radix) return MVS_OK;
    c->coarse_id = 0;
    c->coarse_fm_valid = false;
    int rc = ensure_buf(c, &c->pw_coarse, &c->pw_coarse_bytes, (size_t)s->n_alloc * (size_t)s->d_pad);
    if (rc) return rc;
    rc = ensure_buf(c, &c->pw_rows, &c->pw_rows_bytes, (size_t)s->n_alloc * sizeof(mvs::CoarseRow));
    if (rc) return rc;
    mvs::launch_coarse_build(c->stream, s->planes, s->n, s->n_alloc, s->d_pad, (int8_t*)c->pw_coarse,
                             (mvs::CoarseRow*)c->pw_rows, c->opt.coarse_radix);
    rc = check_kernel("k_coarse_build");
    if (rc) return rc;
    c->coarse_id = s->id;
    c->coarse_gen = s->gen;
    c->coarse_mode = c->opt.coarse_radix;
    return MVS_OK;
}

// The fragment-major copies are a convenience of the matrix-core kernels (which also read the row-major planes, slower):
// they are only made when they fit beside what the comparison itself still has to allocate -- candidate lists, kept
// cells, the dense matrix of a streamed result -- i.e. when growing the buffer leaves the larger of 2 GiB and 1/16 of
// the card free.  A copy that does not fit is skipped, never an error.
bool fm_copy_fits(size_t have_bytes, size_t want_bytes) {
    if (want_bytes <= have_bytes) return true;
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess) return false;
    const size_t reserve = std::max<size_t>((size_t)2 << 30, total_b / 16);
    return free_b + have_bytes >= want_bytes + reserve;      // ensure_buf frees the old buffer before it allocates
}

// the fragment-major copy of the cached coarse plane (after prepare_coarse), built the first time a streaming search filter
// runs on the set; *made = false when it was skipped for lack of room (fm_copy_fits)
int prepare_coarse_fm(mvs_ctx* c, const mvs_sketch_set* s, bool* made) {
    *made = true;
    if (c->coarse_fm_valid) return MVS_OK;
    if (!fm_copy_fits(c->pw_coarse_fm_bytes, (size_t)s->n_alloc * (size_t)s->d_pad)) {
        *made = false;
        return MVS_OK;
    }
    int rc = ensure_buf(c, &c->pw_coarse_fm, &c->pw_coarse_fm_bytes, (size_t)s->n_alloc * (size_t)s->d_pad);
    if (rc) return rc;
    mvs::launch_coarse_fm(c->stream, (const int8_t*)c->pw_coarse, s->n_alloc, s->d_pad, (int8_t*)c->pw_coarse_fm);
    rc = check_kernel("k_coarse_fm");
    if (rc) return rc;
    c->coarse_fm_valid = true;
    return MVS_OK;
}

// the fragment-major copy of the set's limb planes, cached in the context until the set (or its contents) changes:
// a.planes_fm is set when the exact kernel that reads it will run for this block
int attach_planes_fm(mvs_ctx* c, const mvs_sketch_set* s, mvs::PairwiseArgs& a, bool wanted) {
    a.planes_fm = nullptr;
    const int rr = refresh_derived(c, s);
    if (rr) return rr;
    if (!wanted || !c->opt.fragment_major || s->limbs != 2) return MVS_OK;
    if (!(c->planes_fm_id == s->id && c->planes_fm_gen == s->gen)) {
        c->planes_fm_id = 0;
        if (!fm_copy_fits(c->pw_planes_fm_bytes, (size_t)s->n_alloc * 2 * (size_t)s->d_pad)) return MVS_OK;   // row-major kernels
        int rc = ensure_buf(c, &c->pw_planes_fm, &c->pw_planes_fm_bytes, (size_t)s->n_alloc * 2 * (size_t)s->d_pad);
        if (rc) return rc;
        mvs::launch_coarse_fm(c->stream, s->planes, s->n_alloc, s->d_pad, (int8_t*)c->pw_planes_fm, 2);
        rc = check_kernel("k_coarse_fm(limb planes)");
        if (rc) return rc;
        c->planes_fm_id = s->id;
        c->planes_fm_gen = s->gen;
    }
    a.planes_fm = (const int8_t*)c->pw_planes_fm;
    return MVS_OK;
}

// One comparison of rows [rb,re) x columns [cb,ce) appending to `raw` (device) after the first `start`
// cells; the running count is left in c->d_counter[0].  Two-stage (filter + exact re-check of the
// candidates) when the set allows it, otherwise the exact MFMA / vector-ALU kernel on every cell.
// Streamed output (mvs_pairwise_stream): kept cells as packed 64-bit words (mvs_internal.h: PairwiseArgs::packed) in a
// grow-only buffer of the context that the launch sizes itself, so that a comparison never has to be repeated because its
// output did not fit: the two-stage comparison sizes it from the candidate count between the filter and the re-check
// (a kept cell is a candidate or the mirror image of one), the exact kernel's caller sizes the row block for the worst case.
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
    unsigned int* ext_flags = nullptr;   // in: tile flags live here (this launch's tile rows of a larger grid) instead of in
                                         // the context's own array -- the streamed pipeline keeps one array for the whole matrix
};

void fill_args(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, int keep_mode, int64_t rb, int64_t re, int64_t cb,
               int64_t ce, bool symmetric, bool mirror_all, double keep_coeff, mvs::PairwiseArgs& a) {
    a.planes = s->planes;
    a.planes_fm = nullptr;
    a.n = s->n;
    a.n_alloc = s->n_alloc;
    a.d = s->d;
    a.d_pad = s->d_pad;
    a.limbs = s->limbs;
    a.row_begin = rb;
    a.row_end = re;
    a.sym_begin = rb;
    a.sym_end = re;
    a.col_begin = cb;
    a.col_end = ce;
    a.norms_sq = d_n2;
    a.keep_mode = keep_mode;
    a.keep_coeff = keep_coeff;
    a.counter = c->d_counter;
    a.dots = nullptr;
    a.mirror_all = mirror_all ? 1 : 0;
    a.debug_flags = c->opt.pairwise_debug;
    a.map_mode = c->opt.pairwise_map;
    a.stamps = nullptr;
    a.symmetric = (symmetric && c->opt.pairwise_symmetric) ? 1 : 0;   // the launcher checks the alignment
}

// the running cell count starts at `start` (appending calls); kKeepCount: it stays what the device counter holds (a block
// plan appends block after block without the host ever learning the count in between)
constexpr unsigned long long kKeepCount = ~0ULL;
int set_cell_count(mvs_ctx* c, unsigned long long start) {
    if (start == kKeepCount) return MVS_OK;
    c->h_start = start;   // outlives the asynchronous copy
    if (start == 0) HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
    else HIP_TRY(hipMemcpyAsync(c->d_counter, &c->h_start, 8, hipMemcpyHostToDevice, c->stream));
    return MVS_OK;
}

// Stage 1: coarse plane, filter constants, the filter pass, candidate regions -> list, tile flags -> list, pruning.
// `a` comes in with geometry and keep test filled (fill_args); outputs (cells / packed / dense) are the later stages'.
// hold_all: size the candidate list for whatever the filter may pass on, so that it never runs twice (streamed output).
// Returns MVS_OK with `ts` filled, kNeedExact when the filter gave up (the exact kernel should do the block), or an error.
int two_stage_filter(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, double keep_coeff, int64_t capacity_hint,
                     bool hold_all, unsigned long long start, mvs::PairwiseArgs& a, TwoStage& ts) {
    const double block_cells = (double)(a.row_end - a.row_begin) * (double)(a.col_end - a.col_begin);
    int rc = prepare_coarse(c, s);
    if (rc) return rc;
    rc = ensure_buf(c, &c->pw_fmeta, &c->pw_fmeta_bytes, (size_t)s->n_alloc * sizeof(float4));
    if (rc) return rc;
    mvs::launch_filter_meta(c->stream, (const mvs::CoarseRow*)c->pw_rows, d_n2, s->n, s->n_alloc, s->d, keep_coeff,
                            (float4*)c->pw_fmeta);
    rc = check_kernel("k_filter_meta");
    if (rc) return rc;
    const bool forced = c->opt.pairwise_filter == 2;
    ts.tiles = mvs::filter_flags_tiles(a, c->opt);
    rc = attach_planes_fm(c, s, a, ts.tiles);       // the flagged tiles go to the ping-pong exact kernel
    if (rc) return rc;
    mvs::filter_tile_grid(a, &ts.n_tr, &ts.n_tc);
    // the symmetric schedule computes the tiles on and above the diagonal of the square only
    const bool sym = a.symmetric && (a.row_begin - a.col_begin) % 256 == 0 && !a.mirror_all;
    // (tile row t of the launch skips the tiles strictly below the square's diagonal: (row_begin - sym_begin) / 256 + t of them)
    const double r0_tiles = (double)(a.row_begin - a.sym_begin) / 256.0;
    const double tiles_to_do = std::max(1.0, (double)ts.n_tr * (double)ts.n_tc -
                                                 (sym ? (double)ts.n_tr * r0_tiles + 0.5 * (double)ts.n_tr * (double)(ts.n_tr - 1) : 0.0));
    c->last_filter_tiles = (long long)tiles_to_do;
    c->last_flagged_tiles = 0;
    // Listing: re-checking a candidate costs about as much as 80-300 cells of the exact kernel (by how well the rows
    // cache) and the filter pass a third of it.
    //  * Tile-granular (ping-pong filter): a wave with more than tile_dense_thr candidates flags its 256 x 256 tile for
    //    the exact kernel, so the list holds at most 8 x tile_dense_thr pairs per tile and needs no global limit; the
    //    launch stops only when nearly every tile is flagged (the exact kernel alone is then faster: filter + f x exact
    //    against exact, break-even near f = 0.7), and that set's later blocks skip the filter.
    //  * Otherwise (ring filters on small blocks, tile_dense_thr = 0): beyond ~1/128 of the block's cells in the list the
    //    filter tiles and the re-check give up and the exact kernel does the block, as up to round 3.
    // Forced mode (pairwise_filter = 2, tests) has no limit of either kind.
    const unsigned long long limit =
        (forced || ts.tiles) ? ~0ULL : (unsigned long long)std::min(268435456.0, std::max(65536.0, block_cells / 128.0));
    int64_t cand_want = std::max<int64_t>(std::max<int64_t>(1 << 20, capacity_hint), (int64_t)(block_cells / 4096.0));
    if (!forced && !ts.tiles) cand_want = std::min<int64_t>(cand_want, (int64_t)limit);
    if (hold_all && !forced) {
        if (ts.tiles) cand_want = (int64_t)std::min(268435456.0, std::max(1048576.0, tiles_to_do * 8.0 * (double)c->opt.tile_dense_thr));
        else cand_want = (int64_t)limit;
    }
    rc = ensure_buf(c, &c->pw_cand, &c->pw_cand_bytes, (size_t)cand_want * sizeof(int2));
    if (rc) return rc;
    a.coarse = (const int8_t*)c->pw_coarse;
    a.coarse_fm = nullptr;
    if (c->opt.fragment_major && mvs::filter_streams(a, c->opt)) {
        bool made = false;
        rc = prepare_coarse_fm(c, s, &made);
        if (rc) return rc;
        if (made) a.coarse_fm = (const int8_t*)c->pw_coarse_fm;
    }
    a.fmeta = (const float4*)c->pw_fmeta;
    a.cand_counter = c->d_counter + 2;
    a.cand_limit = limit;
    a.cand_stop = reinterpret_cast<unsigned int*>(c->d_counter + 32);
    a.recheck_queue = c->d_counter + 128;
    a.recheck_mode = c->opt.recheck_mode;
    const int64_t n_regions = mvs::filter_region_count(a, c->opt);
    if (n_regions > 0) {
        rc = ensure_buf(c, &c->pw_chdr, &c->pw_chdr_bytes, (size_t)n_regions * 4);
        if (rc) return rc;
        rc = ensure_buf(c, &c->pw_cent, &c->pw_cent_bytes, (size_t)n_regions * mvs::kCandRegion * sizeof(int2));
        if (rc) return rc;
        a.cand_hdr = (unsigned int*)c->pw_chdr;
        a.cand_ent = (int2*)c->pw_cent;
    }
    const size_t n_tiles = (size_t)ts.n_tr * (size_t)ts.n_tc;
    if (ts.tiles) {
        if (!ts.ext_flags) {
            rc = ensure_buf(c, &c->pw_tflag, &c->pw_tflag_bytes, n_tiles * 4);
            if (rc) return rc;
        }
        rc = ensure_buf(c, &c->pw_trow, &c->pw_trow_bytes, (size_t)ts.n_tr * 4);
        if (rc) return rc;
        a.tile_flag = ts.ext_flags ? ts.ext_flags : (unsigned int*)c->pw_tflag;
        a.tile_flag_ld = ts.n_tc;
        a.tile_dense_thr = (unsigned)c->opt.tile_dense_thr;
        a.tile_flag_count = reinterpret_cast<unsigned int*>(c->d_counter + 8);
        a.tile_flag_limit = forced ? 0xffffffffu : (unsigned)std::min(4.0e9, std::max(64.0, 0.7 * tiles_to_do));
    }
    std::vector<int> row_count((size_t)(ts.tiles ? ts.n_tr : 0));
    unsigned long long back[33];
    for (int attempt = 0;; ++attempt) {
        a.cand = (int2*)c->pw_cand;
        a.cand_capacity = c->pw_cand_bytes / sizeof(int2);
        // cell count, (debug), candidate count, ..., pruned count [6], flagged tiles [8] ... stop flag [32]; NOT words 3
        // and 4 (the streamed output's "wide q" and "q beyond a byte" flags: a block's flag must survive the next block's
        // filter pass, which is queued before the block's rows are read)
        HIP_TRY(hipMemsetAsync(c->d_counter + 1, 0, 16, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_counter + 5, 0, 224, c->stream));
        rc = set_cell_count(c, start);
        if (rc) return rc;
        HIP_TRY(hipMemsetAsync(a.recheck_queue, 0, 512, c->stream));
        if (n_regions > 0) HIP_TRY(hipMemsetAsync(a.cand_hdr, 0, (size_t)n_regions * 4, c->stream));
        if (ts.tiles) HIP_TRY(hipMemsetAsync(a.tile_flag, 0, n_tiles * 4, c->stream));
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
        rc = mvs::launch_filter(c->stream, a, c->opt);
        if (rc) return fail(rc, "filter launch rejected");
        rc = check_kernel("k_pairwise_mfma(filter)");
        if (rc) return rc;
        if (c->timing) HIP_TRY(hipEventRecord(c->ev[5], c->stream));   // closes the filter's interval, opens the re-check's
        if (n_regions > 0) {   // the waves' own candidate regions -> the list (counted with the re-check)
            mvs::launch_cand_gather(c->stream, a, n_regions);
            rc = check_kernel("k_cand_gather");
            if (rc) return rc;
        }
        if (ts.tiles) {
            mvs::launch_tile_count(c->stream, a.tile_flag, ts.n_tr, ts.n_tc, (int*)c->pw_trow);
            rc = check_kernel("k_tile_count");
            if (rc) return rc;
        }
        // one host synchronisation between the stages: the later launches are sized from these counts
        rc = read_back(c, c->stream, {{back, c->d_counter, sizeof(back)},
                                      {row_count.data(), c->pw_trow, ts.tiles ? (size_t)ts.n_tr * 4 : 0}});
        if (rc) return rc;
        ts.n_cand = back[2];
        c->last_candidates = ts.n_cand;
        const bool stopped = (back[32] & 0xffffffffULL) != 0;
        if (stopped || ts.n_cand > limit) {   // not paying: exact kernel now and for this set's later blocks
            c->filter_off_id = s->id;
            c->filter_off_coeff = keep_coeff;
            c->last_candidates = 0;
            c->last_flagged_tiles = (long long)(back[8] & 0xffffffffULL);
            return kNeedExact;
        }
        if (ts.n_cand <= a.cand_capacity) break;
        if (attempt >= 2) return fail(MVS_E_HIP, "internal: the candidate list keeps outgrowing its buffer");
        rc = ensure_buf(c, &c->pw_cand, &c->pw_cand_bytes, (size_t)ts.n_cand * sizeof(int2));
        if (rc) return rc;
    }
    ts.n_flagged = 0;
    ts.row_first.assign((size_t)ts.n_tr + 1, 0);
    for (int t = 0; t < (ts.tiles ? ts.n_tr : 0); ++t) {
        ts.row_first[(size_t)t + 1] = ts.row_first[(size_t)t] + row_count[(size_t)t];
    }
    if (ts.tiles) ts.n_flagged = ts.row_first[(size_t)ts.n_tr];
    c->last_flagged_tiles = ts.n_flagged;
    if (ts.n_flagged > 0) {
        rc = ensure_buf(c, &c->pw_tlist, &c->pw_tlist_bytes, ((size_t)ts.n_flagged + 1) * 4);
        if (rc) return rc;
        mvs::launch_tile_list(c->stream, a.tile_flag, ts.n_tr, ts.n_tc, (const int*)c->pw_trow, (int*)c->pw_tlist);
        rc = check_kernel("k_tile_list");
        if (rc) return rc;
        ts.d_list = (const int*)c->pw_tlist + 1;
        if (ts.n_cand > 0) {   // pairs that other waves of a flagged tile listed: those cells come from the exact kernel
            rc = ensure_buf(c, &c->pw_cand2, &c->pw_cand2_bytes, (size_t)ts.n_cand * sizeof(int2));
            if (rc) return rc;
            mvs::launch_cand_prune(c->stream, a, ts.n_cand, (int2*)c->pw_cand2, c->d_counter + 6);
            rc = check_kernel("k_cand_prune");
            if (rc) return rc;
            a.cand = (int2*)c->pw_cand2;
            a.cand_capacity = c->pw_cand2_bytes / sizeof(int2);
            a.cand_counter = c->d_counter + 6;
        }
        // the exact kernel's integer pre-test constants
        rc = ensure_buf(c, &c->pw_thr, &c->pw_thr_bytes, (size_t)s->n_alloc * 4);
        if (rc) return rc;
        mvs::launch_cand_thr(c->stream, d_n2, s->n, s->n_alloc, s->d, keep_coeff, (int32_t*)c->pw_thr);
        rc = check_kernel("k_cand_thr");
        if (rc) return rc;
        a.cand_thr = (const int32_t*)c->pw_thr;
    }
    ts.a = a;
    return MVS_OK;
}

// Stage 2: exact re-check of the listed candidates; kept cells go where ts.a's outputs point (cells / packed / dense)
int two_stage_recheck(mvs_ctx* c, TwoStage& ts) {
    int rc = mvs::launch_exact_pairs(c->stream, ts.a, c->opt);
    if (rc) return fail(rc, "exact re-check launch rejected");
    rc = check_kernel("k_exact_pairs");
    if (rc) return rc;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[7], c->stream));
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        c->ev_valid[1] = c->ev_valid[2] = c->ev_valid[3] = true;
        c->ev_valid[4] = false;
    }
    return MVS_OK;
}

// Stage 3: the exact kernel on flagged tiles [first, first + count) of the row-major list.  `timed`: this launch closes
// the comparison's timing interval (ev[6] .. ev[3]).
int two_stage_tiles(mvs_ctx* c, TwoStage& ts, int first, int count, bool timed) {
    if (count <= 0 && !timed) return MVS_OK;
    if (c->timing && timed) HIP_TRY(hipEventRecord(c->ev[6], c->stream));
    if (count > 0) {
        int rc = mvs::launch_exact_tiles(c->stream, ts.a, ts.d_list + first, count, c->opt);
        if (rc) return fail(rc, "exact tile launch rejected");
        rc = check_kernel("k_pairwise_pp(tiles)");
        if (rc) return rc;
    }
    if (c->timing && timed) {
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        c->ev_valid[4] = true;
    }
    return MVS_OK;
}

// may the two-stage comparison run on this block?  (see the comments at the call sites' old home, pairwise_launch)
bool two_stage_applies(mvs_ctx* c, const mvs_sketch_set* s, int64_t rb, int64_t re, int64_t cb, int64_t ce, double keep_coeff,
                       bool symmetric = true) {
    const int filter_mode = c->opt.pairwise_filter;
    const double block_cells = (double)(re - rb) * (double)(ce - cb);
    // A few rows against everything (a search with a handful of queries; one of very many shards) on a set whose coarse
    // plane does not exist yet: building the plane reads all the limb planes once, which is all the exact kernel needs for
    // such a block -- so the FIRST block of fewer than 1024 rows on a set goes to the exact kernel, and only when a second
    // one follows on the same set (a caller that keeps the set for many such blocks: pairwise_comp_optimized --shard_idx -1
    // with small shards, repeated searches) is the plane built.  Up to 16 rows the exact path is a streaming kernel that
    // runs at HBM speed (k_pairwise_skinny): nothing to filter for.
    const bool coarse_cached = c->coarse_id == s->id && c->coarse_gen == s->gen && c->coarse_mode == c->opt.coarse_radix;
    const bool few_rows = re - rb < 1024;
    const bool few_rows_again = c->few_rows_id == s->id && c->few_rows_gen == s->gen;
    // ... unless the coarse plane is there already: the streaming filter then reads half the bytes the streaming exact kernel
    // does (one coarse plane against two limb planes) and has the matrix cores for the products (16 rows x 10^6 columns:
    // 1.44 ms exact, see LABNOTES.md section 7)
    mvs::PairwiseArgs probe{};
    probe.limbs = s->limbs;
    probe.d_pad = s->d_pad;
    probe.row_begin = rb;
    probe.row_end = re;
    probe.col_begin = cb;
    probe.col_end = ce;
    probe.symmetric = (symmetric && c->opt.pairwise_symmetric) ? 1 : 0;
    const bool streams = mvs::filter_streams_rows(probe, c->opt);
    const bool two_stage = filter_mode != 0 && s->limbs == 2 && s->d_pad <= 32768 &&
                           (filter_mode == 2 ||   // forced: also on small blocks and on sets it was found not to pay for
                            ((block_cells >= 4194304.0 || streams) && (re - rb > 16 || (streams && (coarse_cached || few_rows_again))) &&
                             (coarse_cached || !few_rows || few_rows_again) &&
                             !(c->filter_off_id == s->id && c->filter_off_coeff == keep_coeff)));
    if (!two_stage && few_rows && (re - rb > 16 || streams) && (block_cells >= 4194304.0 || streams)) {
        c->few_rows_id = s->id;
        c->few_rows_gen = s->gen;
    }
    return two_stage;
}

int pairwise_launch(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, int keep_mode, int64_t rb, int64_t re,
                    int64_t cb, int64_t ce, bool symmetric, bool mirror_all, mvs_cell* raw, int64_t capacity,
                    unsigned long long start, unsigned long long* count, double keep_coeff = 0.05,
                    const PackedOut* po = nullptr, const DenseOut* dn = nullptr) {
    // *count: the cell count if this call already had to synchronise for it, ~0 otherwise (read d_counter[0])
    *count = ~0ULL;
    mvs::PairwiseArgs a{};
    fill_args(c, s, d_n2, keep_mode, rb, re, cb, ce, symmetric, mirror_all, keep_coeff, a);
    if (dn) {
        a.dense = dn->matrix;
        a.dense_row0 = dn->row0;
        a.dense_ld = dn->ld;
        a.dense_flag = dn->flag;
        a.sym_begin = dn->sym_begin;
        a.sym_end = dn->sym_end;
    }
    a.cells = raw;
    a.capacity = (unsigned long long)capacity;
    if (po) {
        a.cells = nullptr;
        a.packed = (unsigned long long*)*po->buf;
        a.capacity = *po->bytes / 8;
        a.pack_row0 = po->row0;
        a.pack_shift = po->shift;
    }
#ifdef MVS_ABLATIONS
    // per-workgroup time stamps of k_pairwise_pp (profiling only): one buffer for the process, dumped after the call
    static unsigned long long* g_stamps = nullptr;
    const size_t stamp_bytes = (size_t)mvs::kStampSlots * 64;
    if (c->opt.pairwise_debug & 8) {
        if (!g_stamps) HIP_TRY(hipMalloc((void**)&g_stamps, stamp_bytes));
        HIP_TRY(hipMemsetAsync(g_stamps, 0, stamp_bytes, c->stream));
        a.stamps = g_stamps;
    }
    struct StampDump {
        unsigned long long* p; size_t bytes; hipStream_t st;
        ~StampDump() {
            if (!p) return;
            (void)hipStreamSynchronize(st);
            std::vector<char> h(bytes);
            (void)hipMemcpy(h.data(), p, bytes, hipMemcpyDeviceToHost);
            FILE* f = fopen("/tmp/mvs_stamps.bin", "wb");
            if (f) { fwrite(h.data(), 1, bytes, f); fclose(f); }
        }
    } stamp_dump{a.stamps, stamp_bytes, c->stream};
#endif
    int rc = MVS_OK;
    if (!dn && two_stage_applies(c, s, rb, re, cb, ce, keep_coeff, symmetric)) {
        TwoStage ts;
        rc = two_stage_filter(c, s, d_n2, keep_coeff, capacity, po != nullptr, start, a, ts);
        if (rc == MVS_OK) {
            if (po) {
                // the output is sized between the stages: a kept cell is a candidate or the mirror image of one, or a cell
                // of a flagged tile or of its mirror image
                rc = ensure_buf(c, po->buf, po->bytes,
                                (size_t)(start + 2 * ts.n_cand + (unsigned long long)ts.n_flagged * 131072ULL + 64) * 8);
                if (rc) return rc;
                ts.a.packed = (unsigned long long*)*po->buf;
                ts.a.capacity = *po->bytes / 8;
            }
            rc = two_stage_recheck(c, ts);
            if (rc) return rc;
            return two_stage_tiles(c, ts, 0, ts.n_flagged, true);
        }
        if (rc != kNeedExact) return rc;
    }
    c->last_candidates = 0;
    if (po && po->two_stage_only) return kNeedExact;
    rc = ensure_buf(c, &c->pw_thr, &c->pw_thr_bytes, (size_t)s->n_alloc * 4);
    if (rc) return rc;
    mvs::launch_cand_thr(c->stream, d_n2, s->n, s->n_alloc, s->d, keep_coeff, (int32_t*)c->pw_thr);
    rc = check_kernel("k_cand_thr");
    if (rc) return rc;
    a.cand_thr = (const int32_t*)c->pw_thr;
    rc = set_cell_count(c, start);
    if (rc) return rc;
    rc = attach_planes_fm(c, s, a, mvs::exact_reads_fm(a, c->opt));
    if (rc) return rc;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    rc = mvs::launch_pairwise(c->stream, a, 0, 0, c->opt);
    if (rc) return fail(rc, "pairwise launch rejected");
    rc = check_kernel("k_pairwise");
    if (rc) return rc;
    if (c->timing) {
        HIP_TRY(hipEventRecord(c->ev[3], c->stream));
        c->ev_valid[1] = true;
        c->ev_valid[2] = c->ev_valid[3] = c->ev_valid[4] = false;   // no filter / re-check in this comparison
    }
    return MVS_OK;
}

int sort_on_device(mvs_ctx* c, mvs_cell* in, int64_t n, mvs_cell* out) {
    size_t need = 0;
    int rc = mvs::sort_cells(c->stream, in, out, n, nullptr, 0, &need, c->opt);
    if (rc) return fail(rc, "sort sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::sort_cells(c->stream, in, out, n, c->pw_sort, c->pw_sort_bytes, nullptr, c->opt);
    if (rc) return fail(rc, "sort failed");
    return MVS_OK;
}

}  // namespace

int mvs_pairwise_rows(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                      int64_t row_begin, int64_t row_end, mvs_cell* cells, int64_t capacity, int mem_cells,
                      int64_t* n_cells) {
    if (!c || !s || !n_cells) return fail(MVS_E_INVALID, "NULL argument");
    const Range range(c, "mvs_pairwise_rows");
    *n_cells = 0;
    if (!mem_ok(mem_norms) || !mem_ok(mem_cells) || capacity < 0 ||
        (keep_mode != MVS_KEEP_INT32 && keep_mode != MVS_KEEP_INT16))
        return fail(MVS_E_INVALID, "bad argument");
    if (row_begin < 0 || row_end > s->n || row_begin > row_end)
        return fail(MVS_E_INVALID, "row range [%lld,%lld) outside [0,%lld)", (long long)row_begin,
                    (long long)row_end, (long long)s->n);
    if (row_begin == row_end || s->n == 0) return MVS_OK;
    if (!norms_sq) return fail(MVS_E_INVALID, "norms_sq is NULL");
    if (capacity > 0 && !cells) return fail(MVS_E_INVALID, "cells is NULL");
    HIP_TRY(hipSetDevice(c->device));

    DevBuf dn;
    const double* d_n2 = norms_sq;
    if (mem_norms == MVS_MEM_HOST) {
        HIP_TRY(dn.alloc((size_t)s->n * 8));
        HIP_TRY(hipMemcpyAsync(dn.p, norms_sq, (size_t)s->n * 8, hipMemcpyHostToDevice, c->stream));
        d_n2 = (const double*)dn.p;
    }
    mvs_cell* d_cells = cells;
    if (mem_cells == MVS_MEM_HOST) {   // sorted cells are staged in a grow-only device buffer of the context
        int rc0 = ensure_buf(c, &c->pw_out, &c->pw_out_bytes, (size_t)std::max<int64_t>(capacity, 1) * sizeof(mvs_cell));
        if (rc0) return rc0;
        d_cells = (mvs_cell*)c->pw_out;
    }
    // kept cells are appended (unordered) to a staging buffer and merge-sorted into the caller's
    int rc = ensure_buf(c, &c->pw_tmp, &c->pw_tmp_bytes, (size_t)std::max<int64_t>(capacity, 1) * sizeof(mvs_cell));
    if (rc) return rc;
    // Very large shards go through in row chunks of at most 2^40 cells (chunk borders on multiples of 256 rows so
    // that every chunk can use the symmetric schedule): that bounds the candidate list of the two-stage
    // comparison.  Option pairwise_block_cells overrides the bound (tests).
    const double max_cells = c->opt.pairwise_block_cells;
    int64_t chunk_rows = (int64_t)(max_cells / (double)s->n);
    chunk_rows = std::max<int64_t>(256, chunk_rows / 256 * 256);
    unsigned long long count = 0;
    for (int64_t rb = row_begin; rb < row_end;) {
        const int64_t re = std::min(row_end, (rb / 256) * 256 + chunk_rows);
        unsigned long long got = 0;
        rc = pairwise_launch(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, (mvs_cell*)c->pw_tmp, capacity, count,
                             &got);
        if (rc) return rc;
        if (got == ~0ULL) {
            {
                const int rb_rc = read_back(c, c->stream, {{&got, c->d_counter, 8}});
                if (rb_rc) return rb_rc;
            }
        }
        count = got;
        rb = re;
    }
    *n_cells = (int64_t)count;
    if ((int64_t)count > capacity)
        return fail(MVS_E_CAPACITY, "%llu cells kept but capacity is %lld", count, (long long)capacity);
    if (count == 0) return MVS_OK;
    // order by (row, col): the per-row ascending-column order of the reference's result list
    rc = sort_on_device(c, (mvs_cell*)c->pw_tmp, (int64_t)count, d_cells);
    if (rc) return rc;
    if (mem_cells == MVS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(cells, d_cells, (size_t)count * sizeof(mvs_cell), hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    // device output: the sort is queued on the context's stream; *n_cells is already final
    return MVS_OK;
}

// -------------------------------------------------------------------------------------------------
// streamed output
// -------------------------------------------------------------------------------------------------
namespace {

int bits_for(int64_t max_value) {          // bits that hold 0 .. max_value
    int b = 1;
    while (b < 63 && (max_value >> b) != 0) ++b;
    return b;
}

// Hand-over between the thread that drives the GPU and the one that runs the caller's callback: two pinned buffers,
// a queue of filled ones.  The callback therefore runs beside the next block's kernels and downloads.
struct StreamOut {
    struct Item {
        int slot;
        int64_t row_begin, row_end, n_cells;
        std::vector<int64_t> row_ptr;      // rebased to the block's first cell
        bool wide;
        // encoded pieces: the directory of the piece's non-empty rows, the records' byte count
        std::vector<uint32_t> rows, first_col, jac_bytes;
        std::vector<uint64_t> offset;
        int64_t n_bytes = 0;
    };
    mvs_ctx* c;
    mvs_row_block_cb cb = nullptr;
    mvs_encoded_rows_cb ecb = nullptr;     // set instead of cb by mvs_pairwise_stream_encoded
    void* user;
    std::mutex mu;
    std::condition_variable cv;
    std::deque<Item> queue;
    bool slot_busy[2] = {false, false};
    bool closing = false;
    int cb_status = 0;                     // first non-zero return of the callback
    std::string error;
    std::thread worker;
    // The feeder: hands finished row blocks to the link piece by piece (it blocks on the two pinned buffers), so that the
    // thread that drives the device never waits for the link -- it runs at most two blocks ahead (the device-side arrays
    // of a block are double-buffered: set k & 1).
    std::thread feeder;
    std::deque<std::function<int()>> feed_queue;
    bool feed_closing = false;
    int64_t fed_blocks = 0;                // blocks whose pieces have all been queued on the download stream
    int feed_rc = 0;

    void feed_run() {
        (void)hipSetDevice(c->device);
        for (;;) {
            std::function<int()> task;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return feed_closing || !feed_queue.empty(); });
                if (feed_queue.empty()) return;
                task = std::move(feed_queue.front());
                feed_queue.pop_front();
            }
            const int r = task();
            {
                std::lock_guard<std::mutex> lk(mu);
                if (r != 0 && feed_rc == 0) {
                    feed_rc = r;
                    if (error.empty()) error = std::string("feeding the link failed: ") + mvs_last_error();
                }
                ++fed_blocks;
            }
            cv.notify_all();
        }
    }
    void enqueue_feed(std::function<int()> task) {
        {
            std::lock_guard<std::mutex> lk(mu);
            feed_queue.push_back(std::move(task));
        }
        cv.notify_all();
    }
    void wait_fed(int64_t blocks) {         // until that many blocks have been handed to the download stream
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return fed_blocks >= blocks; });
    }
    void close_feeder() {
        {
            std::lock_guard<std::mutex> lk(mu);
            feed_closing = true;
        }
        cv.notify_all();
        if (feeder.joinable()) feeder.join();
    }

    void run() {
        (void)hipSetDevice(c->device);
        for (;;) {
            Item it;
            {
                std::unique_lock<std::mutex> lk(mu);
                cv.wait(lk, [&] { return closing || !queue.empty(); });
                if (queue.empty()) return;
                it = std::move(queue.front());
                queue.pop_front();
            }
            int status = 0;
            const hipError_t e = hipEventSynchronize(c->dl_done[it.slot]);
            if (e != hipSuccess) {
                std::lock_guard<std::mutex> lk(mu);
                if (error.empty()) error = std::string("download failed: ") + hipGetErrorString(e);
            } else {
                bool skip;
                {
                    std::lock_guard<std::mutex> lk(mu);
                    skip = cb_status != 0 || !error.empty();
                }
                if (!skip && ecb) {
                    mvs_encoded_rows b{};
                    b.row_begin = it.row_begin;
                    b.row_end = it.row_end;
                    b.n_cells = it.n_cells;
                    b.n_rows = (int64_t)it.rows.size();
                    b.rows = it.rows.data();
                    b.first_col = it.first_col.data();
                    b.offset = it.offset.data();
                    b.jac_bytes = it.jac_bytes.data();
                    b.bytes = static_cast<const uint8_t*>(c->dl_pinned[it.slot]);
                    b.n_bytes = it.n_bytes;
                    try {
                        status = ecb(user, &b);
                    } catch (...) {
                        status = -1;
                    }
                } else if (!skip) {
                    mvs_row_block b{};
                    b.row_begin = it.row_begin;
                    b.row_end = it.row_end;
                    b.n_cells = it.n_cells;
                    b.row_ptr = it.row_ptr.data();
                    const char* base = static_cast<const char*>(c->dl_pinned[it.slot]);
                    b.col = reinterpret_cast<const int32_t*>(base);
                    const char* qbase = base + (size_t)it.n_cells * 4;
                    b.q = it.wide ? nullptr : reinterpret_cast<const uint8_t*>(qbase);
                    b.q16 = it.wide ? reinterpret_cast<const uint16_t*>(qbase) : nullptr;
                    try {
                        status = cb(user, &b);
                    } catch (...) {            // a C++ callback that throws: no exception crosses the C boundary
                        status = -1;
                    }
                }
            }
            {
                std::lock_guard<std::mutex> lk(mu);
                if (status != 0 && cb_status == 0) cb_status = status;
                slot_busy[it.slot] = false;
            }
            cv.notify_all();
        }
    }
    int acquire_slot() {                    // blocks until one of the two pinned buffers is free
        std::unique_lock<std::mutex> lk(mu);
        cv.wait(lk, [&] { return !slot_busy[0] || !slot_busy[1]; });
        const int sl = slot_busy[0] ? 1 : 0;
        slot_busy[sl] = true;
        return sl;
    }
    void release_slot(int sl) {
        {
            std::lock_guard<std::mutex> lk(mu);
            slot_busy[sl] = false;
        }
        cv.notify_all();
    }
    void push(Item&& it) {
        {
            std::lock_guard<std::mutex> lk(mu);
            queue.push_back(std::move(it));
        }
        cv.notify_all();
    }
    bool failed() {
        std::lock_guard<std::mutex> lk(mu);
        return cb_status != 0 || !error.empty();
    }
    void close() {
        {
            std::lock_guard<std::mutex> lk(mu);
            closing = true;
        }
        cv.notify_all();
        if (worker.joinable()) worker.join();
    }
    ~StreamOut() {
        close_feeder();
        close();
    }
};

int ensure_download_side(mvs_ctx* c) {
    if (!c->dl_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&c->dl_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(hipEventCreateWithFlags(&c->dl_done[i], hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&c->dl_block[i], hipEventDisableTiming));
        }
        for (int i = 0; i < 2; ++i) HIP_TRY(hipEventCreateWithFlags(&c->dl_ready[i], hipEventDisableTiming));
        HIP_TRY(hipStreamCreateWithFlags(&c->post_stream, hipStreamNonBlocking));
        HIP_TRY(hipEventCreateWithFlags(&c->cmp_done, hipEventDisableTiming));
    }
    return MVS_OK;
}

// pinned buffer `slot` holds at least `bytes`; called by the producer while it owns the slot (nobody reads it)
int ensure_pinned_slot(mvs_ctx* c, int slot, size_t bytes) {
    if (c->dl_bytes[slot] >= bytes) return MVS_OK;
    if (c->dl_pinned[slot]) HIP_TRY(hipHostFree(c->dl_pinned[slot]));
    c->dl_pinned[slot] = nullptr;
    c->dl_bytes[slot] = 0;
    HIP_TRY(hipHostMalloc(&c->dl_pinned[slot], bytes, hipHostMallocDefault));
    c->dl_bytes[slot] = bytes;
    return MVS_OK;
}

// A row block on its way out: its CSR arrays sit in set `set` of the context (device), row_ptr is on the host.
struct BlockCsr {
    int64_t rb = 0, re = 0, n = 0;
    std::vector<int64_t> row_ptr;      // re - rb + 1 entries
    bool wide = false;                 // q is 16 bits wide in this block
    int set = 0;
    // rows encoded on the device: byte offset of every row's record (rows + 1 entries), directory values per row
    bool sizes_ready = false;          // the encoder's per-row sizes (en_size / en_jac / en_first / en_par) are on the device already
    bool encoded = false;
    std::vector<uint64_t> enc_off;
    std::vector<uint32_t> enc_jac, enc_first;
};

// The CSR arrays of `b` (set b.set) -> the rows' shard records in c->st_enc[b.set], directory on the host; on stream `ps`
// (the context's stream, or the side stream on which a dense block is post-processed beside the next comparison)
int encode_block(mvs_ctx* c, BlockCsr& b, hipStream_t ps) {
    const int64_t rows = b.re - b.rb;
    b.encoded = true;
    b.enc_off.assign((size_t)rows + 1, 0);
    b.enc_jac.assign((size_t)rows, 0);
    b.enc_first.assign((size_t)rows, 0);
    if (b.n == 0 || rows == 0) {
        HIP_TRY(hipEventRecord(c->dl_ready[b.set], ps));
        return MVS_OK;
    }
    int rc = ensure_buf(c, &c->en_size, &c->en_size_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_off, &c->en_off_bytes, (size_t)(rows + 1) * 8);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_jac, &c->en_jac_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_first, &c->en_first_bytes, (size_t)rows * 4);
    if (rc == MVS_OK) rc = ensure_buf(c, &c->en_par, &c->en_par_bytes, (size_t)rows * sizeof(mvs::EncRow));
    if (rc) return rc;
    const int qb = b.wide ? 2 : 1;
    HIP_TRY(hipMemsetAsync((char*)c->en_size + (size_t)rows * 8, 0, 8, ps));
    if (!b.sizes_ready) {                  // (a dense block's fill pass has computed them already)
        mvs::launch_encode_sizes(ps, (const long long*)c->st_rowptr, (const int32_t*)c->st_col[b.set], c->st_q[b.set], qb, rows,
                                 (unsigned long long*)c->en_size, (unsigned int*)c->en_jac, (unsigned int*)c->en_first,
                                 (mvs::EncRow*)c->en_par);
        rc = check_kernel("k_enc_size");
        if (rc) return rc;
    }
    size_t need = 0;
    rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, nullptr, 0, &need);
    if (rc) return fail(rc, "scan sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::encode_offsets(ps, (unsigned long long*)c->en_size, (unsigned long long*)c->en_off, rows, c->pw_sort,
                             c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the record sizes failed");
    rc = read_back(c, ps, {{b.enc_off.data(), c->en_off, (size_t)(rows + 1) * 8},
                           {b.enc_jac.data(), c->en_jac, (size_t)rows * 4},
                           {b.enc_first.data(), c->en_first, (size_t)rows * 4}});
    if (rc) return rc;
    const size_t total = (size_t)b.enc_off[(size_t)rows];
    rc = ensure_buf(c, &c->st_enc[b.set], &c->st_enc_bytes[b.set], std::max<size_t>(total, 8));
    if (rc) return rc;
    HIP_TRY(hipMemsetAsync(c->st_enc[b.set], 0, total, ps));       // the unary parts are OR-ed into zeroed words
    mvs::launch_encode_fill(ps, (const long long*)c->st_rowptr, (const int32_t*)c->st_col[b.set], c->st_q[b.set], qb, rows,
                            (const unsigned long long*)c->en_off, (const mvs::EncRow*)c->en_par, (unsigned char*)c->st_enc[b.set],
                            c->opt.encode_stage_words);
    rc = check_kernel("k_enc_fill");
    if (rc) return rc;
    HIP_TRY(hipEventRecord(c->dl_ready[b.set], ps));
    return MVS_OK;
}

// before the CSR arrays of set `set` are rewritten: the downloads of the block that used them last (two blocks ago) are through
int claim_csr_set(mvs_ctx* c, int set, int64_t block_index, int64_t n, bool wide, hipStream_t ps) {
    if (block_index >= 2) HIP_TRY(hipStreamWaitEvent(ps, c->dl_block[set], 0));
    int rc = ensure_buf(c, &c->st_col[set], &c->st_col_bytes[set], (size_t)std::max<int64_t>(n, 1) * 4);
    if (rc) return rc;
    return ensure_buf(c, &c->st_q[set], &c->st_q_bytes[set], (size_t)std::max<int64_t>(n, 1) * (wide ? 2 : 1));
}

// n packed cells of rows [rb, re) sit in c->st_raw: radix sort on the (row, col) bits, then row_ptr / col / q
int csr_from_packed(mvs_ctx* c, int64_t rb, int64_t re, int64_t n, int shift, int col_bits, int64_t block_index, BlockCsr& out) {
    const int64_t rows = re - rb;
    const int row_bits = bits_for(std::max<int64_t>(rows - 1, 1));
    out.rb = rb;
    out.re = re;
    out.n = n;
    out.wide = false;
    out.set = (int)(block_index & 1);
    out.row_ptr.assign((size_t)rows + 1, 0);
    if (n == 0) {
        HIP_TRY(hipEventRecord(c->dl_ready[out.set], c->stream));
        return MVS_OK;
    }
    int rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    rc = ensure_buf(c, &c->st_sorted, &c->st_sorted_bytes, (size_t)n * 8);
    if (rc) return rc;
    size_t need = 0;
    rc = mvs::sort_packed(c->stream, (unsigned long long*)c->st_raw, (unsigned long long*)c->st_sorted, n, 16, shift + row_bits,
                          nullptr, 0, &need);
    if (rc) return fail(rc, "sort sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::sort_packed(c->stream, (unsigned long long*)c->st_raw, (unsigned long long*)c->st_sorted, n, 16, shift + row_bits,
                          c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "sort of the kept cells failed");
    rc = claim_csr_set(c, out.set, block_index, n, false, c->stream);
    if (rc) return rc;
    unsigned int* d_wide = reinterpret_cast<unsigned int*>(c->d_counter + 3);
    HIP_TRY(hipMemsetAsync(d_wide, 0, 4, c->stream));
    const unsigned long long col_mask = (1ULL << col_bits) - 1ULL;
    mvs::launch_packed_csr(c->stream, (const unsigned long long*)c->st_sorted, n, shift, rows, col_mask, (long long*)c->st_rowptr,
                           (int32_t*)c->st_col[out.set], (uint8_t*)c->st_q[out.set], nullptr, d_wide);
    rc = check_kernel("k_packed_csr");
    if (rc) return rc;
    unsigned int h_wide = 0;
    HIP_TRY(hipMemcpyAsync(out.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipMemcpyAsync(&h_wide, d_wide, 4, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (h_wide) {                      // some q needs 16 bits (norms that do not belong to the vectors): redo the q array
        out.wide = true;
        rc = ensure_buf(c, &c->st_q[out.set], &c->st_q_bytes[out.set], (size_t)n * 2);
        if (rc) return rc;
        mvs::launch_packed_csr(c->stream, (const unsigned long long*)c->st_sorted, n, shift, rows, col_mask, nullptr,
                               (int32_t*)c->st_col[out.set], nullptr, (uint16_t*)c->st_q[out.set], nullptr);
        rc = check_kernel("k_packed_csr(16-bit q)");
        if (rc) return rc;
    }
    if (out.row_ptr[(size_t)rows] != n) return fail(MVS_E_HIP, "internal: row index of the sorted cells is inconsistent");
    HIP_TRY(hipEventRecord(c->dl_ready[out.set], c->stream));          // the downloads of this block wait for exactly this point
    return MVS_OK;
}

// rows [rb, re) of the dense byte matrix (first row dense_row0, leading dimension ld) are final: count, scan, fill.
// *odd: some kept cell of the launches so far has a q the byte cannot hold -- the caller redoes the block as a list.
int csr_from_dense(mvs_ctx* c, int64_t rb, int64_t re, int64_t n_cols, int64_t dense_row0, int64_t ld, int64_t block_index,
                   BlockCsr& out, bool* odd, hipStream_t ps, mvs::DenseActive active, bool want_sizes) {
    active.row_rel0 = rb - dense_row0;
    const int64_t rows = re - rb;
    out.rb = rb;
    out.re = re;
    out.wide = false;
    out.set = (int)(block_index & 1);
    out.row_ptr.assign((size_t)rows + 1, 0);
    int rc = ensure_buf(c, &c->st_rowptr, &c->st_rowptr_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    rc = ensure_buf(c, &c->st_counts, &c->st_counts_bytes, (size_t)(rows + 1) * 8);
    if (rc) return rc;
    // the active tiles of the block's tile rows, every row's first / last kept column
    int tr0 = 0, n_trows = 0, n_tc = 0;
    mvs::dense_tile_rows(active, rows, n_cols, &tr0, &n_trows, &n_tc);
    rc = ensure_buf(c, &c->st_tlist, &c->st_tlist_bytes, std::max<size_t>((size_t)n_trows * (size_t)n_tc * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_tlist_n, &c->st_tlist_n_bytes, std::max<size_t>((size_t)n_trows * 4, 4));
    if (rc == MVS_OK) rc = ensure_buf(c, &c->st_ends, &c->st_ends_bytes, std::max<size_t>((size_t)rows * sizeof(int2), 8));
    if (rc) return rc;
    const uint8_t* first = (const uint8_t*)c->st_dense + (size_t)(rb - dense_row0) * (size_t)ld;
    HIP_TRY(hipMemsetAsync((char*)c->st_counts + (size_t)rows * 8, 0, 8, ps));
    mvs::launch_dense_count(ps, first, ld, n_cols, rows, (long long*)c->st_counts, (int2*)c->st_ends, active, (int*)c->st_tlist,
                            (int*)c->st_tlist_n);
    rc = check_kernel("k_dense_count");
    if (rc) return rc;
    size_t need = 0;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, nullptr, 0, &need);
    if (rc) return fail(rc, "scan sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::dense_row_ptr(ps, (long long*)c->st_counts, (long long*)c->st_rowptr, rows, c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "scan of the row counts failed");
    unsigned int h_odd = 0;
    rc = read_back(c, ps, {{out.row_ptr.data(), c->st_rowptr, (size_t)(rows + 1) * 8}, {&h_odd, c->d_counter + 4, 4}});
    if (rc) return rc;
    *odd = h_odd != 0;
    if (*odd) return MVS_OK;
    out.n = out.row_ptr[(size_t)rows];
    rc = claim_csr_set(c, out.set, block_index, out.n, false, ps);
    if (rc) return rc;
    // rows that will be encoded on the device: the record sizes come out of the fill pass (k_enc_size would read the CSR
    // arrays this pass is writing once more)
    if (want_sizes && rows > 0) {
        rc = ensure_buf(c, &c->en_size, &c->en_size_bytes, (size_t)(rows + 1) * 8);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_jac, &c->en_jac_bytes, (size_t)rows * 4);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_first, &c->en_first_bytes, (size_t)rows * 4);
        if (rc == MVS_OK) rc = ensure_buf(c, &c->en_par, &c->en_par_bytes, (size_t)rows * sizeof(mvs::EncRow));
        if (rc) return rc;
    }
    const bool sizes = want_sizes && rows > 0 && out.n > 0;
    mvs::launch_dense_fill(ps, first, ld, n_cols, rows, (const long long*)c->st_rowptr, (int32_t*)c->st_col[out.set],
                           (uint8_t*)c->st_q[out.set], active, (const int*)c->st_tlist, (const int*)c->st_tlist_n,
                           (const int2*)c->st_ends, sizes ? (unsigned long long*)c->en_size : nullptr, (unsigned int*)c->en_jac,
                           (unsigned int*)c->en_first, (mvs::EncRow*)c->en_par);
    rc = check_kernel("k_dense_fill");
    if (rc) return rc;
    out.sizes_ready = sizes;
    HIP_TRY(hipEventRecord(c->dl_ready[out.set], ps));
    return MVS_OK;
}

// the block's CSR arrays out through the two pinned buffers, in pieces of whole rows; the host blocks here only on the
// pinned buffers (the device is free to run the next block's comparison meanwhile)
int feed_block(mvs_ctx* c, StreamOut& out, const BlockCsr& b, size_t piece_bytes) {
    const int64_t rows = b.re - b.rb, n = b.n;
    const std::vector<int64_t>& row_ptr = b.row_ptr;
    const bool wide = b.wide;
    int rc = MVS_OK;
    const size_t cell_bytes = wide ? 6 : 5;
    const int64_t piece_cells = std::max<int64_t>(1, (int64_t)(piece_bytes / cell_bytes));
    // a piece = as many whole rows as fit piece_bytes; one row alone may exceed that
    auto piece_end = [&](int64_t r0) {
        int64_t r1 = r0 + 1;
        const int64_t c0 = row_ptr[(size_t)r0];
        if (row_ptr[(size_t)r1] - c0 <= piece_cells) {
            const int64_t* end = std::upper_bound(row_ptr.data() + r1, row_ptr.data() + rows + 1, c0 + piece_cells);
            r1 = std::max<int64_t>(r1, (end - row_ptr.data()) - 1);
        }
        return r1;
    };
    // a pinned buffer is sized for the block's largest piece when the producer takes it (it is idle then)
    size_t need_bytes = std::min<size_t>(piece_bytes, std::max<size_t>((size_t)n * cell_bytes, 1u << 16));
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        need_bytes = std::max(need_bytes, (size_t)(row_ptr[(size_t)r1] - row_ptr[(size_t)r0]) * cell_bytes);
        r0 = r1;
    }
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        const int64_t c0 = row_ptr[(size_t)r0];
        const int64_t cells = row_ptr[(size_t)r1] - c0;
        if (out.failed()) return MVS_OK;                        // the caller reports the callback's status
        const int sl = out.acquire_slot();
        rc = ensure_pinned_slot(c, sl, need_bytes);
        if (rc) {
            out.release_slot(sl);
            return rc;
        }
        StreamOut::Item it;
        it.slot = sl;
        it.row_begin = b.rb + r0;
        it.row_end = b.rb + r1;
        it.n_cells = cells;
        it.wide = wide;
        it.row_ptr.resize((size_t)(r1 - r0) + 1);
        for (int64_t r = r0; r <= r1; ++r) it.row_ptr[(size_t)(r - r0)] = row_ptr[(size_t)r] - c0;
        hipError_t e = hipStreamWaitEvent(c->dl_stream, c->dl_ready[b.set], 0);
        char* dst = static_cast<char*>(c->dl_pinned[sl]);
        if (e == hipSuccess && cells > 0) {
            e = hipMemcpyAsync(dst, (const char*)c->st_col[b.set] + (size_t)c0 * 4, (size_t)cells * 4, hipMemcpyDeviceToHost,
                               c->dl_stream);
            if (e == hipSuccess)
                e = hipMemcpyAsync(dst + (size_t)cells * 4, (const char*)c->st_q[b.set] + (size_t)c0 * (wide ? 2 : 1),
                                   (size_t)cells * (wide ? 2 : 1), hipMemcpyDeviceToHost, c->dl_stream);
        }
        if (e == hipSuccess) e = hipEventRecord(c->dl_done[sl], c->dl_stream);
        if (e != hipSuccess) {
            out.release_slot(sl);
            return fail(MVS_E_HIP, "download of a row block: %s", hipGetErrorString(e));
        }
        out.push(std::move(it));
        ++c->st_pieces;
        c->st_bytes += (long long)((size_t)cells * cell_bytes);
        r0 = r1;
    }
    HIP_TRY(hipEventRecord(c->dl_block[b.set], c->dl_stream));
    return MVS_OK;
}

// the block's encoded records out through the pinned buffers, in pieces of whole rows of at most piece_bytes
int feed_encoded(mvs_ctx* c, StreamOut& out, const BlockCsr& b, size_t piece_bytes) {
    const int64_t rows = b.re - b.rb;
    const std::vector<uint64_t>& off = b.enc_off;
    auto piece_end = [&](int64_t r0) {
        int64_t r1 = r0 + 1;
        const uint64_t o0 = off[(size_t)r0];
        if (off[(size_t)r1] - o0 <= piece_bytes) {
            const uint64_t* end = std::upper_bound(off.data() + r1, off.data() + rows + 1, o0 + (uint64_t)piece_bytes);
            r1 = std::max<int64_t>(r1, (end - off.data()) - 1);
        }
        return r1;
    };
    size_t need_bytes = std::min<size_t>(piece_bytes, std::max<size_t>((size_t)off[(size_t)rows], 1u << 16));
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        need_bytes = std::max(need_bytes, (size_t)(off[(size_t)r1] - off[(size_t)r0]));
        r0 = r1;
    }
    int rc = MVS_OK;
    for (int64_t r0 = 0; r0 < rows;) {
        const int64_t r1 = piece_end(r0);
        const uint64_t o0 = off[(size_t)r0], bytes = off[(size_t)r1] - o0;
        if (out.failed()) return MVS_OK;
        const int sl = out.acquire_slot();
        rc = ensure_pinned_slot(c, sl, need_bytes);
        if (rc) {
            out.release_slot(sl);
            return rc;
        }
        StreamOut::Item it;
        it.slot = sl;
        it.row_begin = b.rb + r0;
        it.row_end = b.rb + r1;
        it.n_cells = b.row_ptr[(size_t)r1] - b.row_ptr[(size_t)r0];
        it.wide = b.wide;
        it.n_bytes = (int64_t)bytes;
        for (int64_t r = r0; r < r1; ++r)
            if (b.row_ptr[(size_t)r + 1] > b.row_ptr[(size_t)r]) {
                it.rows.push_back((uint32_t)(b.rb + r));
                it.first_col.push_back(b.enc_first[(size_t)r]);
                it.jac_bytes.push_back(b.enc_jac[(size_t)r]);
                it.offset.push_back(off[(size_t)r] - o0);
            }
        hipError_t e = hipStreamWaitEvent(c->dl_stream, c->dl_ready[b.set], 0);
        if (e == hipSuccess && bytes > 0)
            e = hipMemcpyAsync(c->dl_pinned[sl], (const char*)c->st_enc[b.set] + o0, (size_t)bytes, hipMemcpyDeviceToHost, c->dl_stream);
        if (e == hipSuccess) e = hipEventRecord(c->dl_done[sl], c->dl_stream);
        if (e != hipSuccess) {
            out.release_slot(sl);
            return fail(MVS_E_HIP, "download of encoded rows: %s", hipGetErrorString(e));
        }
        out.push(std::move(it));
        ++c->st_pieces;
        c->st_bytes += (long long)bytes;
        r0 = r1;
    }
    HIP_TRY(hipEventRecord(c->dl_block[b.set], c->dl_stream));
    return MVS_OK;
}

}  // namespace

namespace {
int pairwise_stream_impl(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                         int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb,
                         mvs_encoded_rows_cb ecb, void* user, int64_t* n_cells);
}

int mvs_pairwise_stream(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                        int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb, void* user,
                        int64_t* n_cells) {
    try {       // the host side keeps per-row directories in std::vector: no exception may cross the C boundary
        return pairwise_stream_impl(c, s, norms_sq, mem_norms, keep_mode, row_begin, row_end, device_budget_bytes, cb, nullptr, user,
                                    n_cells);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming the comparison result");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_pairwise_stream: %s", e.what());
    }
}

int mvs_pairwise_stream_encoded(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                                int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_encoded_rows_cb cb, void* user,
                                int64_t* n_cells) {
    try {
        return pairwise_stream_impl(c, s, norms_sq, mem_norms, keep_mode, row_begin, row_end, device_budget_bytes, nullptr, cb, user,
                                    n_cells);
    } catch (const std::bad_alloc&) {
        return fail(MVS_E_NOMEM, "out of host memory while streaming the comparison result");
    } catch (const std::exception& e) {
        return fail(MVS_E_HIP, "mvs_pairwise_stream_encoded: %s", e.what());
    }
}

namespace {
int pairwise_stream_impl(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int mem_norms, int keep_mode,
                         int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb,
                         mvs_encoded_rows_cb ecb, void* user, int64_t* n_cells) {
    if (!c || !s || (!cb && !ecb)) return fail(MVS_E_INVALID, "NULL argument");
    const Range range(c, "mvs_pairwise_stream");
    if (n_cells) *n_cells = 0;
    if (!mem_ok(mem_norms) || (keep_mode != MVS_KEEP_INT32 && keep_mode != MVS_KEEP_INT16)) return fail(MVS_E_INVALID, "bad argument");
    if (row_begin < 0 || row_end > s->n || row_begin > row_end)
        return fail(MVS_E_INVALID, "row range [%lld,%lld) outside [0,%lld)", (long long)row_begin, (long long)row_end, (long long)s->n);
    if (row_begin == row_end || s->n == 0) return MVS_OK;
    if (!norms_sq) return fail(MVS_E_INVALID, "norms_sq is NULL");
    HIP_TRY(hipSetDevice(c->device));
    DevBuf dn;
    const double* d_n2 = norms_sq;
    if (mem_norms == MVS_MEM_HOST) {
        HIP_TRY(dn.alloc((size_t)s->n * 8));
        HIP_TRY(hipMemcpyAsync(dn.p, norms_sq, (size_t)s->n * 8, hipMemcpyHostToDevice, c->stream));
        d_n2 = (const double*)dn.p;
    }
    // Device budget for the kept cells of one row block (raw + sorted words, CSR arrays: 21-22 bytes per cell): a quarter of
    // what is free now unless the caller says otherwise.  Only a block that goes through the exact kernel is planned
    // against it (worst case: every cell kept); the two-stage comparison's output is sized from its candidate count.
    size_t budget = device_budget_bytes;
    if (budget == 0) {
        // Default: a quarter of what is free, but no more than 2^30 worst-case cells per block (8 GiB of packed words):
        // where the exact kernel runs the result is dense and the link, not the kernel, sets the pace -- blocks of that
        // size keep the head of the pipeline (first block computed, nothing to download yet) short.
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        budget = std::min<size_t>(free_b / 4, (size_t)22 << 30);
    }
    const int64_t budget_cells = std::max<int64_t>(1 << 16, (int64_t)(budget / 22));
    // the dense byte matrix (one byte per cell of a row block) may take more: a third of what is free unless the caller set a budget
    size_t dense_budget = device_budget_bytes;
    if (dense_budget == 0) {
        size_t free_b = 0, total_b = 0;
        HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        dense_budget = free_b / 3;
    }
    const size_t piece_bytes = 32u << 20;                       // pinned buffer size: pinning costs ~0.3 ms per MiB
    const int col_bits = bits_for(std::max<int64_t>(s->n - 1, 1));
    const int shift = 16 + col_bits;
    int rc = ensure_download_side(c);
    if (rc) return rc;
    c->st_kernel_ms = 0.0;
    c->st_bytes = c->st_blocks = c->st_pieces = c->st_two_stage = 0;
    bool tiles_phase = false;   // the launches being timed are runs of flagged tiles (ev[6] .. ev[3]), not whole comparisons
    auto add_kernel_ms = [&]() {
        float ms = 0.0f;
        if (c->timing && c->ev_valid[1] && hipEventSynchronize(c->ev[3]) == hipSuccess &&
            hipEventElapsedTime(&ms, tiles_phase ? c->ev[6] : c->ev[2], c->ev[3]) == hipSuccess)
            c->st_kernel_ms += ms;
    };
    StreamOut out;
    out.c = c;
    out.cb = cb;
    out.ecb = ecb;
    out.user = user;
    // a block's way out, in two steps so that the next block's comparison can be queued between them: prepare = the
    // device-side work that is left (encoding the rows, where the caller asked for that), deliver = pieces to the link
    hipStream_t ps = c->stream;                                // where a block is turned into CSR / encoded rows (see `side`)
    bool side = false;
    auto prepare = [&](BlockCsr& blk) -> int { return ecb ? encode_block(c, blk, ps) : MVS_OK; };
    auto deliver = [&](BlockCsr& blk) -> int {
        auto sp = std::make_shared<BlockCsr>(std::move(blk));
        const bool enc = ecb != nullptr;
        StreamOut* o = &out;
        out.enqueue_feed([c, o, sp, enc]() -> int {
            return enc ? feed_encoded(c, *o, *sp, piece_bytes) : feed_block(c, *o, *sp, piece_bytes);
        });
        return MVS_OK;
    };
    // the arrays of block k live in set k & 1: before block k is built the feeder must be through with block k - 2
    auto wait_for_set = [&](int64_t k) {
        if (k >= 2) out.wait_fed(k - 1);
    };
    out.worker = std::thread([&out] { out.run(); });
    out.feeder = std::thread([&out] { out.feed_run(); });
    int64_t total = 0;
    // option stream_trace: where the host is when (ms since the call started)
    const auto t_call = std::chrono::steady_clock::now();
    std::vector<std::pair<std::string, double>> trace;
    auto mark = [&](const char* what, long k) {
        if (!c->opt.stream_trace) return;
        char buf[64];
        snprintf(buf, sizeof buf, "%s[%ld]", what, k);
        trace.emplace_back(buf, std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_call).count());
    };
    auto finish = [&](int status) {
        mark("close", -1);
        out.close_feeder();                                     // every delivered block has been handed to the link
        out.close();                                            // ... and consumed
        (void)hipStreamSynchronize(c->dl_stream);
        if (side) (void)hipStreamSynchronize(c->post_stream);
        mark("done", -1);
        if (c->opt.stream_trace) {
            std::string line = "[mvs stream trace]";
            for (auto& t : trace) {
                char buf[96];
                snprintf(buf, sizeof buf, " %s %.2f", t.first.c_str(), t.second);
                line += buf;
            }
            fprintf(stderr, "%s\n", line.c_str());
        }
        if (n_cells) *n_cells = total;
        if (status != MVS_OK) return status;
        if (!out.error.empty()) return fail(MVS_E_HIP, "%s", out.error.c_str());
        if (out.cb_status != 0) return fail(MVS_E_ABORTED, "the row-block callback returned %d", out.cb_status);
        return MVS_OK;
    };
    const int64_t rows_all = row_end - row_begin;
    // ---------------------------------------------------------------------------------------------------------------
    // How the kept cells leave the device is decided by how dense the result is, which only the filter can tell:
    //  A. sparse: ONE filter pass over the whole row range, candidates re-checked, the few flagged tiles computed, all kept
    //     cells in ONE packed list that is sorted on the device (needs the row field to fit the packed word).
    //  M. dense regions: the dense byte matrix -- one byte per cell, rows -> CSR / encoded rows by count / scan / fill passes
    //     that read only the tiles that can hold something (flagged by the filter, mirror images of those, touched by the
    //     re-check's kept cells: nothing else of the matrix is ever cleared or read) -- in row blocks, so that the link is
    //     fed while the comparison goes on.  Two ways to get there:
    //       M1 (pipeline): the filter itself runs block by block (first block one tile row: its flagged share tells sparse
    //          from dense, and costs 1 % of a whole pass when the answer is "sparse"), so a block's rows are final -- and
    //          on the link -- a millisecond after the call started instead of after the whole filter pass;
    //       M2: plan A's whole filter pass found too many cells for a list: its flags and candidates feed the matrix, the
    //          flagged tiles are computed block by block.
    //  B. the filter does not apply or gave up (nearly every tile dense): the exact kernel in row blocks (dense matrix with
    //     every tile active, or packed lists), as up to round 3.
    // ---------------------------------------------------------------------------------------------------------------
    const bool fits_word = shift + bits_for(std::max<int64_t>(rows_all - 1, 1)) <= 64;
    const int64_t ld = (s->n + 127) / 128 * 128;
    mvs::PairwiseArgs probe{};
    probe.limbs = s->limbs;
    probe.d_pad = s->d_pad;
    const bool dense_ok = mvs::exact_kernel_writes_dense(probe, c->opt) && c->opt.stream_dense != 0;
    const bool matrix_fits = (size_t)rows_all * (size_t)ld <= dense_budget;
    const bool applies = two_stage_applies(c, s, row_begin, row_end, 0, s->n, 0.05);
    mvs::PairwiseArgs wa{};                                          // the whole row range as one symmetric block
    fill_args(c, s, d_n2, keep_mode, row_begin, row_end, 0, s->n, true, false, 0.05, wa);
    // the matrix flows need the tile grids to line up with the matrix's rows and the packed word to hold a row
    const bool can_matrix = applies && dense_ok && matrix_fits && fits_word && row_begin % 256 == 0 && mvs::filter_flags_tiles(wa, c->opt);
    int n_tr_all = 0, n_tc_all = 0;
    mvs::filter_tile_grid(wa, &n_tr_all, &n_tc_all);
    const int tile_o = (int)(row_begin / 256);
    enum { kNone, kM1, kM2 } matrix_mode = kNone;
    TwoStage ts;                                                     // M2: the whole pass; M1: the current block's pass
    mvs::DenseActive active{};                                       // flags == NULL: every tile (plan B)
    const int saved_variant = c->opt.filter_variant;
    struct RestoreVariant {                                          // M1 pins the filter kernel the whole range would get
        mvs_ctx* c; int v;
        ~RestoreVariant() { c->opt.filter_variant = v; }
    } restore_variant{c, saved_variant};
    auto matrix_setup = [&]() -> int {                               // matrix, touch map, list of newly touched tiles
        const void* before = c->st_dense;
        int r = ensure_buf(c, &c->st_dense, &c->st_dense_bytes, (size_t)rows_all * (size_t)ld);
        if (r) return r;
        if (c->st_dense != before) c->st_dense_zero = 0;
        const size_t n_tiles = (size_t)n_tr_all * (size_t)n_tc_all;
        r = ensure_buf(c, &c->pw_ttouch, &c->pw_ttouch_bytes, n_tiles * 4);
        if (r) return r;
        r = ensure_buf(c, &c->pw_tnew, &c->pw_tnew_bytes, (n_tiles + 1) * 4);
        if (r) return r;
        HIP_TRY(hipMemsetAsync(c->pw_ttouch, 0, n_tiles * 4, c->stream));
        HIP_TRY(hipMemsetAsync(c->d_counter + 4, 0, 8, c->stream));      // the "q beyond a byte" flag
        c->st_dense_zero = 0;
        active.touch = (const unsigned int*)c->pw_ttouch;
        active.n_tr = n_tr_all;
        active.n_tc = n_tc_all;
        active.o = tile_o;
        active.sym = (c->opt.pairwise_symmetric != 0) ? 1 : 0;
        return MVS_OK;
    };
    // re-check of t's candidates with the kept cells going into the matrix: a packed list first (the re-check decides
    // which candidates are kept), then mark / clear / scatter (mvs_internal.h: launch_packed_to_dense)
    auto recheck_into_matrix = [&](TwoStage& t) -> int {
        int r = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)(2 * t.n_cand + 64) * 8);
        if (r) return r;
        t.a.dense = nullptr;
        t.a.packed = (unsigned long long*)c->st_raw;
        t.a.capacity = c->st_raw_bytes / 8;
        t.a.pack_row0 = row_begin;
        t.a.pack_shift = shift;
        r = two_stage_recheck(c, t);
        if (r) return r;
        HIP_TRY(hipMemsetAsync(c->d_counter + 10, 0, 8, c->stream));     // count of newly touched tiles
        mvs::launch_packed_to_dense(c->stream, (const unsigned long long*)c->st_raw, c->d_counter, shift,
                                    (1ULL << col_bits) - 1ULL, (uint8_t*)c->st_dense, ld, rows_all, (unsigned int*)c->pw_ttouch, n_tc_all,
                                    (int*)c->pw_tnew, reinterpret_cast<unsigned int*>(c->d_counter + 10),
                                    reinterpret_cast<unsigned int*>(c->d_counter + 4));
        r = check_kernel("k_packed_touch / k_clear_tiles / k_packed_scatter");
        if (r) return r;
        // from here on t.a describes the exact kernel's launches on the flagged tiles: bytes of whole tiles into the matrix
        t.a.packed = nullptr;
        t.a.dense = (uint8_t*)c->st_dense;
        t.a.dense_row0 = row_begin;
        t.a.dense_ld = ld;
        t.a.dense_flag = reinterpret_cast<unsigned int*>(c->d_counter + 4);
        return MVS_OK;
    };
    // M1, one block: filter its rows (symmetric square = the whole row range), re-check into the matrix, its flagged tiles
    auto pipeline_filter = [&](int64_t rb, int64_t re, TwoStage& t) -> int {
        mvs::PairwiseArgs fa{};
        fill_args(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, 0.05, fa);
        fa.sym_begin = row_begin;
        fa.sym_end = row_end;
        t = TwoStage();
        t.ext_flags = (unsigned int*)c->pw_tflag + (size_t)((rb - row_begin) / 256) * (size_t)n_tc_all;
        return two_stage_filter(c, s, d_n2, 0.05, 0, true, 0, fa, t);
    };
    if (applies) {
        bool whole_pass = true;
        if (can_matrix && rows_all > 512 && c->opt.stream_pipeline != 0) {
            // M1's first block doubles as the probe: one tile row of the filter
            rc = ensure_buf(c, &c->pw_tflag, &c->pw_tflag_bytes, (size_t)n_tr_all * (size_t)n_tc_all * 4);
            if (rc) return finish(rc);
            if (c->opt.filter_variant < 0) c->opt.filter_variant = 8;    // what the whole range gets (filter_flags_tiles said so)
            rc = pipeline_filter(row_begin, row_begin + 256, ts);
            if (rc != MVS_OK && rc != kNeedExact) return finish(rc);
            // share of flagged tiles in this row of tiles, extrapolated to the tiles of the whole pass, as list cells
            const double tiles_all = std::max(1.0, (double)n_tr_all * (double)n_tc_all - 0.5 * (double)n_tr_all * (double)(n_tr_all - 1));
            const bool probe_gave_up = rc == kNeedExact;                 // the pass stopped: dense everywhere (never M1 then)
            const double est = probe_gave_up ? 1e30
                                             : ((double)ts.n_flagged * 131072.0 + 2.0 * (double)ts.n_cand) / (double)n_tc_all * tiles_all;
            if (probe_gave_up) {
                // more than 70 % of the first tile row is dense: its cluster alone covers half of the matrix -- no further
                // filter pass, the exact kernel does the shard (plan B; the set is marked, two_stage_filter did that)
                whole_pass = false;
                c->opt.filter_variant = saved_variant;
            } else if (est > (double)c->opt.stream_list_cells) {
                matrix_mode = kM1;
                whole_pass = false;
            } else {
                c->opt.filter_variant = saved_variant;
            }
        }
        if (whole_pass && two_stage_applies(c, s, row_begin, row_end, 0, s->n, 0.05)) {
            ts = TwoStage();
            rc = two_stage_filter(c, s, d_n2, 0.05, 0, true, 0, wa, ts);
            if (rc != MVS_OK && rc != kNeedExact) return finish(rc);
            if (rc == MVS_OK) {
                const unsigned long long bound = 2 * ts.n_cand + (unsigned long long)ts.n_flagged * 131072ULL;
                size_t free_b = 0, total_b = 0;
                HIP_TRY(hipMemGetInfo(&free_b, &total_b));
                const bool list_fits = fits_word && (double)bound * 22.0 <= (double)free_b * 0.5;
                // stream_list_cells (2^26): below that the list (8 B per cell written, a radix sort over the key bits) is
                // cheaper than counting and filling a matrix of rows x n bytes
                const bool as_list = list_fits && (bound <= (unsigned long long)c->opt.stream_list_cells || !can_matrix || ts.n_flagged == 0);
                if (as_list) {
                    rc = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)(bound + 64) * 8);
                    if (rc) return finish(rc);
                    ts.a.packed = (unsigned long long*)c->st_raw;
                    ts.a.capacity = c->st_raw_bytes / 8;
                    ts.a.pack_row0 = row_begin;
                    ts.a.pack_shift = shift;
                    rc = two_stage_recheck(c, ts);
                    if (rc == MVS_OK) rc = two_stage_tiles(c, ts, 0, ts.n_flagged, true);
                    if (rc) return finish(rc);
                    unsigned long long got = 0;
                    hipError_t e = hipMemcpyAsync(&got, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream);
                    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
                    if (e != hipSuccess) return finish(fail(MVS_E_HIP, "reading the cell count: %s", hipGetErrorString(e)));
                    if ((size_t)got * 8 > c->st_raw_bytes) return finish(fail(MVS_E_HIP, "internal: kept cells beyond the sized output"));
                    total = (int64_t)got;
                    add_kernel_ms();
                    c->st_blocks = 1;
                    c->st_two_stage = 1;
                    BlockCsr blk;
                    rc = csr_from_packed(c, row_begin, row_end, (int64_t)got, shift, col_bits, 0, blk);
                    if (rc == MVS_OK) rc = prepare(blk);
                    if (rc == MVS_OK) rc = deliver(blk);
                    return finish(rc);
                }
                if (can_matrix) matrix_mode = kM2;
                // neither a list nor the matrix fits: plan B (the filter pass was in vain)
            }
        }
    }
    // ---- row blocks ----
    // Plan B proper: the exact kernel, software-pipelined -- block k+1 is launched before block k's pieces are fed to the
    // link.  Two ways for a block's cells to leave the kernel:
    //  * dense (two limbs on the ping-pong kernel): one byte per cell in a row-major matrix.  If the matrix of ALL the rows
    //    fits the budget the blocks share it and the symmetric schedule spans the whole square: a block's launch computes its
    //    tiles on and above the diagonal and writes the mirror images into later blocks' rows, so block k is final when
    //    launch k is.  Otherwise the matrix holds one block at a time and the symmetric schedule works inside each block's
    //    own square only;
    //  * packed list (any other kernel): blocks whose worst case -- every cell kept -- fits the budget.
    // The matrix flows M1 / M2 use the same loop with the shared matrix; only what a block's "launch" is differs.
    bool dense = dense_ok, whole = false;
    int64_t block_rows = 0;
    const bool aligned = row_begin % 128 == 0;                      // the symmetric schedule needs the tile grids to line up
    if (dense) {
        whole = aligned && matrix_fits;                              // (the matrix flows imply both)
        if (whole) {
            block_rows = std::max<int64_t>(2048, (rows_all / 16 + 255) / 256 * 256);   // (1/12 .. 1/6 of the rows measure the same or worse)
        } else {
            block_rows = (int64_t)(dense_budget / (size_t)ld) / 256 * 256;
            if (block_rows < 256) dense = false;                     // not even 256 rows of bytes: list blocks instead
        }
        if (dense && c->opt.stream_block_rows > 0)                   // tests: many small blocks on small inputs
            block_rows = std::min<int64_t>(block_rows, std::max<int64_t>(256, (int64_t)c->opt.stream_block_rows / 256 * 256));
    }
    if (!dense) {
        block_rows = std::max<int64_t>(256, budget_cells / std::max<int64_t>(s->n, 1) / 256 * 256);
        while (shift + bits_for(std::max<int64_t>(block_rows - 1, 1)) > 64 && block_rows > 256) block_rows /= 2;
    }
    std::vector<std::pair<int64_t, int64_t>> blocks;
    // blocks of one shared matrix start small (M1: one tile row, the probe; otherwise 1024 rows) and double: the link has
    // nothing to do until the first block has been compared, counted, filled and encoded
    int64_t ramp = block_rows;
    if (dense && whole && c->opt.stream_block_rows == 0 && block_rows > 1024) ramp = 1024;
    if (matrix_mode == kM1) ramp = 256;
    for (int64_t rb = row_begin; rb < row_end;) {
        const int64_t re = std::min(row_end, (rb / 256) * 256 + std::min(ramp, block_rows));
        blocks.emplace_back(rb, re);
        rb = re;
        ramp = std::min(block_rows, ramp * 2);
    }
    if (matrix_mode != kNone) {
        rc = matrix_setup();
        if (rc) return finish(rc);
        active.flags = matrix_mode == kM1 ? (const unsigned int*)c->pw_tflag : (const unsigned int*)ts.a.tile_flag;
        if (matrix_mode == kM1) {
            // flags of blocks not yet filtered read as "not flagged"; block 0 has been filtered already (the probe)
            const size_t done = (size_t)n_tc_all;
            HIP_TRY(hipMemsetAsync((unsigned int*)c->pw_tflag + done, 0, ((size_t)n_tr_all * (size_t)n_tc_all - done) * 4, c->stream));
        }
        rc = recheck_into_matrix(ts);                                // M2: all candidates; M1: block 0's
        if (rc) return finish(rc);
        add_kernel_ms();                                             // filter + re-check
        c->st_two_stage = matrix_mode == kM1 ? 3 : 2;
    } else if (dense) {
        const size_t bytes = (size_t)(whole ? rows_all : std::min(block_rows + 256, rows_all)) * (size_t)ld;
        const void* before = c->st_dense;
        rc = ensure_buf(c, &c->st_dense, &c->st_dense_bytes, bytes);
        if (rc) return finish(rc);
        (void)before;
        c->st_dense_zero = 0;
        hipError_t e = hipMemsetAsync(c->d_counter + 4, 0, 8, c->stream);      // the "q beyond a byte" flag
        if (e != hipSuccess) return finish(fail(MVS_E_HIP, "hipMemsetAsync: %s", hipGetErrorString(e)));
    }
    // M1: which blocks open a filter segment.  Segment 0 is the probe (one tile row); the others end where 4 %, 16 % and 45 %
    // of the rows are done -- by the tiles of the symmetric square that is 8 %, 22 %, 40 % and 30 % of the filter's work --
    // option stream_block_rows (tests) makes every block a segment of its own.
    std::vector<char> seg_first(blocks.size(), 0);
    int64_t seg_row0 = row_begin;
    bool seg_exact = false;
    if (matrix_mode == kM1) {
        const double marks[3] = {0.04, 0.16, 0.45};   // (0.05 / 0.3, 0.03 / 0.12 / 0.3, 0.1 / 0.4 measure the same within 2 %)
        int next_mark = 0;
        for (size_t k = 0; k < blocks.size(); ++k) {
            const double done = (double)(blocks[k].first - row_begin) / (double)rows_all;
            bool opens = k <= 1 || c->opt.stream_block_rows > 0;
            while (next_mark < 3 && done >= marks[next_mark]) {
                opens = true;
                ++next_mark;
            }
            seg_first[k] = opens ? 1 : 0;
        }
    }
    const int saved_filter = c->opt.pairwise_filter;
    // the exact kernel on every tile of rows [rb, re) (plan B; also a block of M1 whose filter pass gave up)
    auto launch_exact = [&](int64_t rb, int64_t re, bool as_dense) -> int {
        unsigned long long got = 0;
        c->opt.pairwise_filter = 0;
        int r;
        if (as_dense) {
            DenseOut dno{(uint8_t*)c->st_dense, whole ? row_begin : rb, ld, whole ? row_begin : rb, whole ? row_end : re,
                         reinterpret_cast<unsigned int*>(c->d_counter + 4)};
            r = pairwise_launch(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, nullptr, 0, 0, &got, 0.05, nullptr, &dno);
        } else {
            const int64_t worst = (re - rb) * s->n;
            r = ensure_buf(c, &c->st_raw, &c->st_raw_bytes, (size_t)worst * 8);
            if (r == MVS_OK) {
                PackedOut po{&c->st_raw, &c->st_raw_bytes, rb, shift, false};
                r = pairwise_launch(c, s, d_n2, keep_mode, rb, re, 0, s->n, true, false, nullptr, 0, 0, &got, 0.05, &po);
            }
        }
        c->opt.pairwise_filter = saved_filter;
        return r;
    };
    auto launch = [&](size_t k, bool as_dense) -> int {
        const int64_t rb = blocks[k].first, re = blocks[k].second;
        if (matrix_mode == kM2 && as_dense) {                        // this block's share of the whole pass's flagged tiles
            tiles_phase = true;
            const int t0 = (int)((rb - row_begin) / 256), t1 = (int)std::min<int64_t>(ts.n_tr, (re - row_begin + 255) / 256);
            return two_stage_tiles(c, ts, ts.row_first[(size_t)t0], ts.row_first[(size_t)t1] - ts.row_first[(size_t)t0], true);
        }
        if (matrix_mode == kM1 && as_dense) {
            // The filter runs per SEGMENT of consecutive row blocks (seg_first: the blocks that open one; the probe's tile
            // row is segment 0): few passes -- each costs a launch over the whole column range and a host round trip -- yet
            // the first rows are final, and on the link, a millisecond after the call started.
            int r = MVS_OK;
            const bool opens = k < seg_first.size() && seg_first[k];
            if (k == 0) {
                seg_row0 = rb;                                       // the probe's tile row: filtered and re-checked already
                seg_exact = false;
            } else if (opens) {
                size_t last = k;
                while (last + 1 < blocks.size() && !seg_first[last + 1]) ++last;
                r = pipeline_filter(rb, blocks[last].second, ts);
                seg_row0 = rb;
                seg_exact = r == kNeedExact;
                if (r == MVS_OK) r = recheck_into_matrix(ts);
            } else if (seg_exact) {
                r = kNeedExact;
            }
            tiles_phase = !(opens && k > 0) && !seg_exact;           // a block that opens a segment is timed ev[2] .. ev[3]
            if (r == kNeedExact) {
                // nearly every tile of these rows is dense: the exact kernel on all of them.  Flag the tiles it writes itself
                // -- outside the square, on and above its diagonal -- so that the row passes read them and their mirror
                // images; the tiles below the diagonal stay what earlier blocks made of them
                const int t0 = (int)((rb - row_begin) / 256), t1 = (int)((re - row_begin + 255) / 256);
                for (int t = t0; t < t1; ++t) {
                    unsigned int* rowf = (unsigned int*)c->pw_tflag + (size_t)t * (size_t)n_tc_all;
                    hipError_t e = hipSuccess;
                    if (tile_o > 0) e = hipMemsetD32Async((hipDeviceptr_t)rowf, 1, (size_t)tile_o, c->stream);
                    if (e == hipSuccess && t + tile_o < n_tc_all)
                        e = hipMemsetD32Async((hipDeviceptr_t)(rowf + t + tile_o), 1, (size_t)(n_tc_all - t - tile_o), c->stream);
                    if (e != hipSuccess) return fail(MVS_E_HIP, "hipMemsetD32Async: %s", hipGetErrorString(e));
                }
                c->filter_off_id = 0;                                // a verdict on these rows, not on the set
                return launch_exact(rb, re, true);
            }
            if (r) return r;
            // this block's share of the segment's flagged tiles (tile rows relative to the segment's first row)
            const int t0 = (int)((rb - seg_row0) / 256), t1 = (int)std::min<int64_t>(ts.n_tr, (re - seg_row0 + 255) / 256);
            return two_stage_tiles(c, ts, ts.row_first[(size_t)t0], ts.row_first[(size_t)t1] - ts.row_first[(size_t)t0], true);
        }
        return launch_exact(rb, re, as_dense);
    };
    auto packed_count = [&](size_t k, int64_t* n) -> int {
        unsigned long long got = 0;
        hipError_t e = hipMemcpyAsync(&got, c->d_counter, 8, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
        if (e != hipSuccess) return fail(MVS_E_HIP, "reading the cell count: %s", hipGetErrorString(e));
        if ((int64_t)got > (blocks[k].second - blocks[k].first) * s->n) return fail(MVS_E_HIP, "internal: more kept cells than cells");
        *n = (int64_t)got;
        return MVS_OK;
    };
    // Blocks of ONE shared matrix: block k's rows are final when launch k is and launch k + 1 never touches them (its
    // mirror images land in later blocks' rows), so block k is counted / scanned / filled / encoded on a SIDE stream while
    // launch k + 1 already runs on the context's stream -- memory-bound passes beside a matrix-core-bound kernel instead
    // of between two of them.  (stream_dense = 2: everything on the context's stream, one after the other.)
    side = dense && whole && blocks.size() > 1 &&
           (c->opt.stream_dense == 3 || (c->opt.stream_dense == 1 && matrix_mode != kM2));
    if (side) ps = c->post_stream;
    mark("setup", -1);
    if (!blocks.empty()) {
        rc = launch(0, dense);
        if (rc) return finish(rc);
    }
    mark("launched", 0);
    for (size_t k = 0; k < blocks.size(); ++k) {
        const int64_t rb = blocks[k].first, re = blocks[k].second;
        BlockCsr blk;
        bool next_launched = false;
        if (side) {
            hipError_t e = hipEventRecord(c->cmp_done, c->stream);              // launch k is the last thing queued there
            if (e == hipSuccess) e = hipStreamWaitEvent(ps, c->cmp_done, 0);
            if (e != hipSuccess) return finish(fail(MVS_E_HIP, "ordering the side stream: %s", hipGetErrorString(e)));
            if (c->timing && c->ev_valid[1]) add_kernel_ms();                    // launch k's time, before its events are reused
            if (k + 1 < blocks.size() && !out.failed()) {
                rc = launch(k + 1, true);
                if (rc) return finish(rc);
                next_launched = true;
                mark("launched", (long)k + 1);
            }
        }
        wait_for_set((int64_t)k);
        if (dense) {
            bool odd = false;
            rc = csr_from_dense(c, rb, re, s->n, whole ? row_begin : rb, ld, (int64_t)k, blk, &odd, ps, active, ecb != nullptr);
            if (rc) return finish(rc);
            mark("csr", (long)k);
            if (!side) add_kernel_ms();
            if (odd) {
                if (side) {                 // back to one stream; a launch already queued for block k + 1 is wasted, not wrong
                    (void)hipStreamSynchronize(c->stream);
                    side = false;
                    ps = c->stream;
                }
                // a kept cell whose q a byte cannot hold (norms that do not belong to the vectors): this block and the
                // rest go through the packed list, each block inside its own square -- the one case where a block is
                // compared a second time
                dense = false;
                matrix_mode = kNone;
                tiles_phase = false;
                c->opt.filter_variant = saved_variant;
                int64_t br = std::max<int64_t>(256, budget_cells / std::max<int64_t>(s->n, 1) / 256 * 256);
                while (shift + bits_for(std::max<int64_t>(br - 1, 1)) > 64 && br > 256) br /= 2;
                std::vector<std::pair<int64_t, int64_t>> rest(blocks.begin(), blocks.begin() + (long)k);
                for (int64_t b0 = rb; b0 < row_end;) {
                    const int64_t b1 = std::min(row_end, (b0 / 256) * 256 + br);
                    rest.emplace_back(b0, b1);
                    b0 = b1;
                }
                blocks.swap(rest);
                rc = launch(k, false);
                if (rc) return finish(rc);
                --k;                                                   // take the block again, as a list this time
                continue;
            }
        } else {
            int64_t n = 0;
            rc = packed_count(k, &n);
            if (rc) return finish(rc);
            add_kernel_ms();
            rc = csr_from_packed(c, rb, re, n, shift, col_bits, (int64_t)k, blk);
            if (rc) return finish(rc);
        }
        total += blk.n;
        ++c->st_blocks;
        rc = prepare(blk);
        if (rc) return finish(rc);
        mark("enc", (long)k);
        if (!next_launched && k + 1 < blocks.size() && !out.failed()) {   // the next block computes while this one is fed to the link
            rc = launch(k + 1, dense);
            if (rc) return finish(rc);
            next_launched = true;
            mark("launched", (long)k + 1);
        }
        rc = deliver(blk);
        if (rc) return finish(rc);
        mark("fed", (long)k);
        if (out.failed()) break;
        (void)next_launched;
    }
    return finish(MVS_OK);
}
}  // namespace

int mvs_ctx_stream_stats(const mvs_ctx* c, double* kernel_ms, int64_t* bytes_out, int64_t* row_blocks, int64_t* pieces,
                         int* two_stage) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (kernel_ms) *kernel_ms = c->st_kernel_ms;
    if (bytes_out) *bytes_out = c->st_bytes;
    if (row_blocks) *row_blocks = c->st_blocks;
    if (pieces) *pieces = c->st_pieces;
    if (two_stage) *two_stage = (int)c->st_two_stage;
    return MVS_OK;
}

int mvs_pairwise_block(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int keep_mode, int64_t row_begin,
                       int64_t row_end, int64_t col_begin, int64_t col_end, int flags, mvs_cell* cells,
                       int64_t capacity, int64_t* n_cells) {
    if (!c || !s || !n_cells) return fail(MVS_E_INVALID, "NULL argument");
    const Range range(c, "mvs_pairwise_block");
    if (capacity < 0 || *n_cells < 0 || (keep_mode != MVS_KEEP_INT32 && keep_mode != MVS_KEEP_INT16) ||
        (flags & ~(MVS_BLOCK_SYMMETRIC | MVS_BLOCK_MIRROR_ALL)) != 0 ||
        ((flags & MVS_BLOCK_SYMMETRIC) && (flags & MVS_BLOCK_MIRROR_ALL)))
        return fail(MVS_E_INVALID, "bad argument");
    if (row_begin < 0 || row_end > s->n || row_begin > row_end || col_begin < 0 || col_end > s->n || col_begin > col_end)
        return fail(MVS_E_INVALID, "block [%lld,%lld) x [%lld,%lld) outside [0,%lld)", (long long)row_begin,
                    (long long)row_end, (long long)col_begin, (long long)col_end, (long long)s->n);
    if ((flags & MVS_BLOCK_SYMMETRIC) && (col_begin > row_begin || col_end < row_end))
        return fail(MVS_E_INVALID, "a symmetric block must contain the square of its row range");
    if (row_begin == row_end || col_begin == col_end) return MVS_OK;
    if (!norms_sq || (capacity > 0 && !cells)) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    unsigned long long count = 0;
    int rc = pairwise_launch(c, s, norms_sq, keep_mode, row_begin, row_end, col_begin, col_end,
                             (flags & MVS_BLOCK_SYMMETRIC) != 0, (flags & MVS_BLOCK_MIRROR_ALL) != 0, cells, capacity,
                             (unsigned long long)*n_cells, &count);
    if (rc) return rc;
    if (count == ~0ULL) {
        {
            const int rb_rc = read_back(c, c->stream, {{&count, c->d_counter, 8}});
            if (rb_rc) return rb_rc;
        }
    }
    *n_cells = (int64_t)count;
    if ((int64_t)count > capacity)
        return fail(MVS_E_CAPACITY, "%llu cells appended but capacity is %lld", count, (long long)capacity);
    return MVS_OK;
}

// -------------------------------------------------------------------------------------------------
// block plans (include/mvs_hip.h "block plans"): a rank's share of the symmetric multi-rank schedule
// -------------------------------------------------------------------------------------------------
struct PlanState {
    bool active = false;
    bool two_stage = false;               // false: the exact kernel block by block (other limb codes, filter off, no derived data)
    const mvs_sketch_set* set = nullptr;
    const double* d_n2 = nullptr;
    int keep_mode = MVS_KEEP_INT32;
    int flags = 0;
    int64_t f0 = 0, f1 = 0;               // the frame's rows
    mvs::PairwiseArgs a{};                // frame, outputs, filter buffers
    int n_tr = 0, n_tc = 0;               // the frame's grid of 256 x 256 tiles
    unsigned long long regions_cap = 0, regions_next = 0;
    std::vector<std::array<int64_t, 4>> blocks;   // every rectangle handed in, in order
    std::vector<int> groups;              // blocks per filter launch
    mvs_cell* cells = nullptr;
    int64_t capacity = 0;
    // what it did (mvs_plan_stats)
    long long tiles = 0, launches = 0, candidates = 0, flagged = 0;
    std::vector<hipEvent_t> ev;           // start / stop per filter launch, created once and reused
    size_t ev_used = 0;
    hipEvent_t e_chk0 = nullptr, e_chk1 = nullptr, e_tiles1 = nullptr;
    bool timed = false, finished = false;
    // Running ahead of the read-backs (option plan_speculate): a plan of the same shape as the previous one sizes its second
    // half -- pruning, re-check, flagged tiles -- from THAT plan's counts and does not wait for its own; every kernel reads the
    // real counts on the device, k_plan_verdict says at the end whether the sizes held (if not, the cell count reads
    // kPlanStale and the caller runs the plan again: it will not speculate).  The counts come to the host with the next
    // read-back anybody does: mvs_cells_report's, or plan_resolve's own.
    // Filter launches alternate between the context's stream and a side stream of the plan (option plan_overlap): the last
    // round of one launch leaves CUs idle that the first round of the next can use.  A side launch waits for everything the
    // caller had put on the context's stream when it was issued (the arrival of its columns); mvs_plan_finish joins them.
    hipStream_t side = nullptr;
    hipEvent_t e_fork = nullptr, e_join = nullptr;
    bool side_busy = false;
    const int8_t* lo_wire = nullptr;      // mvs_plan_wire: the other ranks' limb planes are rebuilt from it, row by row, as needed
    bool need_clean = false;              // the row marks (pw_need) were cleared by this plan's reset and not written since
    std::vector<std::pair<int64_t, int64_t>> meta_done;   // rows whose filter constants are in place (mvs_plan_rows_ready, the frame)
    bool speculate = false;               // this plan
    bool pending = false;                 // its counts are still on the device only
    bool stale = false;                   // (after the counts came in) its sizes did not hold
    bool hints_valid = false;
    long long hint_cand = 0, hint_flagged = 0;
    std::array<int64_t, 8> hint_key{};    // the shape the hints belong to
    std::array<int64_t, 8> key{};
};

static void plan_state_free(mvs_ctx* c) {
    PlanState* st = c->plan;
    if (!st) return;
    for (hipEvent_t e : st->ev)
        if (e) (void)hipEventDestroy(e);
    for (hipEvent_t e : {st->e_chk0, st->e_chk1, st->e_tiles1, st->e_fork, st->e_join})
        if (e) (void)hipEventDestroy(e);
    if (st->side) (void)hipStreamDestroy(st->side);
    delete st;
    c->plan = nullptr;
}

namespace {

// tiles of a rectangle the symmetric schedule computes: everything except the tiles strictly below the diagonal of the square
long long plan_block_tiles(const PlanState& st, const std::array<int64_t, 4>& b) {
    const int64_t n_tr = (b[1] - b[0] + 255) / 256, n_tc = (b[3] - b[2] + 255) / 256;
    long long t = 0;
    for (int64_t r = 0; r < n_tr; ++r) {
        const int64_t i0 = b[0] + r * 256;
        // skipped in this tile row: the tiles with j0 >= f0 and j0 + 256 <= i0 (tile origins share the 256 grid)
        const int64_t lo = std::max(b[2], st.f0), hi = std::min(b[2] + n_tc * 256, i0);      // j0 in [lo, hi - 256]
        const int64_t skipped = hi - lo >= 256 ? (hi - lo) / 256 : 0;
        t += n_tc - skipped;
    }
    return t;
}

// the counter block of a finished speculative plan, as read back: its counts become the next plan's hints
void plan_take_counts(mvs_ctx* c, PlanState& st, const unsigned long long* back) {
    st.pending = false;
    st.candidates = (long long)back[2];
    st.flagged = (long long)back[13];
    st.stale = back[12] != 0;
    c->last_candidates = back[2];
    c->last_flagged_tiles = (long long)back[13];
    c->last_filter_tiles = st.tiles;
    st.hints_valid = !st.stale;
    st.hint_cand = st.candidates;
    st.hint_flagged = st.flagged;
    st.hint_key = st.key;
}

// waits for a speculative plan's counts if nobody has fetched them yet
int plan_resolve(mvs_ctx* c) {
    PlanState* st = c->plan;
    if (!st || !st->pending) return MVS_OK;
    unsigned long long back[33];
    const int rc = read_back(c, c->stream, {{back, c->d_counter, sizeof(back)}});
    if (rc) return rc;
    plan_take_counts(c, *st, back);
    return MVS_OK;
}

// counters, candidate-region headers, tile flags and the row marks of mvs_plan_wire cleared by ONE launch (six memsets were
// nine fill kernels of 5 us each in front of every plan: 45 us of the 1.7 ms a rank of an 8-way split spends on its step)
int plan_reset_counters(mvs_ctx* c, PlanState& st, bool cells_too) {
    void* ptrs[7] = {cells_too ? (void*)c->d_counter : nullptr, c->d_counter + 1, c->d_counter + 5, st.a.recheck_queue,
                     st.regions_cap ? c->pw_chdr : nullptr, st.a.tile_flag, c->pw_need};
    const size_t bytes[7] = {8, 16, 224, 512, (size_t)st.regions_cap * 4, (size_t)st.n_tr * (size_t)st.n_tc * 4,
                             c->pw_need ? std::min((size_t)st.set->n_alloc, c->pw_need_bytes) / 4 * 4 : 0};   // (a buffer sized for an earlier, smaller set: never beyond it)
    if (mvs::launch_zero_ranges(c->stream, ptrs, bytes, 7) != 0) return fail(MVS_E_INVALID, "plan reset: misaligned buffer");
    const int rc = check_kernel("k_zero_ranges");
    if (rc) return rc;
    st.need_clean = c->pw_need != nullptr && st.set->n_alloc % 4 == 0 && c->pw_need_bytes >= (size_t)st.set->n_alloc;
    st.regions_next = 0;
    return MVS_OK;
}

// filter constants of rows [r0, r1) (their statistics and norms must be in place on the stream)
int plan_meta(mvs_ctx* c, PlanState& st, int64_t r0, int64_t r1) {
    if (r1 <= r0) return MVS_OK;
    const mvs_sketch_set* s = st.set;
    mvs::launch_filter_meta(c->stream, s->ext_rows + r0, st.d_n2 + r0, r1 - r0, r1 - r0, s->d, st.a.keep_coeff,
                            (float4*)c->pw_fmeta + r0);
    return check_kernel("k_filter_meta");
}

// everything the plan put on its side stream is ordered before what follows on the context's stream
int plan_join(mvs_ctx* c, PlanState& st) {
    if (!st.side_busy) return MVS_OK;
    HIP_TRY(hipStreamWaitEvent(c->stream, st.e_join, 0));
    st.side_busy = false;
    return MVS_OK;
}

// mvs_plan_wire: the limb planes of the rows outside the frame that the re-check and the flagged tiles are about to read --
// the columns of the gathered candidates and of the flagged tiles -- are rebuilt from low limbs + coarse plane; the others
// keep whatever an earlier step left there (nobody reads them)
int plan_rebuild_needed(mvs_ctx* c, PlanState& st) {
    if (!st.lo_wire) return MVS_OK;
    const mvs_sketch_set* s = st.set;
    int rc = ensure_buf(c, &c->pw_need, &c->pw_need_bytes, (size_t)s->n_alloc);
    if (rc) return rc;
    if (!st.need_clean) HIP_TRY(hipMemsetAsync(c->pw_need, 0, (size_t)s->n_alloc, c->stream));   // (the plan's reset cleared it)
    st.need_clean = false;
    mvs::launch_rows_needed(c->stream, st.a, st.n_tr, st.n_tc, st.f0, st.f1, s->n, (unsigned char*)c->pw_need);
    rc = check_kernel("k_rows_needed");
    if (rc) return rc;
    const int64_t dp = s->d_pad;
    const int64_t lo_end = st.f0 & ~(int64_t)15, hi_begin = (st.f1 + 15) & ~(int64_t)15, hi_end = (s->n + 15) & ~(int64_t)15;
    for (int half = 0; half < 2; ++half) {
        const int64_t r0 = half ? hi_begin : 0, r1 = half ? std::min<int64_t>(hi_end, s->n_alloc) : lo_end;
        if (r1 <= r0) continue;
        mvs::launch_planes_from_wire(c->stream, st.lo_wire + r0 * dp, s->ext_coarse_fm + r0 * dp, s->ext_rows + r0, r1 - r0, s->d_pad,
                                     const_cast<int8_t*>(s->planes) + r0 * 2 * dp, (const unsigned char*)c->pw_need + r0);
        rc = check_kernel("k_planes_from_wire(needed rows)");
        if (rc) return rc;
    }
    return MVS_OK;
}

// blocks [first, first + count) of the plan as ONE filter launch
int plan_launch(mvs_ctx* c, PlanState& st, size_t first, int count) {
    int64_t rect[mvs::kPlanSegs][4];
    for (int k = 0; k < count; ++k)
        for (int x = 0; x < 4; ++x) rect[k][x] = st.blocks[first + (size_t)k][(size_t)x];
    mvs::PlanSegs segs;
    const long long wg = mvs::plan_segments(rect, count, &segs);
    if (wg < 0) return fail(MVS_E_INVALID, "plan launch too large");
    if (wg == 0) return MVS_OK;
    mvs::PairwiseArgs a = st.a;
    const unsigned long long regions = (unsigned long long)wg * 8ull;
    if (st.regions_cap && st.regions_next + regions <= st.regions_cap) {
        a.cand_region_base = st.regions_next;
        st.regions_next += regions;
    } else {
        a.cand_hdr = nullptr;            // this launch's waves append with the atomic
        a.cand_ent = nullptr;
    }
    hipStream_t on = c->stream;
    if (c->plan_overlap != 0 && (st.launches & 1) != 0) {          // every second launch of a plan: the side stream
        if (!st.side) {
            HIP_TRY(hipStreamCreateWithFlags(&st.side, hipStreamNonBlocking));
            HIP_TRY(hipEventCreateWithFlags(&st.e_fork, hipEventDisableTiming));
            HIP_TRY(hipEventCreateWithFlags(&st.e_join, hipEventDisableTiming));
        }
        HIP_TRY(hipEventRecord(st.e_fork, c->stream));
        HIP_TRY(hipStreamWaitEvent(st.side, st.e_fork, 0));
        on = st.side;
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (c->timing) {
        while (st.ev.size() < st.ev_used + 2) {
            hipEvent_t e = nullptr;
            HIP_TRY(hipEventCreate(&e));
            st.ev.push_back(e);
        }
        e0 = st.ev[st.ev_used];
        e1 = st.ev[st.ev_used + 1];
        st.ev_used += 2;
        HIP_TRY(hipEventRecord(e0, on));
    }
    const int rc = mvs::launch_filter_plan(on, a, segs, wg);
    if (rc) return fail(rc, "plan filter launch rejected");
    const int rk = check_kernel("k_pairwise_pp(plan filter)");
    if (rk) return rk;
    if (e1) HIP_TRY(hipEventRecord(e1, on));
    if (on != c->stream) {
        HIP_TRY(hipEventRecord(st.e_join, on));
        st.side_busy = true;
    }
    ++st.launches;
    return MVS_OK;
}

}  // namespace

int mvs_shard_layout(int64_t n_total, int world, int64_t* block_rows, int64_t* block_rows_padded) {
    if (n_total < 0 || world < 1) return fail(MVS_E_INVALID, "bad argument");
    const int64_t rps = (n_total + world - 1) / world;                 // src/pairwise_comp_optimized.cpp:938
    if (block_rows) *block_rows = rps;
    if (block_rows_padded) *block_rows_padded = std::max<int64_t>(256, (rps + 255) / 256 * 256);
    return MVS_OK;
}

int mvs_sketch_set_attach_derived(mvs_sketch_set* s, int8_t* coarse_fm, void* row_stats) {
    if (!s) return fail(MVS_E_INVALID, "set is NULL");
    if ((coarse_fm == nullptr) != (row_stats == nullptr)) return fail(MVS_E_INVALID, "both buffers or neither");
    s->ext_coarse_fm = coarse_fm;
    s->ext_rows = static_cast<mvs::CoarseRow*>(row_stats);
    return MVS_OK;
}

int mvs_sketch_set_prepare_rows(mvs_ctx* c, mvs_sketch_set* s, int64_t row_first, int64_t row_count) {
    if (!c || !s) return fail(MVS_E_INVALID, "NULL argument");
    if (row_first < 0 || row_count < 0 || row_first + row_count > s->n_alloc || (row_first & 15) || (row_count & 15))
        return fail(MVS_E_INVALID, "rows [%lld, +%lld): multiples of 16 inside the %lld allocated rows", (long long)row_first,
                    (long long)row_count, (long long)s->n_alloc);
    if (row_count == 0 || s->limbs != 2 || s->d_pad > 32768 || !s->ext_coarse_fm) return MVS_OK;   // nothing the filter could use
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_buf(c, &c->plan_tmp, &c->plan_tmp_bytes, (size_t)row_count * (size_t)s->d_pad);
    if (rc) return rc;
    mvs::launch_coarse_build(c->stream, s->planes + row_first * 2 * (int64_t)s->d_pad, row_count, row_count, s->d_pad,
                             (int8_t*)c->plan_tmp, s->ext_rows + row_first, c->opt.coarse_radix);
    rc = check_kernel("k_coarse_build(rows)");
    if (rc) return rc;
    mvs::launch_coarse_fm(c->stream, (const int8_t*)c->plan_tmp, row_count, s->d_pad, s->ext_coarse_fm + row_first * (int64_t)s->d_pad);
    return check_kernel("k_coarse_fm(rows)");
}

int mvs_sketch_set_recode_rows(mvs_ctx* c, mvs_sketch_set* s, const void* sketches, int elem_bytes, int64_t n_rows, int64_t row_first,
                               int64_t row_count) {
    if (!c || !s) return fail(MVS_E_INVALID, "NULL argument");
    if ((elem_bytes != 4 && elem_bytes != 2) || n_rows < 0 || row_first < 0 || row_count < n_rows || row_first + row_count > s->n_alloc ||
        (row_first & 15) || (row_count & 15) || (n_rows > 0 && !sketches))
        return fail(MVS_E_INVALID, "rows [%lld, +%lld) (%lld of them given): multiples of 16 inside the %lld allocated rows",
                    (long long)row_first, (long long)row_count, (long long)n_rows, (long long)s->n_alloc);
    if (row_count == 0) return MVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    int8_t* planes = const_cast<int8_t*>(s->planes) + row_first * (int64_t)mvs::planes_of(s->limbs) * s->d_pad;
    if (s->limbs == 2 && s->ext_coarse_fm &&
        mvs::launch_recode_rows(c->stream, sketches, elem_bytes, n_rows, row_count, s->d, s->d_pad, planes,
                                s->ext_coarse_fm + row_first * (int64_t)s->d_pad, s->ext_rows + row_first, c->opt.coarse_radix,
                                c->opt.recode_rows_wg))
        return check_kernel("k_recode_rows");
    // other limb codes, longer sketches, no derived data attached: the separate passes
    if (n_rows > 0) {
        mvs::launch_limb_split(c->stream, sketches, elem_bytes, n_rows, s->d, s->limbs, const_cast<int8_t*>(s->planes), s->d_pad, row_first);
        const int rc = check_kernel("k_limb_split");
        if (rc) return rc;
    }
    return mvs_sketch_set_prepare_rows(c, s, row_first, row_count);
}

int mvs_sketch_set_planes_from_wire(mvs_ctx* c, mvs_sketch_set* s, const int8_t* lo_wire, int64_t row_first, int64_t row_count) {
    if (!c || !s) return fail(MVS_E_INVALID, "NULL argument");
    if (row_first < 0 || row_count < 0 || row_first + row_count > s->n_alloc || (row_first & 15) || (row_count & 15))
        return fail(MVS_E_INVALID, "rows [%lld, +%lld): multiples of 16 inside the %lld allocated rows", (long long)row_first,
                    (long long)row_count, (long long)s->n_alloc);
    if (s->limbs != 2 || !s->ext_coarse_fm || !s->ext_rows)
        return fail(MVS_E_INVALID, "a two-limb set with derived data attached (mvs_sketch_set_attach_derived)");
    if (row_count == 0) return MVS_OK;
    if (!lo_wire) return fail(MVS_E_INVALID, "NULL wire buffer");
    HIP_TRY(hipSetDevice(c->device));
    mvs::launch_planes_from_wire(c->stream, lo_wire + row_first * (int64_t)s->d_pad, s->ext_coarse_fm + row_first * (int64_t)s->d_pad,
                                 s->ext_rows + row_first, row_count, s->d_pad,
                                 const_cast<int8_t*>(s->planes) + row_first * 2 * (int64_t)s->d_pad);
    return check_kernel("k_planes_from_wire");
}

int mvs_plan_begin(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, int keep_mode, int64_t f0, int64_t f1, int flags,
                   mvs_cell* cells, int64_t capacity) {
    if (!c || !s) return fail(MVS_E_INVALID, "NULL argument");
    if (capacity < 0 || (keep_mode != MVS_KEEP_INT32 && keep_mode != MVS_KEEP_INT16) || (flags & ~MVS_PLAN_MIRROR_OUTSIDE) != 0 ||
        f0 < 0 || f1 < f0 || f1 > s->n)
        return fail(MVS_E_INVALID, "bad argument");
    if (!norms_sq || (capacity > 0 && !cells)) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    if (!c->plan) {
        c->plan = new (std::nothrow) PlanState();
        if (!c->plan) return fail(MVS_E_NOMEM, "out of host memory");
    }
    PlanState& st = *c->plan;
    {
        int rr = plan_resolve(c);         // (a sync only if the previous plan's counts were never fetched)
        if (rr) return rr;
        rr = plan_join(c, st);            // (a plan that was begun and never finished)
        if (rr) return rr;
    }
    st.active = false;
    st.finished = false;
    st.lo_wire = nullptr;
    st.set = s;
    st.d_n2 = norms_sq;
    st.keep_mode = keep_mode;
    st.flags = flags;
    st.f0 = f0;
    st.f1 = f1;
    st.cells = cells;
    st.capacity = capacity;
    st.blocks.clear();
    st.groups.clear();
    st.tiles = st.launches = st.candidates = st.flagged = 0;
    st.ev_used = 0;
    st.timed = c->timing;
    st.two_stage = s->limbs == 2 && s->d_pad <= 32768 && c->opt.pairwise_filter != 0 && c->opt.pairwise_variant == 8 &&
                   s->ext_coarse_fm != nullptr && (f0 & 255) == 0 && ((f1 & 255) == 0 || f1 == s->n) && f1 > f0;
    mvs::PairwiseArgs& a = st.a;
    a = mvs::PairwiseArgs{};
    fill_args(c, s, norms_sq, keep_mode, f0, f1, 0, s->n, true, (flags & MVS_PLAN_MIRROR_OUTSIDE) != 0, 0.05, a);
    a.symmetric = 1;                      // the plan's mirror rule needs the square (option pairwise_symmetric does not apply)
    a.plan = 1;
    a.cells = cells;
    a.capacity = (unsigned long long)capacity;
    c->last_candidates = 0;
    c->last_flagged_tiles = 0;
    c->last_filter_tiles = 0;
    if (!st.two_stage) {
        HIP_TRY(hipMemsetAsync(c->d_counter, 0, 8, c->stream));
        st.active = true;
        return MVS_OK;
    }
    mvs::filter_tile_grid(a, &st.n_tr, &st.n_tc);
    int rc = ensure_buf(c, &c->pw_fmeta, &c->pw_fmeta_bytes, (size_t)s->n_alloc * sizeof(float4));
    if (rc) return rc;
    const double frame_cells = (double)(f1 - f0) * (double)s->n;
    const int64_t cand_want = std::max<int64_t>(1 << 20, (int64_t)(frame_cells / 4096.0));
    rc = ensure_buf(c, &c->pw_cand, &c->pw_cand_bytes, (size_t)cand_want * sizeof(int2));
    if (rc) return rc;
    // candidate regions: 8 per workgroup of every launch; a launch pads each rectangle to whole super-patches, so the sum
    // over a plan is a little more than the frame's own padded grid -- a launch that no longer fits appends with atomics
    const unsigned long long n_spr = (unsigned long long)(st.n_tr + 15) / 16, n_spc = (unsigned long long)(st.n_tc + 15) / 16;
    st.regions_cap = c->opt.cand_regions ? n_spr * (n_spc + 8) * 2048ull : 0;
    if (st.regions_cap > (8ull << 20)) st.regions_cap = 0;
    if (st.regions_cap) {
        rc = ensure_buf(c, &c->pw_chdr, &c->pw_chdr_bytes, (size_t)st.regions_cap * 4);
        if (rc) return rc;
        rc = ensure_buf(c, &c->pw_cent, &c->pw_cent_bytes, (size_t)st.regions_cap * mvs::kCandRegion * sizeof(int2));
        if (rc) return rc;
    }
    rc = ensure_buf(c, &c->pw_tflag, &c->pw_tflag_bytes, (size_t)st.n_tr * (size_t)st.n_tc * 4);
    if (rc) return rc;
    rc = ensure_buf(c, &c->pw_trow, &c->pw_trow_bytes, (size_t)st.n_tr * 4);
    if (rc) return rc;
    a.coarse = nullptr;                   // plans read the fragment-major plane only
    a.coarse_fm = s->ext_coarse_fm;
    a.planes_fm = nullptr;                // flagged tiles: the exact kernel copies from the row-major limb planes
    a.fmeta = (const float4*)c->pw_fmeta;
    a.cand = (int2*)c->pw_cand;
    a.cand_capacity = c->pw_cand_bytes / sizeof(int2);
    a.cand_counter = c->d_counter + 2;
    a.cand_limit = ~0ULL;
    a.cand_stop = reinterpret_cast<unsigned int*>(c->d_counter + 32);
    a.recheck_queue = c->d_counter + 128;
    a.recheck_mode = c->opt.recheck_mode;
    a.cand_hdr = st.regions_cap ? (unsigned int*)c->pw_chdr : nullptr;
    a.cand_ent = st.regions_cap ? (int2*)c->pw_cent : nullptr;
    a.tile_flag = (unsigned int*)c->pw_tflag;
    a.tile_flag_ld = st.n_tc;
    a.tile_dense_thr = c->opt.tile_dense_thr > 0 ? (unsigned)c->opt.tile_dense_thr : 0xffffffffu;
    a.tile_flag_count = reinterpret_cast<unsigned int*>(c->d_counter + 8);
    a.tile_flag_limit = 0xffffffffu;      // a plan never gives up on its filter: dense tiles go to the exact kernel one by one
    if (s->n_alloc % 4 == 0) {            // the row marks of mvs_plan_wire: cleared with the counters (one launch)
        rc = ensure_buf(c, &c->pw_need, &c->pw_need_bytes, (size_t)s->n_alloc);
        if (rc) return rc;
    }
    rc = plan_reset_counters(c, st, true);
    if (rc) return rc;
    rc = plan_meta(c, st, f0, std::min<int64_t>(f1, s->n));
    if (rc) return rc;
    st.meta_done.clear();
    st.meta_done.emplace_back(f0, f1);
    st.key = {f0, f1, s->n, (int64_t)s->d_pad, (int64_t)flags, (int64_t)keep_mode, (int64_t)s->d, capacity};
    st.speculate = c->opt.plan_speculate != 0 && st.hints_valid && st.key == st.hint_key;
    st.stale = false;
    st.active = true;
    return MVS_OK;
}

int mvs_plan_filter(mvs_ctx* c, const mvs_plan_block* blocks, int n_blocks) {
    if (!c || !c->plan || !c->plan->active || c->plan->finished) return fail(MVS_E_INVALID, "no plan in progress (mvs_plan_begin)");
    if (n_blocks < 0 || (n_blocks > 0 && !blocks)) return fail(MVS_E_INVALID, "bad argument");
    PlanState& st = *c->plan;
    const mvs_sketch_set* s = st.set;
    HIP_TRY(hipSetDevice(c->device));
    const size_t first = st.blocks.size();
    for (int k = 0; k < n_blocks; ++k) {
        const mvs_plan_block& b = blocks[k];
        const bool rows_ok = b.row_begin >= st.f0 && b.row_end <= st.f1 && b.row_begin <= b.row_end;
        const bool cols_ok = b.col_begin >= 0 && b.col_end <= s->n && b.col_begin <= b.col_end;
        const bool inside = b.col_begin >= st.f0 && b.col_end <= st.f1, outside = b.col_end <= st.f0 || b.col_begin >= st.f1;
        if (!rows_ok || !cols_ok || !(inside || outside || b.col_begin == b.col_end))
            return fail(MVS_E_INVALID, "plan block [%lld,%lld) x [%lld,%lld): rows inside the frame [%lld,%lld), columns inside or outside its square",
                        (long long)b.row_begin, (long long)b.row_end, (long long)b.col_begin, (long long)b.col_end, (long long)st.f0, (long long)st.f1);
        if (st.two_stage && (((b.row_begin | b.col_begin) & 255) != 0 || ((b.row_end & 255) != 0 && b.row_end != st.f1) ||
                             ((b.col_end & 255) != 0 && b.col_end != s->n)))
            return fail(MVS_E_INVALID, "plan block bounds must sit on multiples of 256 rows / columns");
        if (b.row_begin == b.row_end || b.col_begin == b.col_end) continue;
        // a dispatch holds at most 2^32 work-items per dimension: a rectangle whose padded grid (super-patches of 16 x 16 tiles,
        // 256 workgroups of 512 threads each) is beyond 2^22 workgroups -- a single 1M x 1M block -- is cut into column strips of
        // whole patch columns (4096 columns), each a rectangle of its own
        const int64_t n_spr = ((b.row_end - b.row_begin + 255) / 256 + 15) / 16, n_spc = ((b.col_end - b.col_begin + 255) / 256 + 15) / 16;
        const int64_t per = std::max<int64_t>(1, (int64_t)c->opt.plan_strip_wgs / (n_spr * 256));
        if (!st.two_stage || n_spc <= per) {
            st.blocks.push_back({b.row_begin, b.row_end, b.col_begin, b.col_end});
        } else {
            for (int64_t c0 = b.col_begin; c0 < b.col_end; c0 += per * 4096)
                st.blocks.push_back({b.row_begin, b.row_end, c0, std::min<int64_t>(b.col_end, c0 + per * 4096)});
        }
    }
    const size_t added = st.blocks.size() - first;
    if (added == 0) return MVS_OK;
    if (!st.two_stage) {
        for (size_t k = first; k < st.blocks.size(); ++k) {
            const auto& b = st.blocks[k];
            const bool inside = b[2] >= st.f0 && b[3] <= st.f1;
            unsigned long long count = 0;
            const int rc = pairwise_launch(c, s, st.d_n2, st.keep_mode, b[0], b[1], b[2], b[3], inside,
                                           !inside && (st.flags & MVS_PLAN_MIRROR_OUTSIDE) != 0, st.cells, st.capacity, kKeepCount, &count);
            if (rc) return rc;
            ++st.launches;
        }
        return MVS_OK;
    }
    for (size_t k = first; k < st.blocks.size(); ++k) {
        const auto& b = st.blocks[k];
        bool have = false;                                 // the frame's rows; rows announced by mvs_plan_rows_ready
        for (const auto& r : st.meta_done) have = have || (b[2] >= r.first && b[3] <= r.second);
        if (!have) {                                        // columns outside: their constants are not there yet
            const int rc = plan_meta(c, st, b[2], b[3]);
            if (rc) return rc;
        }
        st.tiles += plan_block_tiles(st, b);
    }
    // one launch per group of rectangles: at most kPlanSegs of them and 2^23 - 1 workgroups (2^32 work-items) together
    auto padded = [](const std::array<int64_t, 4>& b) {
        return (((b[1] - b[0] + 255) / 256 + 15) / 16) * (((b[3] - b[2] + 255) / 256 + 15) / 16) * 256;
    };
    for (size_t k = first; k < st.blocks.size();) {
        int count = 0;
        int64_t wg = 0;
        while (k + (size_t)count < st.blocks.size() && count < mvs::kPlanSegs &&
               (count == 0 || wg + padded(st.blocks[k + (size_t)count]) < 2 * (int64_t)c->opt.plan_strip_wgs)) {
            wg += padded(st.blocks[k + (size_t)count]);
            ++count;
        }
        const int rc = plan_launch(c, st, k, count);
        if (rc) return rc;
        st.groups.push_back(count);
        k += (size_t)count;
    }
    return MVS_OK;
}

int mvs_plan_wire(mvs_ctx* c, const int8_t* lo_wire) {
    if (!c || !c->plan || !c->plan->active || c->plan->finished) return fail(MVS_E_INVALID, "no plan in progress (mvs_plan_begin)");
    PlanState& st = *c->plan;
    if (lo_wire && (!st.two_stage || !st.set->ext_coarse_fm || !st.set->ext_rows))
        return fail(MVS_E_INVALID, "a plan with a filter on a two-limb set with derived data attached (others: mvs_sketch_set_planes_from_wire)");
    st.lo_wire = lo_wire;
    return MVS_OK;
}

int mvs_plan_rows_ready(mvs_ctx* c, int64_t row_begin, int64_t row_end) {
    if (!c || !c->plan || !c->plan->active || c->plan->finished) return fail(MVS_E_INVALID, "no plan in progress (mvs_plan_begin)");
    PlanState& st = *c->plan;
    if (row_begin < 0 || row_end < row_begin || row_end > st.set->n) return fail(MVS_E_INVALID, "rows outside the sketch set");
    if (!st.two_stage || row_begin == row_end) return MVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    // the part below and the part above the frame (the frame's own constants are in place and its filter may be reading them)
    const int64_t parts[2][2] = {{row_begin, std::min(row_end, st.f0)}, {std::max(row_begin, st.f1), row_end}};
    for (const auto& p : parts) {
        if (p[1] <= p[0]) continue;
        const int rc = plan_meta(c, st, p[0], p[1]);
        if (rc) return rc;
        st.meta_done.emplace_back(p[0], p[1]);
    }
    return MVS_OK;
}

int mvs_plan_finish(mvs_ctx* c, const uint64_t** d_count) {
    if (!c || !c->plan || !c->plan->active || c->plan->finished) return fail(MVS_E_INVALID, "no plan in progress (mvs_plan_begin)");
    PlanState& st = *c->plan;
    const mvs_sketch_set* s = st.set;
    HIP_TRY(hipSetDevice(c->device));
    if (d_count) *d_count = reinterpret_cast<const uint64_t*>(c->d_counter);
    st.finished = true;
    st.active = false;
    if (!st.two_stage || st.blocks.empty()) return MVS_OK;
    {
        const int rj = plan_join(c, st);
        if (rj) return rj;
    }
    auto lazy_event = [&](hipEvent_t& e) -> int {
        if (!e) HIP_TRY(hipEventCreate(&e));
        return MVS_OK;
    };
    if (st.timed) {
        int rc = lazy_event(st.e_chk0);
        if (rc) return rc;
        rc = lazy_event(st.e_chk1);
        if (rc) return rc;
        rc = lazy_event(st.e_tiles1);
        if (rc) return rc;
        HIP_TRY(hipEventRecord(st.e_chk0, c->stream));
    }
    if (st.speculate) {
        // sizes from the previous plan of this shape; counts from the device; no host round trip (PlanState::speculate)
        if (st.regions_next > 0) {
            mvs::launch_cand_gather(c->stream, st.a, (int64_t)st.regions_next);
            const int rc = check_kernel("k_cand_gather");
            if (rc) return rc;
        }
        mvs::launch_tile_count(c->stream, st.a.tile_flag, st.n_tr, st.n_tc, (int*)c->pw_trow);
        int rc = check_kernel("k_tile_count");
        if (rc) return rc;
        rc = plan_rebuild_needed(c, st);
        if (rc) return rc;
        const bool tiles_pass = st.hint_flagged > 0;
        const int tile_cap = tiles_pass ? (int)std::min<long long>((long long)st.n_tr * st.n_tc, 2 * st.hint_flagged + 64) : 0;
        rc = ensure_buf(c, &c->pw_tlist, &c->pw_tlist_bytes, ((size_t)tile_cap + 1) * 4);
        if (rc) return rc;
        mvs::PairwiseArgs a = st.a;
        mvs::launch_tile_list(c->stream, a.tile_flag, st.n_tr, st.n_tc, (const int*)c->pw_trow, (int*)c->pw_tlist, tile_cap);
        rc = check_kernel("k_tile_list");
        if (rc) return rc;
        if (tiles_pass) {
            rc = ensure_buf(c, &c->pw_cand2, &c->pw_cand2_bytes, c->pw_cand_bytes);      // whatever the list holds fits
            if (rc) return rc;
            mvs::launch_cand_prune(c->stream, a, 0, (int2*)c->pw_cand2, c->d_counter + 6, st.hint_cand + st.hint_cand / 4);
            rc = check_kernel("k_cand_prune");
            if (rc) return rc;
            a.cand = (int2*)c->pw_cand2;
            a.cand_capacity = c->pw_cand2_bytes / sizeof(int2);
            a.cand_counter = c->d_counter + 6;
            rc = ensure_buf(c, &c->pw_thr, &c->pw_thr_bytes, (size_t)s->n_alloc * 4);
            if (rc) return rc;
            mvs::launch_cand_thr(c->stream, st.d_n2, s->n, s->n_alloc, s->d, a.keep_coeff, (int32_t*)c->pw_thr);
            rc = check_kernel("k_cand_thr");
            if (rc) return rc;
            a.cand_thr = (const int32_t*)c->pw_thr;
        }
        rc = mvs::launch_exact_pairs(c->stream, a, c->opt, st.hint_cand + st.hint_cand / 4);
        if (rc) return fail(rc, "exact re-check launch rejected");
        rc = check_kernel("k_exact_pairs");
        if (rc) return rc;
        if (st.timed) HIP_TRY(hipEventRecord(st.e_chk1, c->stream));
        if (tiles_pass) {
            rc = mvs::launch_exact_tiles(c->stream, a, (const int*)c->pw_tlist + 1, tile_cap, c->opt, true);
            if (rc) return fail(rc, "exact tile launch rejected");
            rc = check_kernel("k_pairwise_pp(tiles)");
            if (rc) return rc;
        }
        if (st.timed) HIP_TRY(hipEventRecord(st.e_tiles1, c->stream));
        mvs::launch_plan_verdict(c->stream, c->d_counter, st.a.cand_capacity, (const int*)c->pw_tlist, tile_cap, !tiles_pass);
        rc = check_kernel("k_plan_verdict");
        if (rc) return rc;
        st.pending = true;
        return MVS_OK;
    }
    std::vector<int> row_count((size_t)st.n_tr);
    unsigned long long back[33];
    for (int attempt = 0;; ++attempt) {
        if (st.regions_next > 0) {
            mvs::launch_cand_gather(c->stream, st.a, (int64_t)st.regions_next);
            const int rc = check_kernel("k_cand_gather");
            if (rc) return rc;
        }
        mvs::launch_tile_count(c->stream, st.a.tile_flag, st.n_tr, st.n_tc, (int*)c->pw_trow);
        int rc = check_kernel("k_tile_count");
        if (rc) return rc;
        // the plan's ONE host synchronisation: the later launches are sized from these counts
        rc = read_back(c, c->stream, {{back, c->d_counter, sizeof(back)}, {row_count.data(), c->pw_trow, (size_t)st.n_tr * 4}});
        if (rc) return rc;
        st.candidates = (long long)back[2];
        if (back[2] <= st.a.cand_capacity) break;
        if (attempt >= 2) return fail(MVS_E_HIP, "internal: the candidate list keeps outgrowing its buffer");
        // the list did not hold the candidates: grow it and run the plan's filter launches again (their inputs are resident)
        rc = ensure_buf(c, &c->pw_cand, &c->pw_cand_bytes, (size_t)(back[2] + back[2] / 4) * sizeof(int2));
        if (rc) return rc;
        st.a.cand = (int2*)c->pw_cand;
        st.a.cand_capacity = c->pw_cand_bytes / sizeof(int2);
        rc = plan_reset_counters(c, st, true);
        if (rc) return rc;
        st.launches = 0;
        st.ev_used = 0;
        size_t k = 0;
        for (int count : st.groups) {
            rc = plan_launch(c, st, k, count);
            if (rc) return rc;
            k += (size_t)count;
        }
        rc = plan_join(c, st);
        if (rc) return rc;
    }
    c->last_candidates = (unsigned long long)st.candidates;
    c->last_filter_tiles = st.tiles;
    {
        const int rw = plan_rebuild_needed(c, st);
        if (rw) return rw;
    }
    int n_flagged = 0;
    std::vector<int> row_first((size_t)st.n_tr + 1, 0);
    for (int t = 0; t < st.n_tr; ++t) row_first[(size_t)t + 1] = row_first[(size_t)t] + row_count[(size_t)t];
    n_flagged = row_first[(size_t)st.n_tr];
    st.flagged = n_flagged;
    c->last_flagged_tiles = n_flagged;
    mvs::PairwiseArgs a = st.a;
    const int* d_list = nullptr;
    if (n_flagged > 0) {
        int rc = ensure_buf(c, &c->pw_tlist, &c->pw_tlist_bytes, ((size_t)n_flagged + 1) * 4);
        if (rc) return rc;
        mvs::launch_tile_list(c->stream, a.tile_flag, st.n_tr, st.n_tc, (const int*)c->pw_trow, (int*)c->pw_tlist);
        rc = check_kernel("k_tile_list");
        if (rc) return rc;
        d_list = (const int*)c->pw_tlist + 1;
        if (st.candidates > 0) {
            rc = ensure_buf(c, &c->pw_cand2, &c->pw_cand2_bytes, (size_t)st.candidates * sizeof(int2));
            if (rc) return rc;
            mvs::launch_cand_prune(c->stream, a, (unsigned long long)st.candidates, (int2*)c->pw_cand2, c->d_counter + 6);
            rc = check_kernel("k_cand_prune");
            if (rc) return rc;
            a.cand = (int2*)c->pw_cand2;
            a.cand_capacity = c->pw_cand2_bytes / sizeof(int2);
            a.cand_counter = c->d_counter + 6;
        }
        rc = ensure_buf(c, &c->pw_thr, &c->pw_thr_bytes, (size_t)s->n_alloc * 4);
        if (rc) return rc;
        mvs::launch_cand_thr(c->stream, st.d_n2, s->n, s->n_alloc, s->d, a.keep_coeff, (int32_t*)c->pw_thr);
        rc = check_kernel("k_cand_thr");
        if (rc) return rc;
        a.cand_thr = (const int32_t*)c->pw_thr;
    }
    if (st.candidates > 0) {
        int rc = mvs::launch_exact_pairs(c->stream, a, c->opt, st.candidates);
        if (rc) return fail(rc, "exact re-check launch rejected");
        rc = check_kernel("k_exact_pairs");
        if (rc) return rc;
    }
    if (st.timed) HIP_TRY(hipEventRecord(st.e_chk1, c->stream));
    if (n_flagged > 0) {
        int rc = mvs::launch_exact_tiles(c->stream, a, d_list, n_flagged, c->opt);
        if (rc) return fail(rc, "exact tile launch rejected");
        rc = check_kernel("k_pairwise_pp(tiles)");
        if (rc) return rc;
    }
    if (st.timed) HIP_TRY(hipEventRecord(st.e_tiles1, c->stream));
    st.hints_valid = true;
    st.hint_cand = st.candidates;
    st.hint_flagged = st.flagged;
    st.hint_key = st.key;
    return MVS_OK;
}

int mvs_plan_stats(mvs_ctx* c, double ms[4], int64_t counts[6]) {
    if (!c || !c->plan) return fail(MVS_E_INVALID, "no plan has run on this context");
    PlanState& st = *c->plan;
    {
        const int rr = plan_resolve(c);
        if (rr) return rr;
    }
    if (ms) {
        ms[0] = ms[1] = ms[2] = ms[3] = 0.0;
        if (st.timed && st.two_stage && st.finished && st.ev_used >= 2 && st.e_tiles1) {
            HIP_TRY(hipEventSynchronize(st.e_tiles1));
            // the time during which at least one filter launch ran (launches on the two streams overlap: plan_launch)
            std::vector<std::pair<float, float>> iv;
            for (size_t k = 0; k + 1 < st.ev_used; k += 2) {
                float b = 0.0f, d = 0.0f;
                if (k) HIP_TRY(hipEventElapsedTime(&b, st.ev[0], st.ev[k]));
                HIP_TRY(hipEventElapsedTime(&d, st.ev[k], st.ev[k + 1]));
                iv.emplace_back(b, b + d);
            }
            std::sort(iv.begin(), iv.end());
            float upto = -1e30f;
            for (const auto& x : iv) {
                if (x.second <= upto) continue;
                ms[0] += x.second - std::max(x.first, upto);
                upto = x.second;
            }
            float t = 0.0f;
            HIP_TRY(hipEventElapsedTime(&t, st.e_chk0, st.e_chk1));
            ms[1] = t;
            HIP_TRY(hipEventElapsedTime(&t, st.e_chk1, st.e_tiles1));
            ms[2] = t;
            HIP_TRY(hipEventElapsedTime(&t, st.ev[0], st.e_tiles1));
            ms[3] = t;
        }
    }
    if (counts) {
        counts[0] = st.candidates;
        counts[1] = st.flagged;
        counts[2] = st.tiles;
        counts[3] = st.launches;
        counts[4] = (st.two_stage ? 0 : 1) | (st.speculate ? 2 : 0) | (st.stale ? 4 : 0);
        counts[5] = st.set ? st.set->d_pad : 0;
    }
    return MVS_OK;
}

int mvs_cells_route(mvs_ctx* c, const mvs_cell* raw, const uint64_t* d_n_raw, int64_t raw_capacity, int64_t block_rows_padded,
                    int64_t block_rows, int64_t n_total, int64_t own_begin, int64_t own_end, mvs_cell* own_out, int64_t own_capacity,
                    uint64_t* d_own_count, void* send, int64_t foreign_capacity, int64_t status, int64_t max_abs) {
    if (!c || !d_n_raw || !d_own_count) return fail(MVS_E_INVALID, "NULL argument");
    if (raw_capacity < 0 || block_rows_padded < 1 || block_rows < 0 || block_rows > block_rows_padded || n_total < 0 ||
        own_begin < 0 || own_end < own_begin || own_end > n_total || own_capacity < 0 || foreign_capacity < 0 ||
        n_total >= (1LL << 31) - 256 || (raw_capacity > 0 && !raw) || (own_capacity > 0 && !own_out))
        return fail(MVS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    {   // count, max, per-row counts; the header of the send buffer -- one launch
        void* ptrs[2] = {d_own_count, send};
        const size_t bytes[2] = {16 + 4 * (size_t)(own_end - own_begin + 1), send ? (size_t)MVS_CELLS_HEADER_BYTES : 0};
        if (mvs::launch_zero_ranges(c->stream, ptrs, bytes, 2) != 0) return fail(MVS_E_INVALID, "state block / send buffer not 4-byte aligned");
        const int rz = check_kernel("k_zero_ranges");
        if (rz) return rz;
    }
    c->rows_max_done = nullptr;
    mvs::launch_cells_route(c->stream, raw, reinterpret_cast<const unsigned long long*>(d_n_raw), (unsigned long long)raw_capacity,
                            block_rows_padded, block_rows, n_total, (int)own_begin, (int)own_end, own_out,
                            (unsigned long long)own_capacity, reinterpret_cast<unsigned long long*>(d_own_count),
                            static_cast<unsigned long long*>(send), (unsigned long long)foreign_capacity, status, max_abs);
    return check_kernel("k_cells_route");
}

int mvs_cells_collect(mvs_ctx* c, const void* recv, int world, int rank, int64_t foreign_capacity, int64_t own_begin, int64_t own_end,
                      mvs_cell* own_out, int64_t own_capacity, uint64_t* d_own_count) {
    if (!c || !d_own_count) return fail(MVS_E_INVALID, "NULL argument");
    if (world < 1 || rank < 0 || rank >= world || foreign_capacity < 0 || own_capacity < 0 || own_begin < 0 || own_end < own_begin ||
        (world > 1 && !recv) || (own_capacity > 0 && !own_out))
        return fail(MVS_E_INVALID, "bad argument");
    if (world == 1) return MVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    c->rows_max_done = nullptr;
    mvs::launch_cells_collect(c->stream, static_cast<const unsigned long long*>(recv), world, rank, (unsigned long long)foreign_capacity,
                              (int)own_begin, (int)own_end, own_out, (unsigned long long)own_capacity,
                              reinterpret_cast<unsigned long long*>(d_own_count));
    return check_kernel("k_cells_collect");
}

int mvs_cells_sort_rows(mvs_ctx* c, const mvs_cell* cells_in, int64_t n, int64_t own_begin, int64_t own_end, const uint64_t* d_own_state,
                        mvs_cell* cells_out) {
    if (!c || !d_own_state) return fail(MVS_E_INVALID, "NULL argument");
    if (n < 0 || own_begin < 0 || own_end < own_begin || own_end - own_begin >= (1LL << 30)) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0 || own_end == own_begin) return MVS_OK;
    if (!cells_in || !cells_out || cells_in == cells_out) return fail(MVS_E_INVALID, "need two distinct device buffers");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = 0;
    int rc = mvs::sort_cells_rows(c->stream, cells_in, cells_out, n, (int)own_begin, (int)(own_end - own_begin),
                                  reinterpret_cast<const unsigned long long*>(d_own_state), nullptr, 0, &need);
    if (rc) return fail(rc, "row sort sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::sort_cells_rows(c->stream, cells_in, cells_out, n, (int)own_begin, (int)(own_end - own_begin),
                              reinterpret_cast<const unsigned long long*>(d_own_state), c->pw_sort, c->pw_sort_bytes, nullptr);
    if (rc) return fail(rc, "row sort failed");
    return check_kernel("k_rows_sort");
}

int mvs_cells_sort_rows_ahead(mvs_ctx* c, const mvs_cell* cells_in, int64_t in_capacity, int64_t own_begin, int64_t own_end,
                              const uint64_t* d_own_state, mvs_cell* cells_out, int64_t out_capacity) {
    if (!c || !d_own_state) return fail(MVS_E_INVALID, "NULL argument");
    if (in_capacity < 0 || out_capacity < 0 || own_begin < 0 || own_end < own_begin || own_end - own_begin >= (1LL << 30) ||
        in_capacity >= (1LL << 32) || out_capacity >= (1LL << 32))
        return fail(MVS_E_INVALID, "bad argument");
    if (in_capacity == 0 || out_capacity == 0 || own_end == own_begin) return MVS_OK;   // (a shard without rows has nothing to order)
    if (!cells_in || !cells_out || cells_in == cells_out) return fail(MVS_E_INVALID, "need two distinct device buffers");
    HIP_TRY(hipSetDevice(c->device));
    size_t need = 0;
    int rc = mvs::sort_cells_rows(c->stream, cells_in, cells_out, 0, (int)own_begin, (int)(own_end - own_begin),
                                  reinterpret_cast<const unsigned long long*>(d_own_state), nullptr, 0, &need, in_capacity, out_capacity);
    if (rc) return fail(rc, "row sort sizing failed");
    rc = ensure_buf(c, &c->pw_sort, &c->pw_sort_bytes, need);
    if (rc) return rc;
    rc = mvs::sort_cells_rows(c->stream, cells_in, cells_out, 0, (int)own_begin, (int)(own_end - own_begin),
                              reinterpret_cast<const unsigned long long*>(d_own_state), c->pw_sort, c->pw_sort_bytes, nullptr, in_capacity,
                              out_capacity);
    if (rc) return fail(rc, "row sort failed");
    c->rows_max_done = d_own_state;        // the scan left the widest row in the state block: the report need not look again
    return check_kernel("k_rows_sort");
}

int mvs_cells_report(mvs_ctx* c, const void* recv, int world, int64_t foreign_capacity, int64_t own_rows, uint64_t* d_own_count,
                     int64_t* out) {
    if (!c || !d_own_count || !out || world < 1 || foreign_capacity < 0 || own_rows < 0 || (world > 1 && !recv))
        return fail(MVS_E_INVALID, "bad argument");
    HIP_TRY(hipSetDevice(c->device));
    // both read-backs land in the context's pinned buffer (a copy into pageable memory is staged and blocks per copy)
    const size_t hdr_bytes = (size_t)world * MVS_CELLS_HEADER_BYTES;
    const int rc = ensure_read_back(c, 64 + hdr_bytes + 33 * 8);
    if (rc) return rc;
    unsigned long long* own = static_cast<unsigned long long*>(c->rb_pinned);
    unsigned long long* hdr = own + 8;
    unsigned long long* plan_back = hdr + (size_t)world * 8;
    memset(c->rb_pinned, 0, 64 + hdr_bytes);
    // a plan that ran ahead of its read-backs: its counts come along with this one
    const bool with_plan = c->plan && c->plan->pending;
    if (with_plan) HIP_TRY(hipMemcpyAsync(plan_back, c->d_counter, 33 * 8, hipMemcpyDeviceToHost, c->stream));
    if (c->rows_max_done != d_own_count)
        mvs::launch_rows_max(c->stream, reinterpret_cast<unsigned long long*>(d_own_count), (int)own_rows);
    HIP_TRY(hipMemcpyAsync(own, d_own_count, 16, hipMemcpyDeviceToHost, c->stream));
    if (recv) {
        const size_t stride = MVS_CELLS_HEADER_BYTES + (size_t)foreign_capacity * sizeof(mvs_cell);
        HIP_TRY(hipMemcpy2DAsync(hdr, MVS_CELLS_HEADER_BYTES, recv, stride, MVS_CELLS_HEADER_BYTES, (size_t)world,
                                 hipMemcpyDeviceToHost, c->stream));
    }
    if (c->report_spin > 0) {
        // the step's one host synchronisation: the device is typically a fraction of a millisecond from done, and a blocked
        // thread is woken by an interrupt tens of microseconds after the stream drained -- poll first, block if it takes long
        const auto t0 = std::chrono::steady_clock::now();
        while (hipStreamQuery(c->stream) == hipErrorNotReady &&
               std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() < (double)c->report_spin) {
        }
    }
    HIP_TRY(hipStreamSynchronize(c->stream));
    if (with_plan) plan_take_counts(c, *c->plan, plan_back);
    out[0] = (int64_t)own[0];
    for (int r = 0; r < world; ++r)
        for (int k = 0; k < 5; ++k) out[1 + r * 5 + k] = (int64_t)hdr[(size_t)r * 8 + (size_t)k];
    out[1 + 5 * world] = (int64_t)(own[1] & 0xffffffffULL);
    return MVS_OK;
}

int mvs_search_block(mvs_ctx* c, const mvs_sketch_set* s, const double* norms_sq, double jaccard_min,
                     int64_t row_begin, int64_t row_end, int64_t col_begin, int64_t col_end, mvs_cell* cells,
                     int64_t capacity, int64_t* n_cells) {
    if (!c || !s || !n_cells) return fail(MVS_E_INVALID, "NULL argument");
    *n_cells = 0;
    if (capacity < 0 || !(jaccard_min > 0.0) || !(jaccard_min < 1.0)) return fail(MVS_E_INVALID, "bad argument");
    if (row_begin < 0 || row_end > s->n || row_begin > row_end || col_begin < 0 || col_end > s->n || col_begin > col_end)
        return fail(MVS_E_INVALID, "block outside the sketch set");
    if (row_begin == row_end || col_begin == col_end) return MVS_OK;
    if (!norms_sq || (capacity > 0 && !cells)) return fail(MVS_E_INVALID, "NULL buffer");
    HIP_TRY(hipSetDevice(c->device));
    int rc = ensure_buf(c, &c->pw_tmp, &c->pw_tmp_bytes, (size_t)std::max<int64_t>(capacity, 1) * sizeof(mvs_cell));
    if (rc) return rc;
    // J > j  <=>  (P/d) / (n2r + n2c - P/d) > j  <=>  double(P)/d > j/(1+j) * (n2r + n2c)   (for n2r + n2c > P/d >= 0)
    unsigned long long count = 0;
    rc = pairwise_launch(c, s, norms_sq, MVS_KEEP_INT16, row_begin, row_end, col_begin, col_end, false, false,
                         (mvs_cell*)c->pw_tmp, capacity, 0, &count, jaccard_min / (1.0 + jaccard_min));
    if (rc) return rc;
    if (count == ~0ULL) {
        {
            const int rb_rc = read_back(c, c->stream, {{&count, c->d_counter, 8}});
            if (rb_rc) return rb_rc;
        }
    }
    *n_cells = (int64_t)count;
    if ((int64_t)count > capacity)
        return fail(MVS_E_CAPACITY, "%llu hits but capacity is %lld", count, (long long)capacity);
    if (count == 0) return MVS_OK;
    rc = sort_on_device(c, (mvs_cell*)c->pw_tmp, (int64_t)count, cells);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(c->stream));   // documented synchronous: `cells` is final on return
    return MVS_OK;
}

int mvs_cells_sort(mvs_ctx* c, const mvs_cell* cells_in, int64_t n, mvs_cell* cells_out) {
    if (!c) return fail(MVS_E_INVALID, "ctx is NULL");
    if (n < 0) return fail(MVS_E_INVALID, "bad argument");
    if (n == 0) return MVS_OK;
    if (!cells_in || !cells_out || cells_in == cells_out) return fail(MVS_E_INVALID, "need two distinct device buffers");
    HIP_TRY(hipSetDevice(c->device));
    return sort_on_device(c, const_cast<mvs_cell*>(cells_in), n, cells_out);
}

int mvs_pairwise_dots(mvs_ctx* c, const mvs_sketch_set* s, int64_t r0, int64_t r1, int64_t c0, int64_t c1,
                      int32_t* out, int mem_out, int algo) {
    if (!c || !s) return fail(MVS_E_INVALID, "NULL argument");
    if (!mem_ok(mem_out) || r0 < 0 || r1 > s->n || r0 > r1 || c0 < 0 || c1 > s->n || c0 > c1 ||
        (algo != 0 && algo != 1))
        return fail(MVS_E_INVALID, "bad argument");
    if (r0 == r1 || c0 == c1) return MVS_OK;
    if (!out) return fail(MVS_E_INVALID, "out is NULL");
    HIP_TRY(hipSetDevice(c->device));
    const size_t bytes = (size_t)(r1 - r0) * (size_t)(c1 - c0) * 4;
    DevBuf dout;
    int32_t* d_out = out;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(dout.alloc(bytes));
        d_out = (int32_t*)dout.p;
    }
    mvs::PairwiseArgs a{};
    a.planes = s->planes;
    a.n = s->n;
    a.n_alloc = s->n_alloc;
    a.d = s->d;
    a.d_pad = s->d_pad;
    a.limbs = s->limbs;
    a.row_begin = r0;
    a.row_end = r1;
    a.col_begin = c0;
    a.col_end = c1;
    a.dots = d_out;
    int rc = mvs::launch_pairwise(c->stream, a, 1, algo, c->opt);
    if (rc) return fail(rc, "pairwise launch rejected");
    rc = check_kernel("k_pairwise(dots)");
    if (rc) return rc;
    if (mem_out == MVS_MEM_HOST) {
        HIP_TRY(hipMemcpyAsync(out, d_out, bytes, hipMemcpyDeviceToHost, c->stream));
        HIP_TRY(hipStreamSynchronize(c->stream));
    }
    return MVS_OK;
}

int64_t mvs_chunk_size(double max_memory_gb, int d) {
    const int64_t bytes_per_vector = (int64_t)d * 4;
    const int64_t max_bytes = (int64_t)(max_memory_gb * 1024 * 1024 * 1024);
    return bytes_per_vector > 0 ? max_bytes / (bytes_per_vector * bytes_per_vector) : 0;
}

void mvs_shard_rows(int64_t n, int num_shards, int shard_idx, int64_t* begin, int64_t* end) {
    if (num_shards < 1) num_shards = 1;
    const int64_t rps = (n + num_shards - 1) / num_shards;
    int64_t b = (int64_t)shard_idx * rps;
    int64_t e = std::min(b + rps, n);
    if (b > n) b = n;
    if (e < b) e = b;
    if (begin) *begin = b;
    if (end) *end = e;
}

}  // extern "C"
