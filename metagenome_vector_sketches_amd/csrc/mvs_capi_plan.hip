// mvs_capi_plan.hip -- C ABI: block plans (one rank's share of the symmetric multi-rank schedule) and the kept cells of a shard
#include "mvs_capi_internal.h"

using namespace mvs_capi;

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

void mvs_capi::plan_state_free(mvs_ctx* c) {
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

namespace mvs_capi {

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
    // ONE launch over the rows on both sides of the frame (its 16-row groups are skipped inside the kernel)
    const int64_t lo_end = st.f0 & ~(int64_t)15, hi_begin = std::min<int64_t>((st.f1 + 15) & ~(int64_t)15, s->n_alloc);
    const int64_t hi_end = std::min<int64_t>((s->n + 15) & ~(int64_t)15, s->n_alloc);
    const int64_t count = std::max(hi_end, lo_end);                                             // rows [0, count) but [lo_end, hi_begin)
    const int64_t skip_first = lo_end, skip_count = std::max<int64_t>(0, std::min(hi_begin, count) - lo_end);
    mvs::launch_planes_from_wire(c->stream, st.lo_wire, s->ext_coarse_fm, s->ext_rows, count, s->d_pad, const_cast<int8_t*>(s->planes),
                                 (const unsigned char*)c->pw_need, skip_first, skip_count);
    return check_kernel("k_planes_from_wire(needed rows)");
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
    if (c->opt.plan_order != 0) {
        const unsigned* order = nullptr;
        unsigned per = 0;
        const int ro = tile_order_for(c, a, segs, &order, &per);
        if (ro) return ro;
        segs.order = order;
        segs.order_per = per;
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

}  // namespace mvs_capi

extern "C" {

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

}  // extern "C"

namespace mvs_capi {
// planes[(row * 2 + 0) * d_pad + k] -> lo_wire[row * d_pad + k], 16 bytes per lane
__global__ void k_wire_rows(const int4* __restrict__ planes, int4* __restrict__ lo, long long row_first, long long rows, int vec_per_row) {
    const long long total = rows * vec_per_row;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const long long r = row_first + i / vec_per_row;
        const int v = (int)(i % vec_per_row);
        lo[r * vec_per_row + v] = planes[r * 2 * vec_per_row + v];
    }
}
}  // namespace mvs_capi

extern "C" {

int mvs_sketch_set_wire_rows(mvs_ctx* c, const mvs_sketch_set* s, int8_t* lo_wire, int64_t row_first, int64_t row_count) {
    if (!c || !s || !lo_wire) return fail(MVS_E_INVALID, "NULL argument");
    if (s->limbs != 2) return fail(MVS_E_INVALID, "the low-limb wire format is defined for two-limb sets");
    if (row_first < 0 || row_count < 0 || row_first + row_count > s->n_alloc) return fail(MVS_E_INVALID, "rows outside the set");
    if (row_count == 0) return MVS_OK;
    HIP_TRY(hipSetDevice(c->device));
    const int vec = s->d_pad / 16;
    const long long total = (long long)row_count * vec;
    const unsigned blocks = (unsigned)std::min<long long>(4096, (total + 255) / 256);
    hipLaunchKernelGGL(k_wire_rows, dim3(blocks), dim3(256), 0, c->stream, reinterpret_cast<const int4*>(s->planes),
                       reinterpret_cast<int4*>(lo_wire), (long long)row_first, (long long)row_count, vec);
    return check_kernel("k_wire_rows");
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


}  // extern "C"
