// mvs_capi_compare.hip -- C ABI: data derived from a set, the two-stage comparison stage by stage, mvs_pairwise_rows / _block,
// mvs_search_block, kept-cell sort, dense dots (include/mvs_hip.h)
#include "mvs_capi_internal.h"

using namespace mvs_capi;

extern "C" {

}  // extern "C"

namespace mvs_capi {

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
int tile_order_for(mvs_ctx* c, const mvs::PairwiseArgs& a, const mvs::PlanSegs& segs, const unsigned** d, unsigned* per) {
    *d = nullptr;
    *per = 0;
    std::vector<long long> key{segs.n, a.symmetric, a.sym_begin, a.sym_end, a.map_mode};
    for (int k = 0; k < segs.n; ++k) {
        key.push_back(segs.n_tr[k]);
        key.push_back(segs.n_tc[k]);
        key.push_back(segs.i_begin[k]);
        key.push_back(segs.j_begin[k]);
    }
    const mvs_tile_order* hit = nullptr;
    for (const mvs_tile_order& o : c->tile_orders)
        if (o.key == key) hit = &o;
    if (!hit) {
        if (c->tile_orders.size() >= 64) {                         // (other shapes every time: start over)
            HIP_TRY(hipDeviceSynchronize());
            for (mvs_tile_order& o : c->tile_orders)
                if (o.d) (void)hipFree(o.d);
            c->tile_orders.clear();
        }
        mvs_tile_order o;
        o.key = key;
        std::vector<unsigned> list;
        if (mvs::plan_tile_order(a, segs, &list, &o.per)) {
            if (hipMalloc((void**)&o.d, list.size() * 4) != hipSuccess) return fail(MVS_E_NOMEM, "hipMalloc of a tile order failed");
            const hipError_t e = hipMemcpy(o.d, list.data(), list.size() * 4, hipMemcpyHostToDevice);
            if (e != hipSuccess) {
                (void)hipFree(o.d);
                return fail(MVS_E_HIP, "uploading a tile order: %s", hipGetErrorString(e));
            }
        }
        c->tile_orders.push_back(std::move(o));
        hit = &c->tile_orders.back();
    }
    *d = hit->d;
    *per = hit->per;
    return MVS_OK;
}

void fill_args(mvs_ctx* c, const mvs_sketch_set* s, const double* d_n2, int keep_mode, int64_t rb, int64_t re, int64_t cb,
               int64_t ce, bool symmetric, bool mirror_all, double keep_coeff, mvs::PairwiseArgs& a, const mvs::Options* o) {
    const mvs::Options& opt = o ? *o : c->opt;     // (a caller that forces a variant passes its own copy: the context is not written)
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
    a.debug_flags = opt.pairwise_debug;
    a.map_mode = opt.pairwise_map;
    a.stamps = nullptr;
    a.symmetric = (symmetric && opt.pairwise_symmetric) ? 1 : 0;   // the launcher checks the alignment
}

// the running cell count starts at `start` (appending calls); kKeepCount: it stays what the device counter holds (a block
// plan appends block after block without the host ever learning the count in between)
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
                     bool hold_all, unsigned long long start, mvs::PairwiseArgs& a, TwoStage& ts, const mvs::Options* o) {
    const mvs::Options& opt = o ? *o : c->opt;     // (a caller that forces a variant passes its own copy: the context is not written)
    ts.opt = opt;
    const double block_cells = (double)(a.row_end - a.row_begin) * (double)(a.col_end - a.col_begin);
    int rc = prepare_coarse(c, s);
    if (rc) return rc;
    rc = ensure_buf(c, &c->pw_fmeta, &c->pw_fmeta_bytes, (size_t)s->n_alloc * sizeof(float4));
    if (rc) return rc;
    mvs::launch_filter_meta(c->stream, (const mvs::CoarseRow*)c->pw_rows, d_n2, s->n, s->n_alloc, s->d, keep_coeff,
                            (float4*)c->pw_fmeta);
    rc = check_kernel("k_filter_meta");
    if (rc) return rc;
    const bool forced = opt.pairwise_filter == 2;
    ts.tiles = mvs::filter_flags_tiles(a, opt);
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
        if (ts.tiles) cand_want = (int64_t)std::min(268435456.0, std::max(1048576.0, tiles_to_do * 8.0 * (double)opt.tile_dense_thr));
        else cand_want = (int64_t)limit;
    }
    rc = ensure_buf(c, &c->pw_cand, &c->pw_cand_bytes, (size_t)cand_want * sizeof(int2));
    if (rc) return rc;
    a.coarse = (const int8_t*)c->pw_coarse;
    a.coarse_fm = nullptr;
    if (opt.fragment_major && mvs::filter_streams(a, opt)) {
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
    a.recheck_mode = opt.recheck_mode;
    const int64_t n_regions = mvs::filter_region_count(a, opt);
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
        a.tile_dense_thr = (unsigned)opt.tile_dense_thr;
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
        {
            // the one-block launch of the default ping-pong filter with a balanced tile order (option plan_order, as the plans)
            const unsigned* order = nullptr;
            unsigned order_per = 0;
            mvs::PairwiseArgs kb{};
            mvs::PlanSegs segs{};
            if (opt.plan_order != 0 && mvs::filter_order_geometry(a, opt, &kb, &segs)) {
                rc = tile_order_for(c, kb, segs, &order, &order_per);
                if (rc) return rc;
            }
            rc = mvs::launch_filter(c->stream, a, opt, order, order_per);
        }
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
    int rc = mvs::launch_exact_pairs(c->stream, ts.a, ts.opt);
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
        int rc = mvs::launch_exact_tiles(c->stream, ts.a, ts.d_list + first, count, ts.opt);
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
                       bool symmetric, const mvs::Options* o) {
    const mvs::Options& opt = o ? *o : c->opt;     // (a caller that forces a variant passes its own copy: the context is not written)
    const int filter_mode = opt.pairwise_filter;
    const double block_cells = (double)(re - rb) * (double)(ce - cb);
    // A few rows against everything (a search with a handful of queries; one of very many shards) on a set whose coarse
    // plane does not exist yet: building the plane reads all the limb planes once, which is all the exact kernel needs for
    // such a block -- so the FIRST block of fewer than 1024 rows on a set goes to the exact kernel, and only when a second
    // one follows on the same set (a caller that keeps the set for many such blocks: pairwise_comp_optimized --shard_idx -1
    // with small shards, repeated searches) is the plane built.  Up to 16 rows the exact path is a streaming kernel that
    // runs at HBM speed (k_pairwise_skinny): nothing to filter for.
    const bool coarse_cached = c->coarse_id == s->id && c->coarse_gen == s->gen && c->coarse_mode == opt.coarse_radix;
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
    probe.symmetric = (symmetric && opt.pairwise_symmetric) ? 1 : 0;
    const bool streams = mvs::filter_streams_rows(probe, opt);
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
                    unsigned long long start, unsigned long long* count, double keep_coeff, const PackedOut* po,
                    const DenseOut* dn, const mvs::Options* o) {
    const mvs::Options& opt = o ? *o : c->opt;     // (a caller that forces a variant passes its own copy: the context is not written)
    // *count: the cell count if this call already had to synchronise for it, ~0 otherwise (read d_counter[0])
    *count = ~0ULL;
    mvs::PairwiseArgs a{};
    fill_args(c, s, d_n2, keep_mode, rb, re, cb, ce, symmetric, mirror_all, keep_coeff, a, o);
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
    if (opt.pairwise_debug & 8) {
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
    if (!dn && two_stage_applies(c, s, rb, re, cb, ce, keep_coeff, symmetric, o)) {
        TwoStage ts;
        rc = two_stage_filter(c, s, d_n2, keep_coeff, capacity, po != nullptr, start, a, ts, o);
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
    rc = attach_planes_fm(c, s, a, mvs::exact_reads_fm(a, opt));
    if (rc) return rc;
    if (c->timing) HIP_TRY(hipEventRecord(c->ev[2], c->stream));
    rc = mvs::launch_pairwise(c->stream, a, 0, 0, opt);
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

}  // namespace mvs_capi

extern "C" {

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
