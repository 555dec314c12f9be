/*
 * mvs_hip.h -- C ABI of libmvs_hip.so: the MI355X (gfx950) implementation of the reference's
 * random-projection sketching + all-vs-all sketch comparison hot path.
 *
 * The reference (RolandFaure/metagenome_vector_sketches) has no FFI seam on this path: the
 * functions below are what its two executables would bind if the hot loops were moved behind one.
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 * INTEGRATION.md shows the call sites a maintainer would change.
 *
 * Conventions
 *   - plain C types only; the caller owns every buffer it passes in; the library never frees
 *     caller memory and never keeps a caller pointer after the call returns (except the non-owning
 *     views documented at mvs_sketch_set_from_planes);
 *   - every function returns MVS_OK (0) or an MVS_E_* code; mvs_last_error() gives the message of the
 *     last failure on the calling thread; no C++ exception crosses this boundary;
 *   - `mem` arguments say where a buffer lives: MVS_MEM_HOST (pageable or pinned host memory) or
 *     MVS_MEM_DEVICE (HBM of the context's device);
 *   - all work of a context is issued on ONE HIP stream (its own, or the one given to
 *     mvs_ctx_set_stream); calls with only device buffers are asynchronous on that stream unless
 *     stated otherwise; a context is thread-compatible, not thread-safe;
 *   - there is no CPU fallback: if no gfx950 device is usable every call fails with MVS_E_HIP.
 */
#ifndef MVS_HIP_H
#define MVS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MVS_OK           0
#define MVS_E_INVALID    1   /* bad argument */
#define MVS_E_HIP        2   /* HIP runtime / device failure */
#define MVS_E_CAPACITY   3   /* output buffer too small; the needed size is reported */
#define MVS_E_NOMEM      4   /* host or device allocation failed */
#define MVS_E_RANGE      5   /* input outside the supported numeric range */
#define MVS_E_ABORTED    6   /* a caller's callback asked to stop */

#define MVS_MEM_HOST     0
#define MVS_MEM_DEVICE   1

/* limb codes (the `limbs` arguments below): 1..4 = that many signed base-256 limb planes;
 * MVS_LIMBS_K3 = three planes (l0, l1, l0+l1) of signed base-128 digits, |v| <= 8127: the comparison then
 * needs 3 matrix-core passes per cell (Karatsuba) instead of the 4 of two base-256 limbs. */
#define MVS_LIMBS_K3     0x103

/* keep test variants */
#define MVS_KEEP_INT32   0   /* src/pairwise_comp_optimized.cpp:139-141  (int64 dot / d truncates) */
#define MVS_KEEP_INT16   1   /* src/pairwise_comp_optimized_16bits.cpp:211-218 (double(dot)/d)    */

typedef struct mvs_ctx mvs_ctx;
typedef struct mvs_sketch_set mvs_sketch_set;

/* One kept cell of the comparison matrix: what the reference appends to `all_results`
 * (src/pairwise_comp_optimized.cpp:974-980) plus the quantised Jaccard its writer derives from it
 * (:658-665). */
typedef struct {
    int32_t row;   /* global sample index of the row                              */
    int32_t col;   /* global sample index of the column                           */
    int32_t dot;   /* int32 dot product, wrapped mod 2^32 (MatrixXi product, :135) */
    int32_t q;     /* uint16_t(round(min(J,1)*255)), J = (dot/d)/(n2r+n2c-dot/d)   */
} mvs_cell;

/* ---- library / context ------------------------------------------------------------------------ */
const char* mvs_version(void);
const char* mvs_last_error(void);
int mvs_device_count(int* count);

int mvs_ctx_create(int device, mvs_ctx** ctx);
int mvs_ctx_destroy(mvs_ctx* ctx);
/* Issue all later work on `hip_stream`, a hipStream_t of the same device (e.g. the caller's framework
 * stream); NULL is HIP's default ("null") stream.  mvs_ctx_use_own_stream switches back to the
 * non-blocking stream the context created for itself (the initial state). */
int mvs_ctx_set_stream(mvs_ctx* ctx, void* hip_stream);
int mvs_ctx_use_own_stream(mvs_ctx* ctx);
int mvs_ctx_synchronize(mvs_ctx* ctx);
/* Optional kernel timing: when enabled, HIP events are recorded on the context's stream around the
 * dominant kernel of mvs_project_csr (which = 0) and around the comparison kernels of mvs_pairwise_rows /
 * _block / mvs_search_block (which = 1: the whole comparison -- filter + re-check, or the exact kernel;
 * which = 2: the filter kernel alone; which = 3: the re-check kernel with the list passes in front of it; which = 4:
 * the exact kernel on the tiles the filter flagged as dense; 2 and 3 are only recorded by a two-stage comparison, 4 only
 * when it flagged tiles).  mvs_ctx_kernel_ms returns the duration of the most recent such launch in ms. */
int mvs_ctx_set_timing(mvs_ctx* ctx, int enabled);
int mvs_ctx_kernel_ms(mvs_ctx* ctx, int which, float* ms);
/* Tuning / diagnostic options of a context, by name.  The library reads the environment exactly once per
 * context, in mvs_ctx_create: MVS_<NAME> (upper case) gives an option's initial value; afterwards only
 * mvs_ctx_set_option changes it, so two contexts (or two threads with a context each) never interfere.
 *   pairwise_filter       0 exact kernel on every cell; 1 (default) two-stage comparison for blocks of at least
 *                         2^22 cells and more than 16 rows of a two-limb set (fewer than 1024 rows: from the second
 *                         such block on the same set on, or if the set's coarse plane is already cached); 2 two-stage
 *                         whenever the set has two limbs (tests)
 *   filter_variant        kernel of the one-pass filter: -1 (default) by block size, 8 = ping-pong wave groups on
 *                         256 x 256 tiles (7/9/10 its variants), 0 = 128 x 128 ring, 1 = 256 x 256 ring, 3/5/6 other rings
 *   exact_variant         re-check kernel: 3 (default) tree reduction over rounds of 64 pairs, 0 one shuffle butterfly per
 *                         pair, 1 quarter wave per pair, 2 16 pairs per round
 *   pairwise_variant      exact kernel: 8 (default) ping-pong 16x16x64 MFMA for two limbs (7/9 its variants), 6 the ring
 *                         kernel on the same shape, 0-5 32x32x32 tile / ring variants.  With the default, a block of up
 *                         to 16 rows x at least 1024 columns of a two-limb set (a search with a few queries) goes to a
 *                         streaming vector-ALU kernel instead (rows in LDS, one wave per column, v_dot2_i32_i16)
 *   pairwise_symmetric    1 (default) skip tiles below the diagonal and mirror; 0 compute every tile
 *   pairwise_block_cells  row-chunk bound of mvs_pairwise_rows, in cells (default 2^40)
 *   sort                  kept-cell sort: 0 (default) by list length, 1 merge sort, 2 radix sort
 *   enable_k3             1: mvs_sketch_set_create codes sets with 127 < max|v| <= 8127 as MVS_LIMBS_K3
 *   project_variant       projection kernel: 0 (default) by dimension -- 14 = four 64-dim blocks per wave sharing the
 *                         first splitmix64 round when d is a multiple of 256 (>= 512), else 2 or 1 blocks per wave;
 *                         1 / 2 / 12 / 14 force a variant
 *   stream_dense          mvs_pairwise_stream, dense results: 1 (default) one byte per cell in a matrix, a row block turned
 *                         into CSR / encoded rows on a side stream beside the next block's launch where the exact kernel
 *                         does whole row blocks, on the context's stream where the two-stage comparison feeds the matrix;
 *                         2 always the context's stream; 3 always the side stream; 0 packed 64-bit cells + sort
 *   tile_dense_thr        two-stage comparison on large blocks: a filter wave (128 x 64 cells) with more candidates than this
 *                         flags its 256 x 256 tile for the exact kernel instead of listing them (default 64; 0 = list
 *                         everything and give the whole block to the exact kernel once the list passes 1/128 of its cells)
 *   stream_list_cells, stream_pipeline
 *                         mvs_pairwise_stream with the two-stage comparison: up to stream_list_cells kept cells (bound from
 *                         the filter pass) leave as one sorted list, more through the dense byte matrix; stream_pipeline = 0
 *                         always filters the whole row range in one pass first (tests)
 *   fragment_major        1 (default): the matrix-core kernels that move 16 samples x 64 k values per instruction (ping-pong
 *                         filter and exact kernel, streaming search filter) read fragment-major copies of the coarse plane /
 *                         limb planes, built once per resident set (+1 x the planes' bytes); 0: the row-major planes
 *   pairwise_bdirect      1 (default): with those copies, the ping-pong kernels load the B operand's fragments straight into
 *                         registers and LDS carries the A operand only; 0: both operands through LDS
 *   search_stream         1 (default): blocks of up to 640 rows x at least 4096 columns outside the symmetric schedule (a
 *                         search) take the streaming filter (rows resident in LDS, columns streamed); 0: the tile kernels
 *   plan_order            1 (default) the launches of the ping-pong filter -- of a block plan, and the one-block launch of
 *                         mvs_pairwise_rows / _stream -- take a balanced tile order: every XCD the same number of tiles of every
 *                         16 x 16-tile super-patch; 0 the static super-patch map (a triangle's static sub-patches hold 32, 26, 10
 *                         or 0 tiles: launches of a few rounds waited for their fullest XCD)
 *   stream_block_rows, encode_stage_words, pairwise_map, coarse_radix, cand_regions, recheck_mode, recheck_blocks
 *                         test / experiment switches (DESIGN.md, appendix "switches"; encode_stage_words below 64 also keeps
 *                         every row on the device encoder's general loop)
 *   comm_timeout_s        file transport (mvs_comm_create_files / _rendezvous): seconds a rank waits for its peers
 *   markers               1: roctx ranges named after the entry points around mvs_project_csr / mvs_pairwise_rows /
 *                         mvs_pairwise_block (rocprofv3 --marker-trace); libroctx64 is bound at run time
 *   pairwise_debug        profiling aids, bit mask: 1 / 2 skip the k-loop / the epilogue (such runs produce garbage by
 *                         design), 8 per-workgroup time stamps of the comparison kernel dumped to /tmp/mvs_stamps.bin;
 *                         rejected unless the library was built with -DMVS_ABLATIONS (make -C csrc ablations)
 * Unknown names and out-of-range values return MVS_E_INVALID.  None of them changes a result. */
int mvs_ctx_set_option(mvs_ctx* ctx, const char* name, int64_t value);
int mvs_ctx_get_option(const mvs_ctx* ctx, const char* name, int64_t* value);

/* Diagnostics of the most recent comparison (mvs_pairwise_rows / _block / mvs_search_block): the number
 * of candidate pairs its coarse filter passed on to the exact re-check, 0 if the exact kernel ran on
 * every cell (see mvs_pairwise_rows). */
int mvs_ctx_pairwise_candidates(mvs_ctx* ctx, int64_t* candidates);
/* The same with the tile-granular part of the two-stage comparison: `flagged_tiles` of the `filter_tiles` 256 x 256 tiles the
 * filter pass worked on held so many candidates that the exact kernel computed them whole instead (dense regions of the
 * result; the reference's cost is flat in the density, src/pairwise_comp_optimized.cpp:135-147 -- this keeps ours from
 * falling off a cliff between "sparse" and "dense").  Any pointer may be NULL. */
int mvs_ctx_pairwise_stats(mvs_ctx* ctx, int64_t* candidates, int64_t* flagged_tiles, int64_t* filter_tiles);

/* ---- device memory and events for hosts above the ABI -------------------------------------------------
 * The host drivers are plain C++ above this header (no HIP runtime of their own): the buffers a multi-rank step keeps between
 * its calls (limb planes, coarse plane, statistics, kept cells -- INTEGRATION.md B5) and the ordering between the context
 * that compares and the context the communicator exchanges on come from here.  Replaces nothing of the reference by
 * itself: its tiles live in Eigen matrices on the host (src/pairwise_comp_optimized.cpp:33-54).
 * mvs_device_alloc : bytes of HBM on the context's device (zero != 0: cleared, on the context's stream).
 * mvs_device_free  : waits for the context's stream first.
 * mvs_device_zero  : asynchronous on the context's stream.
 * mvs_device_copy  : dst / src each MVS_MEM_HOST or MVS_MEM_DEVICE; device-to-device copies are asynchronous on the
 *                    context's stream, copies that touch host memory return when the host buffer may be reused / read.
 * Events order two contexts of one device without the host waiting: mvs_event_record puts the event on the context's
 * stream, mvs_ctx_wait_event makes everything queued LATER on a context's stream wait for it (an event that was never
 * recorded counts as complete).  timing != 0 at creation: mvs_event_elapsed_ms between two recorded events. */
typedef struct mvs_event mvs_event;
int mvs_device_alloc(mvs_ctx* ctx, size_t bytes, int zero, void** ptr);
int mvs_device_free(mvs_ctx* ctx, void* ptr);
int mvs_device_zero(mvs_ctx* ctx, void* ptr, size_t bytes);
int mvs_device_copy(mvs_ctx* ctx, void* dst, int mem_dst, const void* src, int mem_src, size_t bytes);
int mvs_event_create(mvs_ctx* ctx, int timing, mvs_event** event);
int mvs_event_record(mvs_ctx* ctx, mvs_event* event);
int mvs_ctx_wait_event(mvs_ctx* ctx, mvs_event* event);
int mvs_event_synchronize(mvs_event* event);
int mvs_event_elapsed_ms(mvs_event* begin, mvs_event* end, float* ms);
int mvs_event_destroy(mvs_event* event);

/* ---- projection ----------------------------------------------------------------------------------
 * Replaces transform_set_into_vector() (src/random_projection.cpp:9-26) called once per sample from
 * the OpenMP loop of sketch() (src/project_everything.cpp:289-298) and from standalone_projection
 * (src/standalone_projection.cpp:37).
 *
 * hashes  : concatenated per-sample hash lists (CSR values), `mem_hashes` says where they live.
 *           Precondition: hashes of one sample are unique (the reference holds them in an
 *           std::unordered_set); order is irrelevant.
 * offsets : n_samples+1 HOST int64, offsets[s]..offsets[s+1] is sample s; a sample may be empty.
 *           Each sample must hold < 2^31 hashes.
 * out     : n_samples x d int32, row-major by sample (the vectors.bin record layout,
 *           src/project_everything.cpp:349-353), `mem_out` says where.
 * out[s][k] = sum over hashes h of (1 - 2*bit_{k%64}(splitmix64(h + 64*(k/64)))), exact.
 */
int mvs_project_csr(mvs_ctx* ctx, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                    int64_t n_samples, int d, int32_t* out, int mem_out);

/* mvs_project_csr plus the two statistics the next stages need, produced by the same kernel when every
 * sample holds <= 65536 hashes (otherwise by one extra pass): sumsq[s] = exact sum of squares of sketch s
 * (int64[n_samples], in the same memory space as `out`: device array for device sketches, host array for host
 * sketches) and *max_abs = largest |v| (host; decides the limb code).  Synchronous.
 * Host hash lists beyond 32 MiB travel through two pinned staging buffers: the host-side copy, the DMA and the
 * projection of the samples already complete overlap (measured: 1.07 x the bare PCIe time of the same bytes). */
int mvs_project_csr_stats(mvs_ctx* ctx, const uint64_t* hashes, int mem_hashes, const int64_t* offsets,
                          int64_t n_samples, int d, int32_t* out, int mem_out, int64_t* sumsq, int64_t* max_abs);

/* Sum of squares of each sketch (exact int64): the integer the norm of
 * src/project_everything.cpp:328-329 is derived from (norm = sqrt(sumsq / d)). */
int mvs_sketch_sumsq(mvs_ctx* ctx, const int32_t* sketches, int mem_in, int64_t n, int d,
                     int64_t* sumsq, int mem_out);

/* The vector_norms.txt round trip without the file, on the device: out[i] = strtod(text_i)^2 where text_i is the
 * "%g" (6 significant digits) rendering of sqrt(double(sumsq[i]) / d) -- i.e. exactly the squared norm that
 * src/pairwise_comp_optimized.cpp:893-901 parses from the line src/project_everything.cpp:328-330 writes (with this
 * build's norm definition).  Decimal rounding is done in exact arithmetic (ties included), so
 * the values are bit-identical to the ones read back from the file.  sumsq and out are device arrays of n entries;
 * asynchronous on the context's stream.  For a pipeline that sketches and compares in one process. */
int mvs_norms_sq_text(mvs_ctx* ctx, const int64_t* sumsq, int64_t n, int d, double* out);

/* Both statistics the stages after the projection need, in one pass over the sketches: the per-sketch sum
 * of squares (as mvs_sketch_sumsq) and the largest |v| of the whole array (as mvs_sketch_max_abs, which
 * decides the limb code).  Synchronous: *max_abs is valid on return. */
int mvs_sketch_stats(mvs_ctx* ctx, const int32_t* sketches, int mem_in, int64_t n, int d, int64_t* sumsq,
                     int mem_out, int64_t* max_abs);

/* Saturating int32 -> int16 store of the --int16 mode (src/project_everything.cpp:332-347). */
int mvs_sketch_saturate_i16(mvs_ctx* ctx, const int32_t* sketches, int mem_in, int64_t n_elems,
                            int16_t* out, int mem_out);

/* ---- pairwise comparison --------------------------------------------------------------------------
 * Replaces load_matrix_block() + compute_sparse_dot_products_optimized() + the tiling loop of
 * main() (src/pairwise_comp_optimized.cpp:33-54, :57-160, :949-982) and the per-cell arithmetic of
 * write_sparse_results_jaccard_wo_sort() (:658-665); for elem_bytes == 2 also
 * compute_sparse_dot_products_optimized_16() (src/pairwise_comp_optimized_16bits.cpp:96-244).
 *
 * The sketches are first re-coded into signed base-256 int8 "limb planes" kept in HBM (exact:
 * v == sum_a limb_a * 256^a mod 2^32); the comparison kernel multiplies limb planes on the int8
 * matrix cores with int32 accumulation and recombines them mod 2^32, so `dot` carries exactly the
 * bits of the reference's int32 Eigen product.
 */

/* Largest |v| over n*d sketch entries (elem_bytes 4: int32, 2: int16); synchronous. */
int mvs_sketch_max_abs(mvs_ctx* ctx, const void* sketches, int elem_bytes, int mem, int64_t n_elems,
                       int64_t* max_abs);
/* Limb code for entries up to max_abs: 1 (<=127), 2 (<=32639), 3, or 4.  (MVS_LIMBS_K3 is never chosen here: it
 * measures slower than 2 on MI355X; option enable_k3 makes mvs_sketch_set_create use it for 128..8127.) */
int mvs_limbs_for_max_abs(int64_t max_abs);
/* Geometry of the limb-plane buffer for n samples: rows are padded to n_alloc (multiple of 256,
 * plus one spare tile), the dimension to d_pad (multiple of 128); layout is
 * planes[(row * P + plane) * d_pad + k] with P = limbs & 0xff planes per row, zero filled outside n x d. */
int mvs_limb_geometry(int64_t n, int d, int limbs, int64_t* n_alloc, int* d_pad, size_t* bytes);
/* Re-code rows [0, n_rows) of `sketches` into rows [row_offset, row_offset + n_rows) of a
 * caller-owned DEVICE plane buffer that has been zero-filled (so several producers -- e.g. the
 * ranks of an all-gather -- can fill disjoint row ranges of one buffer). */
int mvs_limb_split(mvs_ctx* ctx, const void* sketches, int elem_bytes, int mem, int64_t n_rows, int d,
                   int limbs, int8_t* planes, int d_pad, int64_t row_offset);

/* A sketch set = limb planes of N samples resident in HBM.
 * mvs_sketch_set_create : allocates the planes, re-codes `sketches` (chooses the limb count itself).
 * mvs_sketch_set_from_planes : NON-owning view of a caller-owned device buffer filled by
 *   mvs_limb_split (and, across GPUs, by an all-gather of per-rank row blocks); the buffer must
 *   outlive the set and keep its contents while the set is compared (the library caches data derived
 *   from the planes per set: make a new view after rewriting the buffer). */
int mvs_sketch_set_create(mvs_ctx* ctx, const void* sketches, int elem_bytes, int mem, int64_t n, int d,
                          mvs_sketch_set** set);
int mvs_sketch_set_from_planes(mvs_ctx* ctx, const int8_t* planes, int64_t n, int64_t n_alloc, int d,
                               int d_pad, int limbs, mvs_sketch_set** set);
/* For databases streamed from disk in row chunks (load_matrix_block, src/pairwise_comp_optimized.cpp:33-54):
 * allocate zeroed planes for n samples with a given limb count, then fill row ranges. */
int mvs_sketch_set_alloc(mvs_ctx* ctx, int64_t n, int d, int limbs, mvs_sketch_set** set);
int mvs_sketch_set_fill(mvs_sketch_set* set, const void* sketches, int elem_bytes, int mem, int64_t row_offset,
                        int64_t n_rows);
/* mvs_sketch_set_fill that also reports the largest |v| of the rows it was given (same upload).  A loader can
 * then stream a database ONCE: allocate for two limbs, fill, and only if some chunk reports |v| beyond what the
 * set's limb count holds (mvs_limbs_for_max_abs) start over with more limbs. */
int mvs_sketch_set_fill_stats(mvs_sketch_set* set, const void* sketches, int elem_bytes, int mem,
                              int64_t row_offset, int64_t n_rows, int64_t* max_abs);
int mvs_sketch_set_info(const mvs_sketch_set* set, int64_t* n, int* d, int* limbs, int64_t* n_alloc,
                        int* d_pad);
/* The DEVICE plane buffer of a set the library allocated (mvs_sketch_set_alloc / _create), for collectives that fill
 * other ranks' row blocks in place (mvs_allgather_planes).  mvs_sketch_set_touch tells the library that the caller
 * has rewritten the planes (data derived from them is rebuilt on the next comparison). */
int mvs_sketch_set_planes(mvs_sketch_set* set, int8_t** planes);
int mvs_sketch_set_touch(mvs_sketch_set* set);
int mvs_sketch_set_destroy(mvs_sketch_set* set);

/* All-vs-all for the row range [row_begin, row_end) against ALL n columns -- one shard of
 * src/pairwise_comp_optimized.cpp:938-982.
 *   norms_sq : n doubles, (parsed norm)^2 as built at :893-901 (`mem_norms` says where): squares, so >= 0,
 *              or nan / inf if the text said so (such a sample keeps nothing, as in the reference);
 *              negative values are outside the contract.
 *   keep_mode: MVS_KEEP_INT32 or MVS_KEEP_INT16.
 *   cells    : capacity entries (`mem_cells`); on success holds *n_cells kept cells sorted by
 *              (row, col) -- the per-row, ascending-column order the reference's writer relies on
 *              (:718-722).  If more than `capacity` cells are kept the call returns MVS_E_CAPACITY
 *              and *n_cells is the number needed (retry with a larger buffer or fewer rows).
 * Synchronous (returns after the count is known).
 * How the cells are found is the library's business and never changes the result: sets of two base-256
 * limbs are compared in two stages -- a one-pass int8 filter on a coarse plane with a proven error bound
 * drops the pairs that cannot pass the keep test, the exact int32 dot and the reference's keep test run on
 * the survivors (rows whose sum of squares reaches 2^31, whose dots may wrap, are re-checked against every
 * column) -- everything else goes through the exact kernel cell by cell (option pairwise_filter = 0 forces
 * that, = 2 forces the two stages on small blocks too; mvs_ctx_pairwise_candidates reports which one ran). */
int mvs_pairwise_rows(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, int mem_norms,
                      int keep_mode, int64_t row_begin, int64_t row_end, mvs_cell* cells,
                      int64_t capacity, int mem_cells, int64_t* n_cells);

/* The same comparison with the result STREAMED out in row blocks instead of returned in one caller-sized buffer -- what a
 * shard writer wants (src/pairwise_comp_optimized.cpp:974-990 keeps all of `all_results` in RAM and then groups it by row;
 * its writer, :718-736, needs per row only the ascending columns and q).  Rows [row_begin,row_end) against all columns;
 * the kept cells come back as CSR pieces of consecutive whole rows, in ascending row order, each piece exactly once:
 *     row r of a piece (row_begin <= r < row_end) holds cells row_ptr[r - row_begin] .. row_ptr[r - row_begin + 1] of
 *     col[] (ascending) and q[]; q16 replaces q (which is then NULL) in the one case where a value does not fit 8 bits
 *     (a negative Jaccard estimate from a norms file that does not belong to the vectors).
 * The arrays live in pinned host memory of the library and are valid only during the callback.  The callback runs on a
 * worker thread of the library, one call at a time, while the device computes and downloads the next pieces; a
 * non-zero return stops the comparison (MVS_E_ABORTED).  Empty rows are covered too (pieces with n_cells 0 occur).
 * Nothing is ever computed twice for lack of space: the two-stage comparison sizes its output from the candidate count
 * between its stages; where the exact kernel runs, the rows are cut into blocks whose worst case fits
 * `device_budget_bytes` of HBM (0: a quarter of what is free), each using the symmetric schedule inside its own square.
 * *n_cells (optional) receives the number of kept cells delivered.  Synchronous. */
typedef struct {
    int64_t row_begin, row_end;
    int64_t n_cells;
    const int64_t* row_ptr;
    const int32_t* col;
    const uint8_t* q;
    const uint16_t* q16;
} mvs_row_block;
typedef int (*mvs_row_block_cb)(void* user, const mvs_row_block* block);
int mvs_pairwise_stream(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, int mem_norms, int keep_mode,
                        int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_row_block_cb cb, void* user,
                        int64_t* n_cells);

/* mvs_pairwise_stream with the rows already ENCODED on the device in this build's shard codec
 * (metagenome_vector_sketches_amd/csrc/host/mvs_codec.hpp; the reference's writer, src/pairwise_comp_optimized.cpp:718-736,
 * stores per row a compact_vector of the q values and, when the row holds more than one cell, a rice_sequence of the
 * column deltas -- through the `bits` library, which is absent from the reference tree, so the byte layout is this
 * build's own).  A piece covers rows [row_begin, row_end) and carries, for its n_rows rows that hold cells (ascending):
 * their ids, their first columns (what goes into neighbor_start.bin), the byte offset of each row's record inside
 * `bytes`, the size of its compact_vector (the "Jac space" statistic of :808) and the records themselves, back to back --
 * exactly the bytes the host writer appends to matrix.bin for the same rows (tests compare the files).  Same delivery
 * rules as mvs_pairwise_stream.  1.4 bytes per kept cell cross the link instead of 5, and no host thread touches a cell. */
typedef struct {
    int64_t row_begin, row_end;
    int64_t n_cells;
    int64_t n_rows;
    const uint32_t* rows;
    const uint32_t* first_col;
    const uint64_t* offset;
    const uint32_t* jac_bytes;
    const uint8_t* bytes;
    int64_t n_bytes;
} mvs_encoded_rows;
typedef int (*mvs_encoded_rows_cb)(void* user, const mvs_encoded_rows* piece);
int mvs_pairwise_stream_encoded(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, int mem_norms, int keep_mode,
                                int64_t row_begin, int64_t row_end, size_t device_budget_bytes, mvs_encoded_rows_cb cb,
                                void* user, int64_t* n_cells);

/* What the most recent mvs_pairwise_stream of the context did: time of its comparison kernels summed over the row blocks
 * (0 unless mvs_ctx_set_timing is on), bytes handed to the callback, row blocks computed, pieces delivered, and whether
 * the two-stage comparison produced them as one list (1), through the dense byte matrix after one filter pass over all rows,
 * its flagged tiles computed row block by row block (2), through the matrix with the filter itself running row block by row
 * block (3: results that look dense from the first tile row on), or the exact kernel alone in row blocks (0).  Any pointer
 * may be NULL. */
int mvs_ctx_stream_stats(const mvs_ctx* ctx, double* kernel_ms, int64_t* bytes_out, int64_t* row_blocks, int64_t* pieces,
                         int* two_stage);

/* One rectangular block of the comparison -- rows [row_begin,row_end) x columns [col_begin,col_end) -- for
 * schedules that split a shard's work by column block (metagenome_vector_sketches_amd/parallel.py: with G
 * shards every unordered pair of row blocks is compared ONCE and the mirrored cells are exchanged).
 *   flags : MVS_BLOCK_SYMMETRIC  the column range contains the square of the row range: inside it tiles
 *                                strictly below the diagonal are skipped and produced by mirroring (this is
 *                                what mvs_pairwise_rows does on its own);
 *           MVS_BLOCK_MIRROR_ALL every kept (row, col) is also appended as (col, row) -- the transposed block
 *                                belongs to another shard, which then does not compute it.
 *   norms_sq, cells : DEVICE buffers.  Cells are APPENDED, unsorted, at index *n_cells (in/out);
 *   MVS_E_CAPACITY reports the needed total in *n_cells.  Order them with mvs_cells_sort.  Synchronous. */
#define MVS_BLOCK_SYMMETRIC   1
#define MVS_BLOCK_MIRROR_ALL  2
int mvs_pairwise_block(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, int keep_mode,
                       int64_t row_begin, int64_t row_end, int64_t col_begin, int64_t col_end, int flags,
                       mvs_cell* cells, int64_t capacity, int64_t* n_cells);
/* Query-by-sketch search, the exact brute-force counterpart of the reference's FAISS IndexFlatIP path
 * (src/jaccard.py:63-224: normalised inner products, then jaccard = ip*qn*nn / (nn^2 + qn^2 - ip*qn*nn) > j).
 * Rows [row_begin,row_end) of the set are the queries (their sketches appended after the database rows),
 * columns [col_begin,col_end) the database; a pair is reported iff its Jaccard estimate
 * (dot/d) / (n2_row + n2_col - dot/d) exceeds jaccard_min (0 < jaccard_min < 1).  Same kernel as the
 * comparison, with the keep coefficient 0.05 replaced by jaccard_min/(1+jaccard_min) and the floating keep
 * test.  norms_sq and cells are DEVICE buffers; cells come back sorted by (row, col), `dot` exact, `q` the
 * 8-bit quantised estimate.  Synchronous. */
int mvs_search_block(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, double jaccard_min,
                     int64_t row_begin, int64_t row_end, int64_t col_begin, int64_t col_end, mvs_cell* cells,
                     int64_t capacity, int64_t* n_cells);

/* ---- block plans: one rank's share of the symmetric multi-rank schedule, compared in few launches ----------------
 * The reference shards by rows and lets every shard process compute its rows against ALL columns
 * (src/pairwise_comp_optimized.cpp:937-982); dot, keep test and quantised Jaccard are symmetric in (row, col), so with G
 * ranks every unordered pair of row blocks needs comparing only once (metagenome_vector_sketches_amd/parallel.py:
 * block_plan).  A rank's share -- its diagonal block under the symmetric schedule plus G/2 off-diagonal blocks whose kept
 * cells are mirrored -- is handed over as rectangles and runs as ONE two-stage comparison: the filter in as few launches
 * as the caller wants (one per group of rectangles whose columns have arrived: the blocks of a peer can be launched as soon
 * as that peer's rows have landed, while the rest of the exchange is still on the wire), then one re-check, one launch of
 * the exact kernel on the flagged tiles.  One host synchronisation, in mvs_plan_finish.
 *
 * STORAGE coordinates.  Plans work on a set whose rows are laid out in per-rank blocks of `block_rows_padded` rows
 * (mvs_shard_layout: the shard size ceil(N / G) of :938-940 rounded up to a multiple of 256, so that every block starts on
 * the tile grid and on the 16-row grid of the fragment-major planes; the rows behind a shard's last sample are zero
 * sketches with zero norms).  Kept cells come out in storage coordinates; mvs_cells_route turns them into sample indices
 * and drops anything that touches a padding row.
 *
 * mvs_sketch_set_attach_derived : the filter's inputs for this set -- the fragment-major coarse plane (n_alloc * d_pad
 *     bytes) and 16 bytes of statistics per row -- live in CALLER buffers from now on: the caller fills them for its own
 *     rows with mvs_sketch_set_prepare_rows and for the other ranks' rows by gathering those ranks' (mvs_allgather_rows on
 *     the coarse plane in units of 16 rows, mvs_allgather_bytes on the statistics), instead of every rank rebuilding them
 *     for all N rows from the gathered limb planes.  Two-limb sets only (others: the calls succeed and plans run the exact
 *     kernel block by block).
 * mvs_sketch_set_prepare_rows   : coarse plane + statistics of rows [row_first, row_first + row_count) from the limb planes
 *     (row_first and row_count multiples of 16).  Asynchronous.
 * mvs_plan_begin  : frame = the rank's own row block [frame_row_begin, frame_row_end) (multiples of 256) x all columns;
 *     the symmetric schedule applies inside the frame's square; MVS_PLAN_MIRROR_OUTSIDE mirrors every kept cell outside it.
 *     norms_sq and cells are DEVICE buffers (norms indexed by storage row); cells are appended from index 0, unsorted.
 * mvs_plan_filter : the filter over these rectangles (rows inside the frame; bounds on multiples of 256 or at the set's
 *     end) as one launch.  Asynchronous: the caller orders it behind the arrival of the columns it reads.
 * mvs_plan_finish : re-check + flagged tiles.  *d_count = DEVICE address of the running cell count (it may exceed the
 *     capacity: cells beyond it were dropped, compare after reading it back).  Synchronises once in the middle.
 *
 * Option plan_speculate = 1 (off by default; set it around mvs_plan_begin, where the decision is taken): a plan with the
 *     same frame, set geometry, keep mode and capacity as the previous plan of this context sizes its second half --
 *     candidate pruning, re-check, the launch on the flagged tiles -- from THAT plan's counts and mvs_plan_finish does not
 *     synchronise at all: the kernels read the real counts on the device, and a last kernel checks that the sizes held.  If
 *     they did not (more flagged tiles than twice the previous count + 64, a candidate list beyond its buffer), the cell
 *     count reads MVS_PLAN_STALE or more: discard the cells and run the same plan again -- it will not speculate.  The
 *     counts reach the host with the caller's next read-back (mvs_cells_report brings them along; mvs_plan_stats and the
 *     next mvs_plan_begin wait for them if nobody has).  A steady multi-rank step thus has ONE host synchronisation. */
#define MVS_PLAN_MIRROR_OUTSIDE 1
#define MVS_PLAN_STALE (1ULL << 62)
typedef struct {
    int64_t row_begin, row_end, col_begin, col_end;
} mvs_plan_block;
int mvs_shard_layout(int64_t n_total, int world, int64_t* block_rows, int64_t* block_rows_padded);
int mvs_sketch_set_attach_derived(mvs_sketch_set* set, int8_t* coarse_fm, void* row_stats);
int mvs_sketch_set_prepare_rows(mvs_ctx* ctx, mvs_sketch_set* set, int64_t row_first, int64_t row_count);
/* mvs_limb_split + mvs_sketch_set_prepare_rows in ONE pass over the sketches (k_recode_rows): rows [row_first, row_first +
 * n_rows) of the set's planes and derived data from `sketches` (DEVICE, n_rows x d, elem_bytes 4 or 2), the remaining rows up to
 * row_first + row_count as zero rows (their planes must be zero already).  WRITES the plane buffer the set was made from
 * (mvs_sketch_set_from_planes).  Two-limb sets of d_pad <= 4096 with derived data attached; anything else takes the separate
 * passes, same result.  Asynchronous. */
int mvs_sketch_set_recode_rows(mvs_ctx* ctx, mvs_sketch_set* set, const void* sketches, int elem_bytes, int64_t n_rows,
                               int64_t row_first, int64_t row_count);
/* The exchange without the high limb.  The coarse plane c, its radix m (in the row statistics) and the LOW limb of a row pin
 * the row's values (v is the one value congruent to the low limb mod 256 near m * c: the radix search only admits radices for
 * which that holds, csrc/mvs_pairwise.hip "The high limb on the wire"), so ranks exchange 2 bytes per entry -- coarse plane
 * and low limbs -- instead of 3 and rebuild the other ranks' limb planes here: rows [row_first, row_first + row_count)
 * (multiples of 16) of the set's plane buffer from lo_wire[row * d_pad + k] (DEVICE; indexed like the set's rows) and the
 * attached coarse plane / statistics of those rows.  Valid for rows whose radix is <= MVS_WIRE_RADIX_MAX, i.e. max|v| <=
 * MVS_WIRE_MAX_ABS: the caller checks the largest |v| of all ranks (it travels in the cells header) and falls back to
 * exchanging the limb planes otherwise.  Asynchronous. */
#define MVS_WIRE_RADIX_MAX 252
#define MVS_WIRE_MAX_ABS 32004
int mvs_sketch_set_planes_from_wire(mvs_ctx* ctx, mvs_sketch_set* set, const int8_t* lo_wire, int64_t row_first, int64_t row_count);
/* The sender's side of that exchange: the LOW limbs of rows [row_first, row_first + row_count) of a two-limb set's planes into
 * lo_wire[row * d_pad + k] (DEVICE, indexed like the set's rows) -- a rank calls it for its own row block before the all-gather
 * of the wire buffer.  Asynchronous. */
int mvs_sketch_set_wire_rows(mvs_ctx* ctx, const mvs_sketch_set* set, int8_t* lo_wire, int64_t row_first, int64_t row_count);
/* The same for a plan, ROW BY ROW AS NEEDED: after mvs_plan_wire, mvs_plan_finish rebuilds -- between gathering the candidates
 * and the re-check -- exactly the rows outside the frame that its second half reads (the columns of the candidates and of the
 * flagged tiles; lo_wire as above, valid until the plan has finished).  A plan whose filter leaves few candidates touches few
 * rows; the others' limb planes keep whatever an earlier step left there.  Plans with a filter only (others read every row of
 * their blocks: mvs_sketch_set_planes_from_wire before mvs_plan_filter). */
int mvs_plan_wire(mvs_ctx* ctx, const int8_t* lo_wire);
/* The caller announces that the row statistics (mvs_sketch_set_attach_derived) and squared norms of rows [row_begin, row_end) are
 * in place on the context's stream -- in a multi-rank step: the small exchange has been joined.  The plan derives the filter's
 * per-row constants of those rows in ONE launch (the part of the range inside the frame is skipped: derived at mvs_plan_begin),
 * and mvs_plan_filter no longer derives them block by block (eight 4-us launches per step of an 8-way split).  Optional: blocks
 * whose columns were never announced are handled as before.  Replaces nothing of the reference by itself -- part of the tile
 * loop's set-up (src/pairwise_comp_optimized.cpp:949-958: norms of the column tile). */
int mvs_plan_rows_ready(mvs_ctx* ctx, int64_t row_begin, int64_t row_end);
int mvs_plan_begin(mvs_ctx* ctx, const mvs_sketch_set* set, const double* norms_sq, int keep_mode, int64_t frame_row_begin,
                   int64_t frame_row_end, int flags, mvs_cell* cells, int64_t capacity);
int mvs_plan_filter(mvs_ctx* ctx, const mvs_plan_block* blocks, int n_blocks);
int mvs_plan_finish(mvs_ctx* ctx, const uint64_t** d_count);
/* What the last plan did.  ms (timing enabled, else zeros): [0] the time during which a filter launch of the plan was running
 * (with option plan_overlap = 1 consecutive launches alternate between the context's stream and a side stream and overlap), [1] re-check (candidate gather,
 * pruning, k_exact_pairs), [2] exact kernel on the flagged tiles, [3] first filter launch .. end of the plan on the stream.
 * counts: [0] candidates, [1] flagged tiles, [2] 256 x 256 filter tiles computed, [3] filter launches, [4] bit 0: the plan ran
 * the exact kernel block by block (no filter), bit 1: it ran ahead of its read-backs (plan_speculate), bit 2: and its sizes
 * did not hold (MVS_PLAN_STALE), [5] d_pad.  Call after the stream has drained. */
int mvs_plan_stats(mvs_ctx* ctx, double ms[4], int64_t counts[6]);

/* Kept cells of a plan -> this rank's shard.  A cell of the plan is in storage coordinates and, under the symmetric
 * multi-rank schedule, about half of them are mirror images that belong to OTHER ranks' rows.
 * d_own_count (DEVICE) is the shard's state block of 16 + 4 * (own_end - own_begin + 1) bytes: u64 cells of own rows, u32 largest
 *     number of cells in one row (valid after mvs_cells_report), u32 unused, then one u32 per own row: its cells.
 * mvs_cells_route   : cells [0, min(*d_n_raw, raw_capacity)) -> sample indices (row = block * block_rows + offset for offset <
 *     block_rows; cells touching a padding row or a row >= n_total are dropped); those of rows [own_begin, own_end) are
 *     appended to own_out (state block zeroed by this call, then counted into), the others to `send` = a 64-byte header
 *     {foreign cells (may exceed the capacity), status, max_abs, *d_n_raw, raw_capacity, ...} followed by foreign_capacity
 *     cells.  status / max_abs travel in the header so that one exchange tells every rank how every other rank fared.
 * mvs_cells_collect : `recv` = world such buffers (after an all-gather of the send buffers): the cells of rows
 *     [own_begin, own_end) in the other ranks' buffers are appended to own_out.
 * mvs_cells_report  : read back -- out[0] = cells of own rows, then per rank {foreign cells, status, max_abs, raw cells,
 *     raw capacity} (5 * world values), then out[1 + 5 * world] = the largest number of cells in one own row.  Synchronous; the
 *     others are asynchronous.
 * mvs_cells_sort_rows : own_out -> (row, col) order by row buckets -- the rows' counts are in the state block: scan, scatter
 *     into the rows' segments, one wave per row orders its cells by column.  For shards whose rows hold at most 64 cells (what
 *     the report says); anything else: mvs_cells_sort. */
int mvs_cells_route(mvs_ctx* ctx, const mvs_cell* raw, const uint64_t* d_n_raw, int64_t raw_capacity, int64_t block_rows_padded,
                    int64_t block_rows, int64_t n_total, int64_t own_begin, int64_t own_end, mvs_cell* own_out,
                    int64_t own_capacity, uint64_t* d_own_count, void* send, int64_t foreign_capacity, int64_t status,
                    int64_t max_abs);
int mvs_cells_collect(mvs_ctx* ctx, const void* recv, int world, int rank, int64_t foreign_capacity, int64_t own_begin,
                      int64_t own_end, mvs_cell* own_out, int64_t own_capacity, uint64_t* d_own_count);
int mvs_cells_report(mvs_ctx* ctx, const void* recv, int world, int64_t foreign_capacity, int64_t own_rows, uint64_t* d_own_count,
                     int64_t* out);
int mvs_cells_sort_rows(mvs_ctx* ctx, const mvs_cell* cells_in, int64_t n, int64_t own_begin, int64_t own_end,
                        const uint64_t* d_own_count, mvs_cell* cells_out);
/* mvs_cells_sort_rows queued IN FRONT of mvs_cells_report's read-back: the number of cells is the one in the state block (at
 * most in_capacity of them are in cells_in), cells whose place lies beyond out_capacity are dropped.  For a caller that knows
 * from its previous step that no row holds more than 64 cells and that the buffers suffice; what the report then says -- cell
 * count, largest row -- tells it whether that held (if not: sort again with what the report says).  Asynchronous. */
int mvs_cells_sort_rows_ahead(mvs_ctx* ctx, const mvs_cell* cells_in, int64_t in_capacity, int64_t own_begin, int64_t own_end,
                              const uint64_t* d_own_count, mvs_cell* cells_out, int64_t out_capacity);
#define MVS_CELLS_HEADER_BYTES 64

/* A shard's kept cells -> the writer.  `cells` (DEVICE) holds n_cells cells ordered by (row, col) -- what a step leaves behind
 * (mvs_cells_sort_rows / mvs_cells_sort) --, rows [row_begin, row_end) of them are delivered exactly as mvs_pairwise_stream /
 * mvs_pairwise_stream_encoded deliver their result: CSR pieces of whole rows in ascending order, or the rows' finished shard
 * records, through the same callbacks under the same rules (pinned buffers of the library, a worker thread, a non-zero return
 * aborts).  A process that owns several shards (src/pairwise_comp_optimized.cpp:937-940: one row range each) compares the union
 * of their rows ONCE and calls this per shard folder with that shard's row range; cells outside the range are skipped.
 * *n_delivered (optional): the cells handed over.  Synchronous.  Writer loop replaced: src/pairwise_comp_optimized.cpp:700-790. */
int mvs_cells_stream(mvs_ctx* ctx, const mvs_cell* cells, int64_t n_cells, int64_t row_begin, int64_t row_end,
                     mvs_row_block_cb cb, void* user, int64_t* n_delivered);
int mvs_cells_stream_encoded(mvs_ctx* ctx, const mvs_cell* cells, int64_t n_cells, int64_t row_begin, int64_t row_end,
                             mvs_encoded_rows_cb cb, void* user, int64_t* n_delivered);

/* Sort n cells by (row, col) from one DEVICE buffer into another (asynchronous on the context's stream). */
int mvs_cells_sort(mvs_ctx* ctx, const mvs_cell* cells_in, int64_t n, mvs_cell* cells_out);

/* Dense int32 dot products of rows [r0,r1) x cols [c0,c1) (row-major, leading dimension c1-c0):
 * the bare `block_i.transpose() * block_j` of src/pairwise_comp_optimized.cpp:135.  For validation
 * and small problems.  `algo` 0 = matrix-core path, 1 = plain vector-ALU path (independent check). */
int mvs_pairwise_dots(mvs_ctx* ctx, const mvs_sketch_set* set, int64_t r0, int64_t r1, int64_t c0,
                      int64_t c1, int32_t* out, int mem_out, int algo);

/* ---- multi-GPU exchange ---------------------------------------------------------------------------
 * One process per GPU; shard k of src/pairwise_comp_optimized.cpp:938-940 is rank k.  Where the reference's shard
 * processes each re-read the whole vectors.bin (:953, :962), a rank here loads (or sketches) only its own rows,
 * re-codes them into ITS row block of the global plane buffer (mvs_limb_split with row_offset = rank *
 * rows_per_rank) and ONE all-gather over RCCL / xGMI gives every GPU all N columns.  Norms travel the same way.
 *
 * mvs_comm_unique_id  : rank 0 draws the 128-byte RCCL id; the caller hands it to the other ranks out of band
 *                       (a file, an environment variable, MPI, a torch.distributed store ...).
 * mvs_comm_create     : collective over all ranks (ncclCommInitRank on the context's device).  RCCL is bound at
 *                       run time (dlopen): if it is missing this call fails with MVS_E_HIP, nothing else does.
 * mvs_comm_create_callbacks : the same interface over caller-supplied collectives -- for ranks that share one
 *                       device (RCCL refuses that), for tests, or for an application's own transport.  The
 *                       callbacks receive DEVICE pointers and the context's stream and return 0 on success.
 * The collectives below are in place and asynchronous on the context's stream (the callback kind: as the
 * callback chooses); every rank must call them in the same order with the same sizes. */
#define MVS_COMM_ID_BYTES 128
typedef struct mvs_comm mvs_comm;
typedef struct {
    void* user;
    /* buf holds world * bytes_per_rank bytes on the device; block `rank` is filled in, the others are wanted */
    int (*allgather)(void* user, void* buf, size_t bytes_per_rank, int rank, int world, void* hip_stream);
    /* *value (host) becomes the maximum over the ranks */
    int (*allreduce_max_i64)(void* user, int64_t* value, int rank, int world);
} mvs_comm_callbacks;
int mvs_comm_unique_id(void* id /* MVS_COMM_ID_BYTES */);
int mvs_comm_create(mvs_ctx* ctx, const void* id, int rank, int world, mvs_comm** comm);
int mvs_comm_create_callbacks(mvs_ctx* ctx, const mvs_comm_callbacks* callbacks, int rank, int world, mvs_comm** comm);
/* File transport: ranks exchange their blocks through files named <path_prefix>_<sequence>_<rank> in a directory
 * all of them can reach.  For ranks that share one device (RCCL refuses that) and for one-GPU test boxes; every
 * rank passes the same prefix.  Collective: the call returns when all `world` ranks have met under the prefix and
 * agreed on a job nonce (option comm_timeout_s, default 600, bounds the wait); every block carries that nonce and its
 * sequence number, so files an earlier job left under the same prefix are never read as this job's data, and a rank
 * removes its own files when it fails and when the communicator is destroyed. */
int mvs_comm_create_files(mvs_ctx* ctx, const char* path_prefix, int rank, int world, mvs_comm** comm);
/* RCCL communicator for shard processes that share nothing but a directory (how the reference's shard processes
 * relate: one process per --shard_idx, src/pairwise_comp_optimized.cpp:937-940): the ranks meet through the file
 * transport's handshake above, rank 0's id (mvs_comm_unique_id) travels as one verified block, the meeting point is
 * removed again, then mvs_comm_create.  Replaces "rank 0 leaves the id in a file": a file a previous job left behind
 * can no longer be taken for this job's id. */
int mvs_comm_create_rendezvous(mvs_ctx* ctx, const char* path_prefix, int rank, int world, mvs_comm** comm);
/* Collective semantics of the file transport (mvs_comm_create_files / _rendezvous): creation is a blocking handshake --
 * every rank must be inside its create call at the same time (ranks created one after the other from ONE thread would
 * wait for each other until comm_timeout_s); destruction is local once the communicator has carried at least one exchange,
 * otherwise it waits up to 10 s for the peers so that nobody is left polling for this rank's handshake files. */
int mvs_comm_destroy(mvs_comm* comm);
int mvs_comm_info(const mvs_comm* comm, int* rank, int* world, int* is_rccl);
/* Which RCCL the library bound at run time: the path of the shared object its collectives come from (dladdr; in a process
 * that already carries a copy -- PyTorch ships one -- the loader hands back that copy) and ncclGetVersion's number
 * (e.g. 22105).  Binds the library if that has not happened yet; MVS_E_HIP when no librccl can be loaded.  A measurement
 * that claims "RCCL over xGMI" should say which RCCL. */
int mvs_comm_library(char* path, size_t path_len, int* version);
/* planes: the global plane buffer (mvs_limb_geometry of rows_per_rank * world rows); rank r has filled rows
 * [r * rows_per_rank, (r+1) * rows_per_rank).  After the call (on the stream) every block is present. */
int mvs_allgather_planes(mvs_ctx* ctx, mvs_comm* comm, int8_t* planes, int64_t rows_per_rank, int limbs, int d_pad);
/* The same for a sub-range of every rank's block: rows [row_first, row_first + row_count) of each block of
 * rows_per_rank rows (each rank has filled that sub-range of ITS block).  A rank can thus hand over the rows it has
 * finished while it is still producing the rest -- e.g. sketch the first half of its samples, start this exchange on
 * the communicator's stream, sketch the second half meanwhile (metagenome_vector_sketches_amd/parallel.py does). */
int mvs_allgather_rows(mvs_ctx* ctx, mvs_comm* comm, int8_t* planes, int64_t rows_per_rank, int64_t row_first,
                       int64_t row_count, int limbs, int d_pad);
/* values: DEVICE array of world * count_per_rank doubles (e.g. squared norms), block `rank` filled in */
int mvs_allgather_f64(mvs_ctx* ctx, mvs_comm* comm, double* values, int64_t count_per_rank);
/* any DEVICE buffer of world * bytes_per_rank bytes (e.g. kept cells that belong to other ranks' rows) */
int mvs_allgather_bytes(mvs_ctx* ctx, mvs_comm* comm, void* buf, int64_t bytes_per_rank);
/* *value (HOST) becomes the maximum over all ranks (largest |v| -> one limb code for everybody).  Synchronous. */
int mvs_allreduce_max_i64(mvs_ctx* ctx, mvs_comm* comm, int64_t* value);

/* src/pairwise_comp_optimized.cpp:903-906 and :938-940, for drivers that keep the reference CLI */
int64_t mvs_chunk_size(double max_memory_gb, int d);
void mvs_shard_rows(int64_t n, int num_shards, int shard_idx, int64_t* begin, int64_t* end);

#ifdef __cplusplus
}
#endif
#endif /* MVS_HIP_H */
