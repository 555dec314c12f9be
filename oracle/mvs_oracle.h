/*
 * mvs_oracle.h -- CPU restatement of the reference's sketch + pairwise hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libmvs_hip.so) never
 * links, loads or calls anything in oracle/.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   projection half  -- PINNED: checked bit-exact against the reference binaries built from
 *                       /root/reference into oracle/_ref/ (oracle/Makefile) and against the
 *                       known-answer values recorded in SURVEY.md section 4.
 *   pairwise half    -- the reference translation units need the absent `bits` submodule and are
 *                       unbuildable as a whole, but their bits-free functions are not: loaders, int32 /
 *                       int16 products and keep tests (src/pairwise_comp_optimized.cpp:33-160,
 *                       _16bits.cpp:40-244) and the first lines of both writers (Jaccard quantiser :654-672;
 *                       round(dot / d) _16bits.cpp:260-280) are compiled from line ranges into
 *                       oracle/_ref/ref_pairwise32 / ref_pairwise16 (oracle/Makefile ref_pairwise) and this
 *                       restatement reproduces their output cell for cell, in order, on 23 recorded runs
 *                       (tests/golden/ref_pairwise.json): PINNED.  What the writers hand to bits:: after
 *                       those lines is out of reach.  Codec bytes: parity unpinned.
 *
 * Every function cites the reference file:line it restates (paths relative to /root/reference).
 */
#ifndef MVS_ORACLE_H
#define MVS_ORACLE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* one kept cell of the all-vs-all matrix, in the order the reference appends them
 * (src/pairwise_comp_optimized.cpp:974-980) */
typedef struct {
    int32_t row;
    int32_t col;
    int32_t dot;   /* int32 dot product, wrapped mod 2^32 (MatrixXi, :135) */
    int32_t q;     /* quantised Jaccard, round(J*255) (:660-665) */
} mvs_oracle_cell;

/* src/random_projection.cpp:13-17 */
uint64_t mvs_oracle_splitmix64(uint64_t x);

/* src/random_projection.cpp:9-26 -- hashes must be unique (the reference holds them in an unordered_set) */
void mvs_oracle_project(const uint64_t* hashes, int64_t n, int d, int32_t* out);

/* src/project_everything.cpp:289-298 -- OpenMP loop over samples; threads<=0 means OpenMP default */
void mvs_oracle_project_csr(const uint64_t* hashes, const int64_t* offsets, int64_t n_samples,
                            int d, int32_t* out, int threads);

/* same result, restructured for speed (x computed once per (hash, 64-dim block), popcount form
 * v[k] = n - 2*count_k).  Used as the "port" CPU baseline so the baseline is not a strawman. */
void mvs_oracle_project_csr_fast(const uint64_t* hashes, const int64_t* offsets, int64_t n_samples,
                                 int d, int32_t* out, int threads);

/* sum of squares of one sketch (exact, int64) */
int64_t mvs_oracle_sumsq(const int32_t* v, int d);

/* src/project_everything.cpp:328-329 -- float32 path, sequential accumulation order.
 * The reference's value comes from Eigen's vectorised reduction under -ffast-math and is not
 * bit-reproducible; tolerance 1e-5 relative. */
double mvs_oracle_norm_f32path(const int32_t* v, int d);

/* the build's definition: sqrt(double(sum v^2) / d) (SURVEY.md 8c "Recommended definition") */
double mvs_oracle_norm(const int32_t* v, int d);

/* "%g" (6 significant digits), what `ostream << double` prints at src/project_everything.cpp:330;
 * returns the number of characters written */
int mvs_oracle_format_norm(double norm, char* buf, int buflen);

/* src/pairwise_comp_optimized.cpp:893-901 -- stod(text)^2 */
double mvs_oracle_norm_sq_from_text(const char* text);

/* src/project_everything.cpp:332-347 */
void mvs_oracle_saturate_i16(const int32_t* in, int64_t n, int16_t* out);

/* src/pairwise_comp_optimized.cpp:135 -- int32 dot product, wraps mod 2^32 */
int32_t mvs_oracle_dot_i32(const int32_t* a, const int32_t* b, int d);
/* src/pairwise_comp_optimized_16bits.cpp:144-208 -- int16 inputs, int32 accumulate (wraps) */
int32_t mvs_oracle_dot_i16(const int16_t* a, const int16_t* b, int d);

/* src/pairwise_comp_optimized.cpp:139-141 -- truncating integer division, double compare */
int mvs_oracle_keep_i32(int32_t dot, int d, double n2_i, double n2_j);
/* src/pairwise_comp_optimized_16bits.cpp:211-218 -- floating division */
int mvs_oracle_keep_i16(int32_t dot, int d, double n2_i, double n2_j);

/* src/pairwise_comp_optimized.cpp:658-665 */
int32_t mvs_oracle_quantize(int32_t dot, int d, double n2_row, double n2_col);

/* src/pairwise_comp_optimized.cpp:903-982 (tiling + shard loop) + :135-147 + :658-665.
 * elem_bytes 4: int32 sketches, integer keep test.  elem_bytes 2: int16 sketches, floating keep
 * test (src/pairwise_comp_optimized_16bits.cpp:96-244).  Cells are emitted in the reference's
 * (i-chunk, j-chunk, i, j) order; returns the number of kept cells (may exceed cap; only the
 * first cap are stored). */
int64_t mvs_oracle_pairwise_rows(const void* sketches, int elem_bytes, int64_t n, int d,
                                 const double* norms_sq, int64_t row_begin, int64_t row_end,
                                 int64_t chunk, mvs_oracle_cell* out, int64_t cap, int threads);

/* src/pairwise_comp_optimized.cpp:903-906 */
int64_t mvs_oracle_chunk_size(double max_memory_gb, int d);
/* src/pairwise_comp_optimized.cpp:938-940 */
void mvs_oracle_shard_rows(int64_t n, int num_shards, int shard_idx, int64_t* begin, int64_t* end);

/* dense int32 dot-product block rows[r0,r1) x cols[c0,c1), row-major, OpenMP.  Baseline timing
 * and dot-product parity. */
void mvs_oracle_dots_dense(const int32_t* sk, int64_t n, int d, int64_t r0, int64_t r1,
                           int64_t c0, int64_t c1, int32_t* out, int threads);

int mvs_oracle_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
