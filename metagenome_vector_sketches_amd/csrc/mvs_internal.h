// mvs_internal.h -- declarations shared by the translation units of libmvs_hip.so (not installed).
#ifndef MVS_INTERNAL_H
#define MVS_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/mvs_hip.h"

namespace mvs {

// One unit of projection work: a run of <= 65536 hashes of one sample.
struct ProjUnit {
    int64_t begin;    // index of the unit's first hash in the CSR value array
    int32_t count;    // hashes in the unit
    int32_t sample;   // output row
    int32_t single;   // 1: the unit is the whole sample -> plain store; 0: combine with atomics
    int32_t pad;
};
constexpr int kProjUnitMax = 65536;

// ---- geometry of the pairwise kernel (mvs_pairwise.hip) ----
constexpr int kTile = 128;      // samples per workgroup tile edge (rows and cols)
constexpr int kBK = 128;        // bytes (= int8 k values) per staged k-slice
constexpr int kMaxLimbs = 4;
// limb code: 1..4 = signed base-256 limbs; kLimbsK3 = three planes (l0, l1, l0+l1) of signed base-128
// digits for the 3-pass Karatsuba product (|v| <= 8127)
constexpr int kLimbsK3 = MVS_LIMBS_K3;
inline __host__ __device__ int planes_of(int code) { return code & 0xff; }
inline __host__ __device__ bool is_k3(int code) { return code == kLimbsK3; }
inline bool limb_code_ok(int code) { return (code >= 1 && code <= kMaxLimbs) || code == kLimbsK3; }

// Tuning / diagnostic options of a context (mvs_ctx_set_option).  Initial values come from the MVS_*
// environment variables, read once by mvs_ctx_create; nothing below the C ABI reads the environment.
struct Options {
    int pairwise_filter = 1;        // 0 exact kernel on every cell; 1 two-stage for blocks >= 2^22 cells; 2 always two-stage
    int filter_variant = -1;        // -1 by block size; 0 128x128 ring; 1 256x256 ring; 8 256x256 ping-pong (7/9/10: its
                                    // variants); 3/5/6 other ring shapes / depths
    int exact_variant = 3;          // re-check kernel: 3 = tree reduction over a round of 64 pairs (default); 0 = 64 pairs per
                                    // round with a shuffle butterfly per pair, 1 quarter wave per pair, 2 = 16 per round
    int pairwise_variant = 8;       // exact kernel, two limbs: 8 = ping-pong 16x16x64 (7/9: its variants), 6 = 16x16x64 ring;
                                    // 0-5 = 32x32x32 tile / ring variants (the only ones for other limb codes)
    int pairwise_symmetric = 1;     // 0: compute every tile (no mirroring)
    int pairwise_debug = 0;         // profiling ablations; only honoured by a -DMVS_ABLATIONS build
    int sort = 0;                   // kept-cell sort: 0 by list length, 1 merge, 2 radix
    int project_variant = 0;        // projection kernel: 0 by dimension; 1 / 2 blocks per wave; 12 / 14 = 2 / 4 blocks per
                                    // wave sharing the first splitmix64 round
    int markers = 0;                // 1: roctx ranges around the main entry points (rocprofv3 --marker-trace)
    int enable_k3 = 0;              // 1: mvs_sketch_set_create picks the three-plane Karatsuba code for |v| <= 8127
    int pairwise_map = 0;           // sub-patch an XCD takes in k_pairwise_pp: 0 = 4 rows x 8 cols, 1 = 8 x 4, 2 = 2 x 16
    int coarse_radix = 1;           // radix of the filter's coarse plane: 1 = smallest residual (default), 0 = ceil(max|v| / 127)
    int stream_dense = 1;           // mvs_pairwise_stream, dense results: 1 = dense byte matrix + count / scan / fill -- on a side
                                    // stream beside the next block's launch (exact kernel on whole row blocks; segmented filter),
                                    // on the context's stream where ONE filter pass feeds the matrix (short tile launches lose more
                                    // to the contention than the bubbles cost: 38.4 against 41.6 ms); 2 = always the context's
                                    // stream; 3 = always the side stream; 0 = packed list + sort
    int encode_stage_words = 64;    // device encoder: LDS words a chunk of unary codes may span before it falls back to atomics (tests)
    int stream_block_rows = 0;      // > 0: upper bound on the rows of a dense row block (tests); 0 = by the budget
    int recheck_mode = 1;           // re-check work split: 1 first round fixed + per-XCD counter, 2 counter only, 0 fixed stride, 3 eighths
    int recheck_blocks = 24;        // re-check grid in units of 256 workgroups (24: one round per wave at 100k samples)
    int cand_regions = 1;           // 1: filter waves leave up to 8 candidates in a region of their own (no atomic to wait for)
    int tile_dense_thr = 64;        // ping-pong filter: a wave with more candidates than this flags its 256 x 256 tile for the
                                    // exact kernel instead of listing them (0: list everything, give up on the block when the
                                    // list passes 1/128 of its cells -- the behaviour up to round 3)
    int stream_list_cells = 1 << 26;   // mvs_pairwise_stream, two-stage comparison: up to this many cells (candidates + cells
                                    // of flagged tiles, mirror images included) leave as ONE packed list; beyond it the dense
                                    // byte matrix takes the flagged tiles (tests lower it)
    int stream_pipeline = 1;        // mvs_pairwise_stream: 1 (default) = where the result looks dense (first tile row of the filter:
                                    // the probe, 1 % of a pass) the filter runs per SEGMENT of row blocks (5 passes), so the first
                                    // rows are final -- and on the link -- a millisecond after the call started (10 % dense 100k:
                                    // encoded rows 37.4 against 38.4 ms, CSR pieces 94.5 against 102.9 ms = 1.02 x the link);
                                    // 0 = always one filter pass over the whole row range first
    int pairwise_bdirect = 1;       // ping-pong kernels: 1 = the B operand's fragments come straight from the fragment-major plane into
                                    // registers (LDS carries the A operand only), 0 = both operands through LDS
    int fragment_major = 1;         // 1: the matrix-core kernels that copy 16 samples x 64 k values per instruction (ping-pong filter and
                                    // exact kernel, streaming search filter) read fragment-major copies of the coarse plane / limb planes
                                    // (PairwiseArgs::coarse_fm, planes_fm), built once per set; 0: the row-major planes (A/B, tests)
    int search_stream = 1;          // blocks of few rows x >= 4096 columns outside the symmetric schedule: 1 = the streaming
                                    // filter (rows resident in LDS, columns streamed into the matrix cores), 0 = the tile kernels
    int search_depth = 5;           // k_search_filter: register buffers of one k-slice x 64 columns per wave (3 .. 6); all but one
                                    // are in flight (10^6 x 2048: 256 queries 0.654 -> 0.622 ms, 512 queries 1.146 -> 1.034 from 3 to 5)
    int stream_trace = 0;           // 1: mvs_pairwise_stream prints the host-side time line of its row blocks to stderr
    int recode_rows_wg = 8;         // k_recode_rows: rows (= waves) per workgroup, 8 or 16
    int stream_copy = 0;            // streamed output, device -> host copies of the pieces: 0 = hipMemcpyAsync on the download stream (a blit
                                    // KERNEL on this stack: it holds CUs while the link drains it), 1 = hsa_amd_memory_async_copy (a DMA engine:
                                    // no CU involved; the copies' ordering then lives on the host threads of the streamed output)
    int stream_spec = 0;            // mvs_pairwise_stream_encoded, dense row blocks: 1 = a block's row passes (count, scan, fill, scan, encode)
                                    // are queued in one go with buffers sized from the blocks before it and ONE read-back at the end says
                                    // whether the sizes held (else the block is done again the careful way); 0 (default) = read the count
                                    // back before the fill and the record sizes before the encode (two host round trips per block).
                                    // Measured (10 % dense 100k): 33.5 against 33.0 ms -- the producer is not bound by its host round
                                    // trips but by the link's copy kernels on the CUs (profiles/r06_exp_dense_output.log)
    int stream_piece_mib = 32;      // mvs_pairwise_stream / mvs_cells_stream: MiB per pinned buffer = per device-to-host copy (a copy is a
                                    // blit kernel that fills the card while the link drains it: kernels that start beside one end with it)
    int plan_order = 1;             // block plans: 1 = filter launches of up to 2^20 tiles take the balanced tile order (PlanSegs::order),
                                    // 0 = the static super-patch map (A/B, tests)
    int plan_speculate = 0;         // block plans: second half of a plan sized from the previous plan's counts, no host round trip
    int plan_strip_wgs = 1 << 22;   // block plans: a rectangle whose padded grid holds more workgroups than this is cut into column
                                    // strips (a dispatch holds 2^32 work-items per dimension = 2^23 workgroups; tests lower it)
    int comm_timeout_s = 600;       // file transport: how long a rank waits for a peer's block before it gives up
    double pairwise_block_cells = 1099511627776.0;   // row-chunk bound of mvs_pairwise_rows (2^40 cells)
};

constexpr int kCandRegion = 8;   // candidate entries a filter wave can leave in its own region (one 64-byte line)

constexpr unsigned long long kStampSlots = 400000ULL;   // workgroups the time-stamp buffer of an ablation build holds

struct PairwiseArgs {
    const int8_t* planes;   // [(row*limbs + limb) * d_pad + k]
    const int8_t* planes_fm;  // the same values fragment-major, or NULL: [((row / 16 * limbs + limb) * (d_pad / 64) + k / 64) * 1024 + lane * 16],
                              // lane = (k / 16 % 4) * 16 + row % 16 (see coarse_fm); read by the ping-pong exact kernel (two limbs)
    int64_t n;              // samples
    int64_t n_alloc;        // allocated (zero padded) rows
    int d;
    int d_pad;
    int limbs;              // limb code (see kLimbsK3)
    int64_t row_begin, row_end;   // row range of this call
    int64_t sym_begin, sym_end;   // symmetric schedule: the square whose lower triangle is skipped and produced by mirroring
                                  // (the call's own row range unless a caller walks a larger square block by block)
    int64_t col_begin, col_end;   // column range (dots) / [0,n) for the comparison
    // comparison outputs
    const double* norms_sq;       // n
    const int32_t* cand_thr;      // n_alloc: conservative integer per-sample threshold part
    int keep_mode;
    double keep_coeff;            // 0.05 in the reference's keep test; j/(1+j) for a Jaccard > j search
    mvs_cell* cells;
    unsigned long long capacity;
    unsigned long long* counter;  // number of kept cells (may exceed capacity)
    // streamed output (mvs_pairwise_stream): when non-NULL a kept cell is ONE word in `packed` instead of an mvs_cell:
    // (row - pack_row0) << pack_shift | col << 16 | q (16 bits) -- sorts by (row, col) as an integer
    unsigned long long* packed;
    int64_t pack_row0;
    int pack_shift;
    // dense output (mvs_pairwise_stream where the two-limb exact kernel runs): one byte per cell, q or 0, row-major with
    // leading dimension dense_ld (a multiple of 128), row dense_row0 first; *dense_flag is set when a kept cell's q is
    // not in 1..255 (the byte cannot say so: the caller redoes that block through the packed list)
    uint8_t* dense;
    int64_t dense_row0, dense_ld;
    unsigned int* dense_flag;
    // dense outputs (dots mode)
    int32_t* dots;                // (row_end-row_begin) x (col_end-col_begin)
    int mirror_all;               // 1: every kept (row, col) is appended as (col, row) too (the transposed
                                  //    block belongs to another shard and is not computed there)
    int symmetric;                // 1: tiles strictly below the diagonal of the row range are skipped and
                                  //    produced by mirroring the kept cells of their transposes
    int debug_flags;              // profiling ablations (-DMVS_ABLATIONS builds only): 1 skip k-loop, 2 skip epilogue,
                                  // 4 filter epilogue: injected candidates instead of the accumulators' verdict
    int map_mode;                 // workgroup -> tile map: 0 = 4 x 8 sub-patch per XCD, 1 = 8 x 4, 2 = 2 x 16 (option pairwise_map),
                                  // 3 = 16 x 2 (set by the launchers for blocks less than 16 tile rows high)
    // two-stage comparison (coarse filter + exact re-check, see "filter" in mvs_pairwise.hip)
    const int8_t* coarse;         // [row * d_pad + k], c = round(v / radix[row]), |c| <= 127
    const int8_t* coarse_fm;      // the same values fragment-major, or NULL: [(row / 16 * (d_pad / 64) + k / 64) * 1024 + lane * 16],
                                  // lane = (k / 16 % 4) * 16 + row % 16 -- 1 KiB = one B fragment of v_mfma_i32_16x16x64_i8 for 16
                                  // rows and 64 k values, what a wave of the streaming search filter loads with ONE coalesced
                                  // instruction (from the row-major plane the same instruction touches 16 rows, 64 bytes each)
    const float4* fmeta;          // n_alloc: per-row filter constants {s, w, a, p}
    int2* cand;                   // candidate list: {row, col | mirror << 31}
    unsigned long long cand_capacity;
    unsigned long long* cand_counter;
    unsigned long long cand_limit;   // once the counter is beyond this the remaining filter tiles and the re-check
                                     // give up at once: the caller falls back to the exact kernel
    unsigned long long* stamps;      // ablation builds (pairwise_debug bit 8): kStampSlots x 8 words, per workgroup of
                                     //    k_pairwise_pp {XCC / HW id, start, end, k-loop end, epilogue phases} on the
                                     //    100 MHz realtime clock; the library dumps them to /tmp/mvs_stamps.bin
                                     //    (tools/exp/stamps.py reads that)
    int recheck_mode;                    // work split of k_exact_pairs_tree (see there)
    unsigned long long* recheck_queue;   // 8 zeroed counters, 64 bytes apart: the re-check hands its rounds out per XCD
    unsigned int* cand_hdr;          // per (workgroup, wave) region of the ping-pong filter: number of candidates the wave
    int2* cand_ent;                  //    left in its kCandRegion entries (0: none, or it went to the list itself)
    unsigned int* cand_stop;         // set by the wave that takes the counter past the limit; lives on its own cache
                                     // line (polling the counter itself queues behind its atomics)
    // Tile-granular two-stage comparison (ping-pong filter on 256 x 256 tiles).  A filter wave that finds more than
    // tile_dense_thr candidates in its 128 x 64 cells does not list them: it flags its tile, tile_flag[tr * tile_flag_ld +
    // tc] (tr, tc relative to row_begin / col_begin in units of 256), and the exact ping-pong kernel later computes the
    // flagged tiles -- and only those -- from tile_list (four 128 x 128 tiles each).  Candidates that other waves of a
    // flagged tile had already listed are dropped by k_cand_prune before the re-check, so no cell is produced twice.
    unsigned int* tile_flag;         // NULL: every candidate is listed (ring filters, option tile_dense_thr = 0)
    int tile_flag_ld;
    unsigned int tile_dense_thr;
    unsigned int* tile_flag_count;   // flagged tiles so far; beyond tile_flag_limit the filter raises cand_stop (the result
    unsigned int tile_flag_limit;    //    is dense nearly everywhere: the exact kernel alone is faster)
    const int* tile_list;            // k_pairwise_pp<0>: flagged filter tiles (row-major ids), NULL = the whole grid
    int tile_list_n;
    // Block plans (mvs_plan_*: a rank's share of the symmetric multi-rank schedule in ONE launch, PlanSegs below).
    // [row_begin, row_end) x [col_begin, col_end) is then the FRAME the tile flags, the flagged-tile list and the candidate
    // pruning index into; the tiles themselves come from the launch's segments.  plan = 1 also keeps `symmetric` together
    // with `mirror_all`: inside the square [sym_begin, sym_end)^2 the symmetric schedule rules (tiles below the diagonal
    // skipped, cells above it mirrored), every kept cell outside it is mirrored (its transposed block is nobody else's).
    int plan;
    unsigned long long cand_region_base;   // first candidate region of this launch (several filter launches share the arrays)
};

// The rectangles of one plan launch of k_pairwise_pp: a 1-D grid, segment s owning workgroups [wg_begin[s], wg_begin[s + 1])
// = n_spr x n_spc super-patches of 256 workgroups (map_tile inside each, so the XCD-aware placement of the single-block
// launch carries over: every segment starts on a multiple of 256 workgroups).  n = 0: the launch is one block on the 2-D grid.
constexpr int kPlanSegs = 16;
struct PlanSegs {
    int n;
    unsigned wg_begin[kPlanSegs + 1];
    int n_tr[kPlanSegs], n_tc[kPlanSegs], n_spc[kPlanSegs];
    long long i_begin[kPlanSegs], j_begin[kPlanSegs];
    // Balanced order (plan_tile_order; NULL: the static map above): the tiles the launch really computes, XCD label x
    // (blockIdx.x % 8) takes entries [x * order_per, (x + 1) * order_per) -- its share of every super-patch, in sub-patch
    // order, so that the XCDs still walk the same super-patch at the same time (panels shared through the memory-side cache)
    // but no XCD carries more tiles than another: a triangle's static sub-patches hold 32, 26, 10 or 0 tiles, and a launch
    // of a few rounds (1225 tiles of a rank's diagonal block, 820 of a 10k x 10k comparison) waited for its fullest XCD --
    // 174 tiles where 154 would do.  Entry = segment << 28 | tile row << 14 | tile column; ~0u = no tile.
    const unsigned* order;
    unsigned order_per;
};

// Which 256 x 256 tiles of the dense byte matrix can hold a kept cell (tile-granular comparison feeding the matrix): the
// tiles the filter flagged (the exact kernel wrote them whole), their mirror images, and the tiles the re-check's kept
// cells were scattered into.  The row -> CSR passes read only those; nothing else of the matrix is ever cleared or read.
struct DenseActive {
    const unsigned int* flags;   // [n_tr * n_tc] of the matrix's tile grid (rows relative to its first row); NULL: every tile
    const unsigned int* touch;   // same shape
    int n_tr, n_tc;
    int o;                       // tile column of the matrix's first row (row_begin / 256): mirror of (tr, tc) = (tc - o, tr + o)
    int sym;                     // the symmetric schedule was on: mirror images of flagged tiles exist
    long long row_rel0;          // first row of the launch, relative to the matrix's first row
};

// per-row statistics of the coarse plane: radix m, sum c^2, sum r^2 (r = v - m*c), and whether the row's sum
// of squares reaches 2^31 (its dots may wrap: the filter then passes every pair of that row on to the re-check)
struct CoarseRow {
    int32_t radix;
    int32_t c2;
    int32_t r2;
    int32_t big;
};

// d_sumsq / d_max_abs non-NULL: fused statistics (all samples must be single units; d_sumsq zeroed by the caller)
int launch_project(hipStream_t stream, const uint64_t* d_hashes, const ProjUnit* d_units, int64_t n_units,
                   int d, int32_t* d_out, int variant, unsigned long long* d_sumsq, unsigned long long* d_max_abs);
int launch_sumsq(hipStream_t stream, const int32_t* d_sk, int64_t n, int d, int64_t* d_out);
int launch_saturate_i16(hipStream_t stream, const int32_t* d_in, int64_t n, int16_t* d_out);
int launch_stats(hipStream_t stream, const int32_t* d_sk, int64_t n, int d, int64_t* d_sumsq,
                 unsigned long long* d_max_abs);

int launch_max_abs(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_elems,
                   unsigned long long* d_out);
int launch_limb_split(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_rows, int d, int limbs,
                      int8_t* d_planes, int d_pad, int64_t row_offset);
int launch_cand_thr(hipStream_t stream, const double* d_norms_sq, int64_t n, int64_t n_alloc, int d,
                    double coeff, int32_t* d_thr);
// mode 0: comparison (kept cells), mode 1: dense dots.  algo 0: MFMA, 1: vector ALU.
int launch_pairwise(hipStream_t stream, const PairwiseArgs& a, int mode, int algo, const Options& opt);
// two-stage comparison for two base-256 limbs: coarse plane + row statistics from the limb planes,
// per-call filter constants, the one-pass filter
// that appends candidate pairs, and the exact re-check of the candidates that appends kept cells
int launch_coarse_build(hipStream_t stream, const int8_t* d_planes, int64_t n, int64_t n_alloc, int d_pad,
                        int8_t* d_coarse, CoarseRow* d_rows, int radix_mode);
// limb split + coarse build + fragment-major copy of a row range in one pass (k_recode_rows); false: no fused kernel for
// this geometry (the caller takes the three separate launches)
int launch_planes_from_wire(hipStream_t stream, const int8_t* d_lo_wire, const int8_t* d_coarse_fm, const CoarseRow* d_rows,
                            int64_t count, int d_pad, int8_t* d_planes, const unsigned char* d_need = nullptr, int64_t skip_first = 0,
                            int64_t skip_count = 0);
int launch_rows_needed(hipStream_t stream, const PairwiseArgs& a, int n_tr, int n_tc, int64_t f0, int64_t f1, int64_t n_rows,
                       unsigned char* d_need);
bool launch_recode_rows(hipStream_t stream, const void* d_sk, int elem_bytes, int64_t n_rows, int64_t count, int d, int d_pad,
                        int8_t* d_planes, int8_t* d_coarse_fm, CoarseRow* d_rows, int radix_mode, int rows_per_wg);
// fragment-major copy of the coarse plane (PairwiseArgs::coarse_fm) / of the limb planes (planes_fm): `limbs` planes per row
int launch_coarse_fm(hipStream_t stream, const int8_t* d_coarse, int64_t n_alloc, int d_pad, int8_t* d_fm, int limbs = 1);
// true when launch_pairwise / launch_exact_tiles run the kernel that reads planes_fm for this set
bool exact_reads_fm(const PairwiseArgs& a, const Options& opt);
int launch_filter_meta(hipStream_t stream, const CoarseRow* d_rows, const double* d_norms_sq, int64_t n,
                       int64_t n_alloc, int d, double coeff, float4* d_meta);
int launch_filter(hipStream_t stream, const PairwiseArgs& a, const Options& opt, const unsigned* order = nullptr, unsigned order_per = 0);
struct PlanSegs;
bool filter_order_geometry(const PairwiseArgs& a, const Options& opt, PairwiseArgs* b, PlanSegs* segs);
// block plans (mvs_plan_*): rectangles {row_begin, row_end, col_begin, col_end} -> the segments of one launch (returns its
// 1-D grid in workgroups, 0: empty, -1: too many rectangles / too large for one launch), and that launch of the ping-pong filter
long long plan_segments(const int64_t (*blocks)[4], int n, PlanSegs* segs);
int launch_filter_plan(hipStream_t stream, const PairwiseArgs& a, const PlanSegs& segs, long long workgroups);
// the balanced order of a plan launch (host): per XCD label the tiles it computes, order_per entries each (padded with ~0u);
// false when a coordinate does not fit an entry or the launch is too large for a list to pay (the static map is used then)
bool plan_tile_order(const PairwiseArgs& a, const PlanSegs& segs, std::vector<unsigned>* order, unsigned* per);
int launch_exact_pairs(hipStream_t stream, const PairwiseArgs& a, const Options& opt, long long n_hint = -1);
// candidate regions of the ping-pong filter: how many (workgroups x 8 waves) launch_filter's grid has for this block, 0 if
// the variant it would pick appends with atomics only; k_cand_gather moves the regions' contents into the candidate list
int64_t filter_region_count(const PairwiseArgs& a, const Options& opt);
int launch_cand_gather(hipStream_t stream, const PairwiseArgs& a, int64_t n_regions);
// tile-granular two-stage comparison: does launch_filter pick a kernel that can flag tiles for this block, and the tile
// grid (256 x 256) it works on; flags -> per-tile-row counts -> row-major list of flagged tile ids (d_list holds the total);
// candidates whose tile is flagged are dropped (d_out / d_out_count: the pruned list); the exact kernel on a run of the list
bool filter_flags_tiles(const PairwiseArgs& a, const Options& opt);
// true when launch_filter would take the streaming search filter for this block (few rows, many columns, no symmetry)
bool filter_streams_rows(const PairwiseArgs& a, const Options& opt);
// ... or another filter kernel that reads the fragment-major coarse plane (the ping-pong tile filter)
bool filter_streams(const PairwiseArgs& a, const Options& opt);
void filter_tile_grid(const PairwiseArgs& a, int* n_tr, int* n_tc);
int launch_tile_count(hipStream_t stream, const unsigned int* d_flags, int n_tr, int n_tc, int* d_row_count);
int launch_tile_list(hipStream_t stream, const unsigned int* d_flags, int n_tr, int n_tc, const int* d_row_count, int* d_list,
                     int cap = 0x7fffffff);
int launch_cand_prune(hipStream_t stream, const PairwiseArgs& a, unsigned long long n_cand, int2* d_out,
                      unsigned long long* d_out_count, long long n_hint = -1);
int launch_exact_tiles(hipStream_t stream, const PairwiseArgs& a, const int* d_list, int n_list, const Options& opt,
                       bool device_count = false);
// the cell count of a plan whose speculative second half ran on stale counts (k_plan_verdict)
constexpr unsigned long long kPlanStale = 1ULL << 62;
int launch_plan_verdict(hipStream_t stream, unsigned long long* d_counter, unsigned long long cand_capacity, const int* d_tile_total,
                        int tile_cap, bool tiles_skipped);
// packed cells of the streamed output: radix sort on the (row, col) bits, then CSR arrays (row_ptr over `rows` rows,
// col, q as 8 bits -- *d_wide set if some q needs 16 -- or as 16 bits when d_q16 is given)
int sort_packed(hipStream_t stream, unsigned long long* d_in, unsigned long long* d_out, int64_t n, int begin_bit, int end_bit,
                void* d_scratch, size_t scratch_bytes, size_t* scratch_needed);
int launch_packed_csr(hipStream_t stream, const unsigned long long* d_keys, int64_t n, int shift, int64_t rows,
                      unsigned long long col_mask, long long* d_row_ptr, int32_t* d_col, uint8_t* d_q8, uint16_t* d_q16,
                      unsigned int* d_wide);
// the tile rows [tr0, tr0 + n_trows) the rows of a launch fall into and the tile columns of the matrix: sizes of the
// active-tile lists (d_list: n_trows x n_tc ints, d_list_n: n_trows ints) the row passes walk
void dense_tile_rows(const DenseActive& active, int64_t rows, int64_t n_cols, int* tr0, int* n_trows, int* n_tc);
int launch_dense_count(hipStream_t stream, const uint8_t* d_dense, int64_t ld, int64_t n_cols, int64_t rows, long long* d_counts,
                       int2* d_ends, const DenseActive& active, int* d_list, int* d_list_n);
int dense_row_ptr(hipStream_t stream, long long* d_counts, long long* d_row_ptr, int64_t rows, void* d_scratch, size_t scratch_bytes,
                  size_t* scratch_needed);
struct EncRow;
int launch_dense_fill(hipStream_t stream, const uint8_t* d_dense, int64_t ld, int64_t n_cols, int64_t rows, const long long* d_row_ptr,
                      int32_t* d_col, uint8_t* d_q, const DenseActive& active, const int* d_list, const int* d_list_n,
                      const int2* d_ends, unsigned long long* d_size, unsigned int* d_jac, unsigned int* d_first_col, EncRow* d_par,
                      long long capacity = 0x7fffffffffffffffLL);
// the re-check's kept cells (packed words, *d_n of them, rows relative to pack_row0 = the matrix's first row) into the dense
// byte matrix: mark the tiles they fall into (newly touched ones are listed in d_new, count in d_new[-1] .. i.e. d_new_count),
// clear those tiles, then write the bytes; *d_odd is set when a q is not in 1..255
int launch_packed_to_dense(hipStream_t stream, const unsigned long long* d_keys, const unsigned long long* d_n, int shift,
                           unsigned long long col_mask, uint8_t* d_dense, int64_t ld, int64_t matrix_rows, unsigned int* d_touch,
                           int n_tc, int* d_new, unsigned int* d_new_count, unsigned int* d_odd);
// true when launch_pairwise would run the kernel whose epilogue can write the dense byte matrix (two base-256 limbs on the
// ping-pong kernel)
bool exact_kernel_writes_dense(const PairwiseArgs& a, const Options& opt);
// kept cells of a block plan (storage coordinates) -> sample indices, own rows / other ranks' rows (mvs_cells_route);
// the cells of own rows out of the other ranks' send buffers (mvs_cells_collect)
int launch_cells_route(hipStream_t stream, const mvs_cell* d_raw, const unsigned long long* d_n_raw, unsigned long long raw_capacity,
                       long long block_pad, long long block_rows, long long n_total, int own_begin, int own_end, mvs_cell* d_own,
                       unsigned long long own_capacity, unsigned long long* d_own_count, unsigned long long* d_send,
                       unsigned long long foreign_capacity, long long status, long long max_abs);
int launch_cells_collect(hipStream_t stream, const unsigned long long* d_recv, int world, int rank, unsigned long long capacity,
                         int own_begin, int own_end, mvs_cell* d_own, unsigned long long own_capacity, unsigned long long* d_own_count);
// the shard's cells by row buckets (rows of <= 64 cells): state block as the route / collect kernels fill it
int launch_rows_max(hipStream_t stream, unsigned long long* d_state, int rows);
// up to 8 device ranges (4-byte aligned, sizes multiples of 4; NULL / empty ones are skipped) zeroed by ONE launch
int launch_zero_ranges(hipStream_t stream, void* const* ptrs, const size_t* bytes, int n);
// sort_cells_rows also leaves the widest row in d_state[1] (what launch_rows_max computes)
int sort_cells_rows(hipStream_t stream, const mvs_cell* d_in, mvs_cell* d_out, int64_t n, int row0, int rows,
                    const unsigned long long* d_state, void* d_scratch, size_t scratch_bytes, size_t* scratch_needed,
                    int64_t in_cap = -1, int64_t out_cap = -1);
// sort cells by (row, col); tmp buffers owned by the caller
int sort_cells(hipStream_t stream, mvs_cell* d_cells, mvs_cell* d_tmp, int64_t n, void* d_scratch,
               size_t scratch_bytes, size_t* scratch_needed, const Options& opt);

}  // namespace mvs

#endif
