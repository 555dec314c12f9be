// mvs_encode.h -- device-side shard encoder (mvs_encode.hip), shared with mvs_capi.hip (not installed).
#ifndef MVS_ENCODE_H
#define MVS_ENCODE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/mvs_hip.h"

namespace mvs {

// what the size pass leaves per row for the fill pass
struct EncRow {
    unsigned long long high_bits;   // length of the unary part of the row's rice_sequence (0: the row has < 2 cells)
    unsigned int wq;                // bit width of the row's q values (compact_vector)
    unsigned int k;                 // Rice parameter of the column deltas
};

// Pass 1, one wave per row of a CSR block (row_ptr over `rows` rows, col ascending per row, q 8 or 16 bits wide):
// size[r] = bytes of the row's record (0 for a row without cells), jac[r] = bytes of its compact_vector,
// first_col[r], par[r].
int launch_encode_sizes(hipStream_t stream, const long long* d_row_ptr, const int32_t* d_col, const void* d_q, int q_bytes,
                        int64_t rows, unsigned long long* d_size, unsigned int* d_jac, unsigned int* d_first_col, EncRow* d_par);
// offsets = exclusive scan of size over rows + 1 entries (size[rows] must be 0): offset[rows] = total bytes
int encode_offsets(hipStream_t stream, unsigned long long* d_size, unsigned long long* d_offset, int64_t rows, void* d_scratch,
                   size_t scratch_bytes, size_t* scratch_needed);
// Pass 2, one wave per row: the records, back to back, into d_out (zero-filled by the caller, 8-byte aligned).
// stage_words (1..64): words of LDS a chunk of 64 unary codes may span before the kernel falls back to atomics (64; tests
// lower it to exercise the fallback)
int launch_encode_fill(hipStream_t stream, const long long* d_row_ptr, const int32_t* d_col, const void* d_q, int q_bytes,
                       int64_t rows, const unsigned long long* d_offset, const EncRow* d_par, unsigned char* d_out, int stage_words,
                       unsigned long long cap_cells = ~0ULL, unsigned long long cap_bytes = ~0ULL);

// A list of kept cells ordered by (row, col) -> the CSR arrays above for rows [row0, row0 + rows) (mvs_cells_stream*):
// d_abs_ptr[r] = index of the first cell of row row0 + r in the list (rows + 1 entries), then the columns / q of those cells
// into arrays that start at 0 with the row index rebased (d_rel_ptr, may be NULL); n_upper bounds the cells (grid size);
// *d_wide is set when a q does not fit q_bytes == 1.
int launch_cells_rowptr(hipStream_t stream, const mvs_cell* d_cells, int64_t n, int64_t row0, int64_t rows, long long* d_abs_ptr);
int launch_cells_split(hipStream_t stream, const mvs_cell* d_cells, const long long* d_abs_ptr, int64_t rows, int64_t n_upper,
                       long long* d_rel_ptr, int32_t* d_col, void* d_q, int q_bytes, unsigned int* d_wide);

}  // namespace mvs

#endif
