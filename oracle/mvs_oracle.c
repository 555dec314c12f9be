/*
 * mvs_oracle.c -- CPU restatement of the reference's sketch + pairwise hot path.
 * TEST INFRASTRUCTURE ONLY (see mvs_oracle.h for the rules and the pinning status).
 * Citations are relative to /root/reference.
 */
#include "mvs_oracle.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* src/random_projection.cpp:13-17 (the `+ i` of :13 is done by the caller) */
uint64_t mvs_oracle_splitmix64(uint64_t x) {
    x += 0x9e3779b97f4a7c15ULL;
    x = (x ^ (x >> 30)) * 0xbf58476d1ce4e5b9ULL;
    x = (x ^ (x >> 27)) * 0x94d049bb133111ebULL;
    x = x ^ (x >> 31);
    return x;
}

/* src/random_projection.cpp:9-26, loop for loop */
void mvs_oracle_project(const uint64_t* hashes, int64_t n, int d, int32_t* out) {
    for (int k = 0; k < d; ++k) out[k] = 0;                       /* :10 VectorXi::Zero(d) */
    for (int64_t h = 0; h < n; ++h) {                             /* :11 */
        for (int i = 0; i < d; i += 64) {                         /* :12 */
            uint64_t x = mvs_oracle_splitmix64(hashes[h] + (uint64_t)i);   /* :13-17, wraps mod 2^64 */
            for (int b = 0; b < 64 && (i + b) < d; ++b) {         /* :19 tail block when d % 64 != 0 */
                int projected = 1 - 2 * (int)((x >> b) & 1);      /* :20 */
                out[i + b] += projected;                          /* :21 */
            }
        }
    }
}

/* src/project_everything.cpp:289-298 */
void mvs_oracle_project_csr(const uint64_t* hashes, const int64_t* offsets, int64_t n_samples,
                            int d, int32_t* out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(dynamic)
    for (int64_t s = 0; s < n_samples; ++s)
        mvs_oracle_project(hashes + offsets[s], offsets[s + 1] - offsets[s], d, out + s * (int64_t)d);
}

/* Same values as mvs_oracle_project: v[k] = n - 2*count_k with count_k = #hashes whose bit is set.
 * Counting set bits per position in 8-bit lanes (flushed every 255 hashes) lets the compiler
 * vectorise; this is the strongest CPU form we could write without intrinsics. */
static void project_fast_one(const uint64_t* hashes, int64_t n, int d, int32_t* out) {
    const int nblk = (d + 63) / 64;
    for (int k = 0; k < d; ++k) out[k] = 0;
    for (int blk = 0; blk < nblk; ++blk) {
        uint32_t cnt[64];
        memset(cnt, 0, sizeof cnt);
        int64_t h = 0;
        while (h < n) {
            int64_t lim = h + 255 < n ? h + 255 : n;
            uint8_t c8[64];
            memset(c8, 0, sizeof c8);
            for (; h < lim; ++h) {
                uint64_t x = mvs_oracle_splitmix64(hashes[h] + (uint64_t)(blk * 64));
                for (int b = 0; b < 64; ++b) c8[b] += (uint8_t)((x >> b) & 1);
            }
            for (int b = 0; b < 64; ++b) cnt[b] += c8[b];
        }
        for (int b = 0; b < 64 && blk * 64 + b < d; ++b)
            out[blk * 64 + b] = (int32_t)(n - 2 * (int64_t)cnt[b]);
    }
}

void mvs_oracle_project_csr_fast(const uint64_t* hashes, const int64_t* offsets, int64_t n_samples,
                                 int d, int32_t* out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
#pragma omp parallel for schedule(dynamic)
    for (int64_t s = 0; s < n_samples; ++s)
        project_fast_one(hashes + offsets[s], offsets[s + 1] - offsets[s], d, out + s * (int64_t)d);
}

int64_t mvs_oracle_sumsq(const int32_t* v, int d) {
    int64_t s = 0;
    for (int k = 0; k < d; ++k) s += (int64_t)v[k] * (int64_t)v[k];
    return s;
}

/* src/project_everything.cpp:328-329: vec.cast<float>() / sqrt(float(d)), then .norm() */
double mvs_oracle_norm_f32path(const int32_t* v, int d) {
    const float inv = sqrtf((float)d);
    float acc = 0.0f;
    for (int k = 0; k < d; ++k) {
        float f = (float)v[k] / inv;
        acc += f * f;
    }
    return (double)sqrtf(acc);
}

double mvs_oracle_norm(const int32_t* v, int d) {
    return sqrt((double)mvs_oracle_sumsq(v, d) / (double)d);
}

/* src/project_everything.cpp:330: `norm_out << base_name << " " << norm` with the default
 * ostream precision (6) and no format flags == printf("%g") */
int mvs_oracle_format_norm(double norm, char* buf, int buflen) {
    return snprintf(buf, (size_t)buflen, "%g", norm);
}

/* src/pairwise_comp_optimized.cpp:898-899: stod(line.substr(pos + 1)); norm*norm */
double mvs_oracle_norm_sq_from_text(const char* text) {
    double norm = strtod(text, NULL);
    return norm * norm;
}

/* src/project_everything.cpp:332-347 */
void mvs_oracle_saturate_i16(const int32_t* in, int64_t n, int16_t* out) {
    for (int64_t k = 0; k < n; ++k) {
        int32_t v = in[k];
        out[k] = v > 32767 ? (int16_t)32767 : (v < -32768 ? (int16_t)-32768 : (int16_t)v);
    }
}

/* src/pairwise_comp_optimized.cpp:135: MatrixXi product -> int32 arithmetic, wraps mod 2^32.
 * Unsigned arithmetic gives the same bits without signed-overflow UB. */
int32_t mvs_oracle_dot_i32(const int32_t* a, const int32_t* b, int d) {
    uint32_t s = 0;
    for (int k = 0; k < d; ++k) s += (uint32_t)a[k] * (uint32_t)b[k];
    return (int32_t)s;
}

/* src/pairwise_comp_optimized_16bits.cpp:144-208: _mm256_madd_epi16 pairs + int32 accumulate */
int32_t mvs_oracle_dot_i16(const int16_t* a, const int16_t* b, int d) {
    uint32_t s = 0;
    for (int k = 0; k < d; ++k) s += (uint32_t)((int32_t)a[k] * (int32_t)b[k]);
    return (int32_t)s;
}

/* src/pairwise_comp_optimized.cpp:139-141:
 *   double threshold = 0.05 * (norms_i(i) + norms_j(j));
 *   int64_t dot_product = dot_products(i, j);
 *   if (dot_product / dimension > threshold)          <- int64 / int: truncates toward zero */
int mvs_oracle_keep_i32(int32_t dot, int d, double n2_i, double n2_j) {
    double threshold = 0.05 * (n2_i + n2_j);
    int64_t q = (int64_t)dot / (int64_t)d;
    return (double)q > threshold;
}

/* src/pairwise_comp_optimized_16bits.cpp:211,218 */
int mvs_oracle_keep_i16(int32_t dot, int d, double n2_i, double n2_j) {
    double threshold = 0.05 * (n2_i + n2_j);
    return (double)dot / (double)d > threshold;
}

/* src/pairwise_comp_optimized.cpp:654-665 */
int32_t mvs_oracle_quantize(int32_t dot, int d, double n2_row, double n2_col) {
    const double MULT_CONST = 255.0;                       /* :654 (1ULL << 8) - 1 */
    double inter = (double)dot / (double)d;                /* :661 */
    double jaccard = inter / (n2_row + n2_col - inter);    /* :662 */
    if (jaccard > 1) jaccard = 1;                          /* :663 */
    double r = round(jaccard * MULT_CONST);                /* :664 round = half away from zero */
    if (!(r == r)) return 0;                               /* NaN: undefined in the reference */
    return (int32_t)(uint16_t)(int64_t)r;                  /* :664 static_cast<uint16_t> */
}

int64_t mvs_oracle_chunk_size(double max_memory_gb, int d) {
    int bytes_per_vector = d * (int)sizeof(int32_t);                       /* :903 */
    int64_t max_bytes = (int64_t)(max_memory_gb * 1024 * 1024 * 1024);     /* :904 */
    return max_bytes / ((int64_t)bytes_per_vector * bytes_per_vector);     /* :906 (int*int in the
                                             reference; identical while 16 d^2 < 2^31, d < 11586) */
}

void mvs_oracle_shard_rows(int64_t n, int num_shards, int shard_idx, int64_t* begin, int64_t* end) {
    int64_t rows_per_shard = (n + num_shards - 1) / num_shards;            /* :938 */
    int64_t b = (int64_t)shard_idx * rows_per_shard;                       /* :939 */
    int64_t e = b + rows_per_shard < n ? b + rows_per_shard : n;           /* :940 */
    if (b > n) b = n;
    if (e < b) e = b;
    *begin = b;
    *end = e;
}

static void dots_tile(const void* sk, int elem_bytes, int d, int64_t i0, int64_t i1, int64_t j0,
                      int64_t j1, int32_t* out) {
    const int64_t cj = j1 - j0;
    /* i x j collapsed: a tile has <= 192 rows, which alone would starve a host with more threads than that */
#pragma omp parallel for collapse(2) schedule(static)
    for (int64_t i = i0; i < i1; ++i) {
        for (int64_t j = j0; j < j1; ++j) {
            int32_t v;
            if (elem_bytes == 4)
                v = mvs_oracle_dot_i32((const int32_t*)sk + i * d, (const int32_t*)sk + j * d, d);
            else
                v = mvs_oracle_dot_i16((const int16_t*)sk + i * d, (const int16_t*)sk + j * d, d);
            out[(i - i0) * cj + (j - j0)] = v;
        }
    }
}

int64_t mvs_oracle_pairwise_rows(const void* sketches, int elem_bytes, int64_t n, int d,
                                 const double* norms_sq, int64_t row_begin, int64_t row_end,
                                 int64_t chunk, mvs_oracle_cell* out, int64_t cap, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);                         /* :927 */
#endif
    (void)threads;
    if (chunk < 1) return -1;
    int64_t count = 0;
    int64_t ci_max = chunk < row_end - row_begin ? chunk : row_end - row_begin;
    int64_t cj_max = chunk < n ? chunk : n;
    if (ci_max < 1) return 0;
    int32_t* tile = (int32_t*)malloc((size_t)ci_max * (size_t)cj_max * sizeof(int32_t));
    if (!tile) return -2;
    for (int64_t bi = row_begin; bi < row_end; bi += chunk) {              /* :949 */
        int64_t ei = bi + chunk < row_end ? bi + chunk : row_end;          /* :950 */
        for (int64_t bj = 0; bj < n; bj += chunk) {                        /* :958 */
            int64_t ej = bj + chunk < n ? bj + chunk : n;                  /* :959 */
            dots_tile(sketches, elem_bytes, d, bi, ei, bj, ej, tile);      /* :135 */
            const int64_t cj = ej - bj;
            for (int64_t i = bi; i < ei; ++i) {                            /* :137 */
                for (int64_t j = bj; j < ej; ++j) {                        /* :138 */
                    int32_t dot = tile[(i - bi) * cj + (j - bj)];
                    int keep = elem_bytes == 4
                                   ? mvs_oracle_keep_i32(dot, d, norms_sq[i], norms_sq[j])
                                   : mvs_oracle_keep_i16(dot, d, norms_sq[i], norms_sq[j]);
                    if (!keep) continue;
                    if (count < cap) {
                        out[count].row = (int32_t)i;                       /* :976 */
                        out[count].col = (int32_t)j;                       /* :977 */
                        out[count].dot = dot;                              /* :978 */
                        out[count].q = mvs_oracle_quantize(dot, d, norms_sq[i], norms_sq[j]);
                    }
                    ++count;
                }
            }
        }
    }
    free(tile);
    return count;
}

void mvs_oracle_dots_dense(const int32_t* sk, int64_t n, int d, int64_t r0, int64_t r1,
                           int64_t c0, int64_t c1, int32_t* out, int threads) {
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    (void)threads;
    (void)n;
    dots_tile(sk, 4, d, r0, r1, c0, c1, out);
}

int mvs_oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
